"""Marching-cubes case table: the three encodings agree and obey the algorithm's invariants (CPU only)."""
import json
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CORNERS = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]  # ChunkManager.cpp:67-69


def golden():
    return json.load(open(os.path.join(HERE, "golden", "mc_triangle_table.json")))


def product_table():
    txt = open(os.path.join(ROOT, "cvids_amd", "csrc", "mc_tables.h")).read()
    body = txt.split("CHISEL_MC_PACKED_CASES", 1)[1].split("}", 1)[0]
    words = [int(w, 16) for w in re.findall(r"0x([0-9a-f]{16})ull", body)]
    assert len(words) == 256
    table = np.full((256, 16), -1, np.int32)
    for i, w in enumerate(words):
        for k in range(16):
            nib = (w >> (4 * k)) & 0xF
            if nib == 0xF:
                break
            table[i, k] = nib
    counts = [int(v) for v in re.findall(r"\d+", txt.split("CHISEL_MC_VERTEX_COUNTS", 1)[1].split("}", 1)[0])]
    edges = [int(v, 16) for v in re.findall(r"0x([0-9a-f]{2})", txt.split("CHISEL_MC_EDGE_CORNERS", 1)[1])]
    return table, counts, [(e & 0xF, e >> 4) for e in edges]


def test_three_encodings_agree(oracle_mod):
    g = golden()
    gt = np.array(g["triangle_table"], np.int32)
    assert np.array_equal(oracle_mod.triangle_table(), gt)
    pt, counts, edges = product_table()
    assert np.array_equal(pt, gt)
    assert counts == [int((row >= 0).sum()) for row in gt]
    assert edges == [tuple(p) for p in g["edge_index_pairs"]]


def test_table_invariants():
    g = golden()
    table, pairs = np.array(g["triangle_table"]), g["edge_index_pairs"]
    # the 12 edges join corners that differ in exactly one coordinate
    for a, b in pairs:
        assert sum(abs(p - q) for p, q in zip(CORNERS[a], CORNERS[b])) == 1
    assert (table[0] == -1).all() and (table[255] == -1).all()
    for case in range(256):
        row = table[case]
        n = int((row >= 0).sum())
        assert n % 3 == 0 and n <= 15 and (row[n:] == -1).all()
        inside = [(case >> i) & 1 for i in range(8)]
        used = set(int(e) for e in row[:n])
        crossing = {e for e, (a, b) in enumerate(pairs) if inside[a] != inside[b]}
        # triangles only use edges with a sign change, and every crossed edge is used
        assert used == crossing, case
        # complementary configurations cut the same edges
        assert set(int(e) for e in table[255 - case] if e >= 0) == used
