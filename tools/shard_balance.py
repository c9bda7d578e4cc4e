#!/usr/bin/env python3
"""How evenly the chunks a launch set updates fall on the ranks, per ownership-block edge (chisel_hip_config.shard_block) -- host arithmetic
only, no GPU and no map: every pixel of the driver's stream (640x480 @ 1 cm, 16^3 chunks, ten frames per launch set) is put back into the world
at its depth and at both ends of its truncation band (InverseTruncator(1): d^2 / (0.10 * 471.27), InverseTruncator.h:42-52, plus a voxel diagonal: the
band the integration updates), the chunks
those points fall into are the set's work, chunk_owner deals them out.  A launch set lasts as long as its slowest rank: what counts is
max / mean PER SET, not over the run.
    python3 tools/shard_balance.py [sets] [agents]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from cvids_amd import synth

def owner(ids, world, block):
    """chunk_owner of kernels_cull.h / chisel_hip_chunk_owner, vectorised: (bx + 3 by + 5 bz) mod world on blocks of `block` chunks"""
    b = np.floor_divide(ids, block)
    return np.mod(b[:, 0] + 3 * b[:, 1] + 5 * b[:, 2], world)

n_sets = int(sys.argv[1]) if len(sys.argv) > 1 else 20
agents = int(sys.argv[2]) if len(sys.argv) > 2 else 1
W, H, N, res = 640, 480, 16, 0.01
K = 10 if agents == 1 else 16
fx, fy, cx, cy = synth.intrinsics(W, H)
u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
frames = list(synth.stream("sphere_room", (n_sets * K) // agents + 1, W, H, agents=agents))[:n_sets * K]
rows = {}
for s in range(n_sets):
    ids = []
    for d, pose in frames[s * K:(s + 1) * K]:
        d = np.asarray(d, np.float64)
        ok = np.isfinite(d) & (d > 0.05) & (d < 5.0)
        band = d * d / (0.10 * 471.27) + res * np.sqrt(3.0)
        for dd in (d, d - band, d - 0.5 * band, d + 0.5 * band, d + band):   # the surface, both ends of the band and half way
            dd = np.where(ok, np.clip(dd, 0.05, 6.0), np.nan)
            pc = np.stack([(u - cx) / fx * dd, (v - cy) / fy * dd, dd], -1)[ok]
            pw = pc @ np.asarray(pose, np.float64)[:3, :3].T + np.asarray(pose, np.float64)[:3, 3]
            ids.append(np.unique(np.floor(pw / (N * res)).astype(np.int64), axis=0))
    ids = np.unique(np.concatenate(ids), axis=0)
    for block in (2, 4, 8):
        for world in (2, 4, 8):
            cnt = np.bincount(owner(ids, world, block), minlength=world)
            rows.setdefault((block, world), []).append((cnt.max(), cnt.mean()))
    print("set %2d: %5d chunks in the band" % (s, len(ids)), flush=True)
print("max / mean of a launch set's chunks per rank (median and worst of %d sets; %d agent%s, %d frames per set):" % (n_sets, agents, "" if agents == 1 else "s", K))
print("  (sets of fewer than 200 chunks are left out of median and worst -- a handful of chunks cannot be dealt evenly and does not take long either;")
print("   `whole stream` = the slowest ranks' chunks summed over all sets / the mean ranks' summed: what the imbalance costs the stream)")
for (block, world), vals in sorted(rows.items()):
    big = [mx / mean for mx, mean in vals if mean * world >= 200]
    print("  block %d, %d ranks: median %.3f  worst %.3f  whole stream %.3f" % (block, world, float(np.median(big)), max(big), sum(mx for mx, _ in vals) / sum(mean for _, mean in vals)))
