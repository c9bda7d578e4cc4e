#!/usr/bin/env python3
"""Run-to-run / mode-to-mode determinism of the map and the meshes at full image sizes (inside gpurun):
    python3 tools/determinism_sweep.py
Every configuration is run six times -- default schedule (x3), forced two-stream pipeline, conservative lookup mode,
single stream -- with a mesh recompute after every batch; voxels, counters and every chunk's mesh arrays must be
bit-identical."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from cvids_amd import synth, chisel as ch
def sweep(W, H, N, res, color_on, scale, n_frames, batch, scene="sphere_room", agents=1):
    intr = synth.intrinsics(W, H)
    cam = ch.PinholeCamera(*intr, W, H, 0.05, 5.0)
    integ = ch.ProjectionIntegrator(ch.InverseTruncator(scale), ch.ConstantWeighter(1.0), 0.05, True)
    frames = list(synth.stream(scene, n_frames, W, H, agents=agents))[:n_frames]
    color = synth.render_color(W, H, 3)
    dev = torch.device("cuda:0")
    d_dev = [torch.from_numpy(d).to(dev) for d, _ in frames]
    c_dev = torch.from_numpy(color).to(dev)
    def run(env, mesh=False):
        for k in ("CHISEL_HIP_FORCE_UNCERTAIN", "CHISEL_HIP_FORCE_PIPELINE", "CHISEL_HIP_SERIAL"):
            os.environ.pop(k, None)
        if env: os.environ[env] = "1"
        m = ch.Chisel((N,) * 3, res, color_on, max_chunks=20000 if N == 8 else 0)
        if env: os.environ.pop(env, None)
        for lo in range(0, n_frames, batch):
            idx = range(lo, min(lo + batch, n_frames))
            m.IntegrateBatch(integ, [(d_dev[i], frames[i][1], cam) for i in idx], [(c_dev, frames[i][1], cam) for i in idx] if color_on else None)
            if mesh: m.UpdateMeshes(force=True)
        f, c = m.fields(), m.counters()
        meshes = None
        if mesh:
            meshes = {tuple(i): m.GetMesh(tuple(i)) for i in m.GetMeshIDs().tolist()}
        m.close()
        return f, c, meshes
    ref_f, ref_c, ref_m = run(None, mesh=True)
    bad = 0
    for env in (None, None, "CHISEL_HIP_FORCE_PIPELINE", "CHISEL_HIP_FORCE_UNCERTAIN", "CHISEL_HIP_SERIAL"):
        f, c, ms = run(env, mesh=True)
        ok = set(f) == set(ref_f) and all(c[k] == ref_c[k] for k in ("sdf", "col", "probe", "carved", "new_chunks", "updated_chunks"))
        ok = ok and all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for cid in f for a, b in zip(f[cid], ref_f[cid]) if a is not None)
        ok = ok and set(ms) == set(ref_m) and all(np.array_equal(np.asarray(ms[k][key]).view(np.uint32), np.asarray(ref_m[k][key]).view(np.uint32)) for k in ms for key in ("vertices", "normals", "grids"))
        bad += not ok
        print("   ", env, "OK" if ok else "DIFFERENT", {k: (c[k], ref_c[k]) for k in c if c[k] != ref_c[k] and k != "work_chunks"})
    print("sweep", (W, H, N, res, color_on, batch, scene, agents), "chunks", len(ref_f), "meshes", len(ref_m), "->", "ok" if not bad else "%d BAD" % bad)
sweep(640, 480, 16, 0.01, True, 1.0, 27, 9)
sweep(640, 480, 8, 0.02, False, 2.0, 24, 7)
sweep(640, 480, 32, 0.01, True, 1.0, 12, 5)
sweep(640, 480, 16, 0.02, True, 2.0, 40, 13, scene="box_room", agents=2)
sweep(1280, 720, 16, 0.01, True, 1.0, 16, 8)
