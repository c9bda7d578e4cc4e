#!/usr/bin/env python3
"""Latency of the reference's own call pattern (inside gpurun): one frame per call, the caller waits after every call
(chisel::Chisel::IntegrateDepthScanColor of the facade does), UpdateMeshes() after every frame (ChiselServer.cpp:489-516).
    python3 tools/sync_latency.py [--frames 120]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvids_amd import synth
from cvids_amd.chisel import Chisel, ConstantWeighter, InverseTruncator, PinholeCamera, ProjectionIntegrator
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=120)
args = ap.parse_args()
W, H = 640, 480
intr = synth.intrinsics(W, H)
cam = PinholeCamera(*intr, W, H, 0.05, 5.0)
integ = ProjectionIntegrator(InverseTruncator(1.0), ConstantWeighter(1.0), 0.05, True)
frames = list(synth.stream("sphere_room", args.frames, W, H))
dev = torch.device("cuda:0")
d_dev = [torch.from_numpy(d).to(dev) for d, _ in frames]
c_dev = torch.from_numpy(synth.render_color(W, H, 3)).to(dev)
for mesh in (False, True):
    m = Chisel((16,) * 3, 0.01, True)
    t = []
    for i, (_, pose) in enumerate(frames):
        t0 = time.perf_counter()
        m.IntegrateDepthScanColor(integ, d_dev[i], pose, cam, c_dev, pose, cam)
        if mesh:
            m.UpdateMeshes()  # recomputes on every 10th call
        m.synchronize()
        t.append(time.perf_counter() - t0)
    t = np.array(t[20:]) * 1e6
    print("frame-by-frame, waiting after every frame%s: p50 %.1f us  p90 %.1f us  mean %.1f us  (%.0f frames/s)" % (
        " + UpdateMeshes()" if mesh else "", np.median(t), np.percentile(t, 90), t.mean(), 1e6 / t.mean()))
    m.close()
