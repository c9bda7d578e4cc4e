#!/usr/bin/env python3
"""Point-cloud fusion mode (chisel_hip_integrate_pointcloud) on one GPU: clouds per second on a synthetic stream.

Workload: the sphere room scaled to 1.5 m (the reference skips points deeper than 5 m / 2 m), one cloud per camera pose of
BASELINE's trajectory, W x H points each (default 640 x 480 = the depth image ChiselServer back-projects), 1 cm voxels, 16^3
chunks, colours, InverseTruncator(1), clouds resident in HBM.  Prints one JSON line; `--cpu` times the oracle on a 160 x 120
cloud of the same scene beside it (the reference's loop is chunks x points ray walks: seconds per cloud at full size).
Not the driver's bench (that is bench.py, the depth-image path BASELINE.json names); this is the measurement of SURVEY 8(f) row 3.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--res", type=float, default=0.01)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-color", action="store_true")
    ap.add_argument("--cpu", action="store_true")
    args = ap.parse_args()
    import torch
    from cvids_amd import chisel as ch
    from cvids_amd import synth
    W, H = args.width, args.height
    intr = synth.intrinsics(W, H)
    color = not args.no_color
    scale = 0.6 if color else 0.3   # depth limit 5 m with colours, 2 m without
    n = args.steps + args.warmup
    clouds = []
    for k in range(n):
        pose = synth.trajectory_pose(k)
        pose = pose.copy()
        pose[:3, 3] *= np.float32(scale)
        depth = synth.render_depth("sphere_room", synth.trajectory_pose(k), intr, W, H)
        out = synth.depth_to_cloud(depth, intr, scale, colors=color)
        pts, col = out if color else (out, None)
        clouds.append((torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda() if color else None, pose, len(pts)))
    gm = ch.Chisel((16, 16, 16), args.res, color, max_chunks=60000)
    integ = ch.ProjectionIntegrator(ch.InverseTruncator(1.0 * args.res / 0.01), ch.ConstantWeighter(1.0), 0.05, True)
    for k in range(args.warmup):
        p, c, pose, _ = clouds[k]
        gm.IntegratePointCloud(integ, (p, c), pose, 0.1, 5.0)
    gm.synchronize()
    gm.counters(reset=True)
    gm.set_profiling(True)
    t0 = time.perf_counter()
    for k in range(args.warmup, n):
        p, c, pose, _ = clouds[k]
        gm.IntegratePointCloud(integ, (p, c), pose, 0.1, 5.0)
    gm.synchronize()
    dt = time.perf_counter() - t0
    cnt = gm.counters()
    prof = gm.profile()["cloud"]
    points = sum(c[3] for c in clouds[args.warmup:])
    out = {
        "metric": "point clouds/sec integrated (point-cloud fusion mode)", "value": args.steps / dt, "unit": "clouds/s",
        "ms_per_cloud": 1e3 * dt / args.steps, "device_ms_per_cloud": prof["ms"] / max(1, prof["launches"]),
        "Mpoints_per_s": points / dt / 1e6, "Mvoxel_updates_per_s": (cnt["sdf"] + cnt["carved"]) / dt / 1e6,
        "ray_cells_per_cloud": cnt["probe"] / args.steps, "listed_chunks_per_cloud": cnt["work_chunks"] / args.steps,
        "chunks": gm.NumChunks(), "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "sphere room x %.1f, %dx%d points per cloud, %.3g m voxels, 16^3 chunks, %s" %
                   (scale, W, H, args.res, "colour" if color else "no colour")},
    }
    import ctypes as C
    st = (C.c_int64 * 4)()
    if gm.L.chisel_hip_debug_cloud_stats(gm.h, st) == 0:
        out["last_cloud"] = {"listed_chunks": st[0], "unit_point_pairs": st[1], "rays_of_largest_unit": st[2], "units_with_rays": st[3]}
    if args.cpu:
        import oracle
        w2, h2 = 160, 120
        intr2 = synth.intrinsics(w2, h2)
        om = oracle.OracleMap(16, args.res, color)
        om.set_integrator(oracle.TRUNC_INVERSE, 1.0 * args.res / 0.01, 1.0, True, 0.05)
        t = time.perf_counter()
        m = 3
        npts = 0
        for k in range(m):
            pose = synth.trajectory_pose(k).copy()
            pose[:3, 3] *= np.float32(scale)
            depth = synth.render_depth("sphere_room", synth.trajectory_pose(k), intr2, w2, h2)
            o = synth.depth_to_cloud(depth, intr2, scale, colors=color)
            pts, col = o if color else (o, None)
            om.integrate_pointcloud(pts, pose, col, 0.1, 5.0)
            npts += len(pts)
        dtc = time.perf_counter() - t
        out["cpu_baseline"] = {"value": m / dtc, "unit": "clouds/s", "Mpoints_per_s": npts / dtc / 1e6, "cores": 1, "kind": "port",
                               "sample": "%d clouds of %dx%d points, same scene (the work grows with chunks x points)" % (m, w2, h2)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
