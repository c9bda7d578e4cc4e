#!/usr/bin/env python3
"""What one rank of an N-way sharded map does, measured on ONE GPU: the frames of the whole stream, the chunks of one shard.
Every rank of the real run sees every frame, so the job's frame rate is the slowest rank's; the all-gather is not in here.
The frame structs are rebuilt in Python for every call, so the RATE printed here is host-bound at high shard counts: use
`bench.py --sim-shards N --sim-rank r` for rates (structs prebuilt) and this script under rocprofv3 for per-kernel times
(tools/shard_kstats.sh).
    python3 tools/shard_sim.py [--agents 4] [--batch 16] [--worlds 1,2,4,8]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=4)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--frames", type=int, default=320)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--worlds", default="1,2,4,8")
    args = ap.parse_args()
    import torch
    from cvids_amd import chisel as ch
    from cvids_amd import synth
    W, H = 640, 480
    intr = synth.intrinsics(W, H)
    cam = ch.PinholeCamera(*intr, W, H, 0.05, 5.0)
    total = args.frames + args.warmup
    frames = list(synth.stream("sphere_room", (total + args.agents - 1) // args.agents, W, H, agents=args.agents))[:total]
    dev = torch.device("cuda:0")
    d_dev = [torch.from_numpy(d).to(dev) for d, _ in frames]
    c_dev = torch.from_numpy(synth.render_color(W, H, 3)).to(dev)
    integ = ch.ProjectionIntegrator(ch.InverseTruncator(1.0), ch.ConstantWeighter(1.0), 0.05, True)
    out = {}
    for world in [int(v) for v in args.worlds.split(",")]:
        rates = []
        for rank in sorted(set([0, world // 2, world - 1])):
            gm = ch.Chisel((16, 16, 16), 0.01, True, max_chunks=40000, n_shards=world, shard_rank=rank, shard_block=2)

            def run(a, b):
                for s in range(a, b, args.batch):
                    fr = [(d_dev[i], frames[i][1], cam) for i in range(s, min(s + args.batch, b))]
                    gm.IntegrateBatch(integ, fr, [(c_dev, f[1], cam) for f in fr])
            run(0, args.warmup)
            gm.synchronize()
            t0 = time.perf_counter()
            run(args.warmup, total)
            gm.synchronize()
            rates.append(args.frames / (time.perf_counter() - t0))
            gm.close()
        out[world] = {"frames_per_s_of_the_slowest_rank_tried": min(rates), "ranks": [round(r) for r in rates]}
    print(json.dumps({"agents": args.agents, "batch": args.batch, "per_world": out}))


if __name__ == "__main__":
    main()
