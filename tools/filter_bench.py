#!/usr/bin/env python3
"""DepthFilter::Update on the GPU (chisel_hip_depth_filter_update): microseconds per 640x480 update, inputs resident in HBM,
beside the numpy restatement on the host.  Per update the kernel reads 4 state maps + 1 input map and writes up to 4 maps of
W*H doubles = 72 bytes per pixel."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from cvids_amd.chisel import DepthFilter
    from oracle.depth_filter import DepthFilter as Oracle
    H, W, n = 480, 640, 500
    rng = np.random.default_rng(1)
    mu = rng.uniform(0.3, 1.2, (H, W))
    d_mu = torch.from_numpy(mu).cuda()
    gf = DepthFilter(H, W)
    for _ in range(20):
        gf.Update(d_mu, 4.05e-3)
    gf.GetA()
    t0 = time.perf_counter()
    for _ in range(n):
        gf.Update(d_mu, 4.05e-3)
    gf.GetA()  # waits
    dt = (time.perf_counter() - t0) / n
    of = Oracle(H, W)
    t0 = time.perf_counter()
    for _ in range(5):
        of.update(mu, 4.05e-3)
    dtc = (time.perf_counter() - t0) / 5
    print(json.dumps({"metric": "DepthFilter::Update, 640x480", "gpu_us_per_update": dt * 1e6, "GBps_algorithmic": 72.0 * H * W / dt / 1e9,
                      "numpy_ms_per_update": dtc * 1e3}))


if __name__ == "__main__":
    main()
