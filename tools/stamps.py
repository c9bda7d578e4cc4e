#!/usr/bin/env python3
"""Timeline of integrate_kernel from s_memrealtime stamps (diagnostic build: make -C cvids_amd/csrc stamps).

    CHISEL_HIP_LIB=libchisel_hip_stamps.so python3 tools/stamps.py [--frames 30] [--batch 1]
Stamps per workgroup (100 MHz ticks): 0 entry, 1 work_count read, 2 item read, 3 tile staged (last frame),
4 frame applied (last frame), 5 item stored, 6 exit.  Only the first item of each workgroup is stamped.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--res", type=float, default=0.01)
    ap.add_argument("--no-color", action="store_true")
    args = ap.parse_args()
    import torch
    from cvids_amd import capi, synth
    from cvids_amd.chisel import Chisel, ConstantWeighter, InverseTruncator, PinholeCamera, ProjectionIntegrator
    W, H = 640, 480
    intr = synth.intrinsics(W, H)
    cam = PinholeCamera(*intr, W, H, 0.05, 5.0)
    integ = ProjectionIntegrator(InverseTruncator(100 * args.res), ConstantWeighter(1.0), 0.05, True)
    frames = list(synth.stream("sphere_room", args.frames, W, H))
    dev = torch.device("cuda:0")
    d_dev = [torch.from_numpy(d).to(dev) for d, _ in frames]
    c_dev = torch.from_numpy(synth.render_color(W, H, 3)).to(dev)
    m = Chisel((16,) * 3, args.res, not args.no_color)
    L = m.L
    L.chisel_hip_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    G = 2048
    buf = np.zeros((G, 32), np.uint64)
    ptr = buf.ctypes.data_as(C.POINTER(C.c_uint64))
    n_warm = args.frames - args.batch
    for i in range(n_warm):
        if args.no_color:
            m.IntegrateDepthScan(integ, d_dev[i], frames[i][1], cam)
        else:
            m.IntegrateDepthScanColor(integ, d_dev[i], frames[i][1], cam, c_dev, frames[i][1], cam)
    m.synchronize()
    capi.check(L.chisel_hip_debug_stamps(m.h, None, 0))
    fr = [(d_dev[i], frames[i][1], cam) for i in range(n_warm, args.frames)]
    co = None if args.no_color else [(c_dev, frames[i][1], cam) for i in range(n_warm, args.frames)]
    m.IntegrateBatch(integ, fr, co)
    m.synchronize()
    capi.check(L.chisel_hip_debug_stamps(m.h, ptr, G))
    s = buf.astype(np.int64)
    live = s[:, 0] > 0
    t0 = s[live, 0].min()
    print("workgroups that ran: %d; with an item: %d" % (live.sum(), (s[:, 2] > 0).sum()))
    us = lambda a: (a - t0) / 100.0
    for name, col in (("entry", 0), ("work_count read", 1), ("item read", 2), ("tile staged", 3), ("frame applied", 4),
                      ("item stored", 5), ("exit", 6)):
        v = s[:, col][s[:, col] > 0]
        if len(v):
            x = us(v)
            print("  %-16s n %5d  min %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f us" % (name, len(v), x.min(), np.median(x), np.percentile(x, 90), x.max()))
    ex = s[live, 6]
    en = s[live, 0]
    dur = (ex.max() - t0) / 100.0
    busy = ((ex - en) / 100.0).sum()
    # resident workgroups at once: 2 per CU (register file), 256 CUs
    print("  kernel (first entry -> last exit) %.1f us; workgroup-time %.0f us = %.1f %% of 512 slots x kernel" % (dur, busy, 100.0 * busy / (512 * dur)))
    order = np.sort(us(ex))
    print("  workgroups still running at 50/60/70/80/90 %% of the kernel: %s" % ", ".join(
        "%d" % (ex > t0 + f * (ex.max() - t0)).sum() for f in (0.5, 0.6, 0.7, 0.8, 0.9)))
    w = s[:, 2] > 0
    for a, b, name in ((0, 1, "entry->count"), (1, 2, "count->item"), (2, 16, "item->prefetched"), (16, 17, "prefetched->staged(t0)"), (17, 3, "staged(t0)->barrier"), (3, 4, "tile->applied"), (4, 5, "applied->stored"), (5, 6, "stored->exit")):
        ok = w & (s[:, a] > 0) & (s[:, b] > 0)
        d = (s[ok, b] - s[ok, a]) / 100.0
        if len(d):
            print("  d %-16s p50 %7.2f  p90 %7.2f  max %7.2f us" % (name, np.median(d), np.percentile(d, 90), d.max()))
    ok = w & (s[:, 7] > 0) & (s[:, 6] > s[:, 0])
    if ok.any():
        mhz = s[ok, 7] / ((s[ok, 6] - s[ok, 0]) / 100.0)
        print("  shader clock during the kernel: p50 %.0f MHz (min %.0f, max %.0f)" % (np.median(mhz), mhz.min(), mhz.max()))
    names = ["scalars + z tests", "tile wait + barrier", "dma issue + apply", "hand-shake", "item prologue"]
    it_n = np.maximum(1, s[w, 13]).astype(float)
    for k, nm in enumerate(names):
        v = s[w, 8 + k] / (it_n if k < 4 else 1.0)
        print("  wave-0 cycles per frame iteration: %-20s mean %8.0f  p50 %8.0f  p90 %8.0f" % (nm, v.mean(), np.median(v), np.percentile(v, 90)))
    print("  frame iterations per workgroup: mean %.1f" % it_n.mean())
    tp = s[w, 14]
    fl = s[w, 15]
    bw, bh, nq = (fl >> 8) & 0xffff, (fl >> 24) & 0xffff, (fl >> 40) & 0xff
    print("  staged tile pixels: p50 %d p90 %d max %d; items without a staged tile: %d of %d; flags inband %d carve %d tile %d" % (
        np.median(tp), np.percentile(tp, 90), tp.max(), (tp == 0).sum(), len(tp), ((fl & 1) > 0).sum(), ((fl & 2) > 0).sum(), ((fl & 4) > 0).sum()))
    print("  box w p50 %d max %d, h p50 %d max %d; needed quads of thread 0: p50 %d" % (np.median(bw), bw.max(), np.median(bh), bh.max(), np.median(nq)))
    ln = s[live, 18:22].sum(axis=0).astype(float)
    wv = s[live, 22:26].sum(axis=0).astype(float) / 64.0
    for k, nm in enumerate(("quad projection", "band update", "colour update", "carve test")):
        print("  wave executions of %-16s %9.0f  lanes that wanted it %11.0f  = %.1f %% of the lanes" % (nm, wv[k], ln[k], 100.0 * ln[k] / max(1.0, 64.0 * wv[k])))
    cs = s[:, 26:32]
    cl = cs[:, 0] > 0
    if cl.any():
        c0 = cs[cl, 0].min()
        print("cull kernel blocks stamped: %d" % cl.sum())
        for name, col in (("entry", 0), ("own frame judged (wave 0)", 1), ("after barrier 1", 2), ("merged+compacted", 3), ("after barrier 2", 4), ("exit", 5)):
            v = cs[cl, col]
            v = v[v > 0]
            x = (v - c0) / 100.0
            print("  cull %-26s n %5d  min %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f us" % (name, len(v), x.min(), np.median(x), np.percentile(x, 90), x.max()))
    print("counters:", m.counters())


if __name__ == "__main__":
    main()
