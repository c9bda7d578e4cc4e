"""How many chunks are in meshesToUpdate before each recompute: one map against an in-library group of 2 on the same stream."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvids_amd import synth
from cvids_amd import chisel as ch

W, H, N, res = 320, 240, 16, 0.02
intr = synth.intrinsics(W, H)
cam = ch.PinholeCamera(*intr, W, H, 0.05, 5.0)
integ = ch.ProjectionIntegrator(ch.InverseTruncator(2.0), ch.ConstantWeighter(1.0), 0.05, True)
color = synth.render_color(W, H, 3)
frames = list(synth.stream("sphere_room", 120, W, H))
single = ch.Chisel((N,) * 3, res, True, max_chunks=1 << 14)
grp = ch.Chisel((N,) * 3, res, True, max_chunks=1 << 14, devices=[0, 0])
for lo in range(0, 120, 10):
    part = frames[lo:lo + 10]
    row = []
    for m in (single, grp):
        m.IntegrateBatch(integ, [(d, p, cam) for d, p in part], [(color, p, cam) for _, p in part])
        ids = np.asarray(m.GetMeshesToUpdate()).reshape(-1, 3)
        row.append(len(ids))
        m.UpdateMeshes(force=True)
    print("frames %3d-%3d: meshesToUpdate single %5d  group %5d | chunks %d %d" % (lo, lo + 10, row[0], row[1], single.NumChunks(), grp.NumChunks()))
