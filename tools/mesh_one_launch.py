"""Diagnostic (libchisel_hip_ph.so, -DCHISEL_PHASES): ONE mesh recompute after n frames of the bench stream -- the count kernel's stage
timers, the lives of its working waves and the launch's span from the first wave's entry to the last one's exit.
   CHISEL_HIP_LIB=libchisel_hip_ph.so python3 tools/mesh_one_launch.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cvids_amd import synth
from cvids_amd.chisel import Chisel, ConstantWeighter, InverseTruncator, PinholeCamera, ProjectionIntegrator

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
W, H = 640, 480
intr = synth.intrinsics(W, H)
cam = PinholeCamera(*intr, W, H, 0.05, 5.0)
color = synth.render_color(W, H, 3)
gm = Chisel((16, 16, 16), 0.01, True, device_id=0)
integ = ProjectionIntegrator(InverseTruncator(1.0), ConstantWeighter(1.0), 0.05, True)
frames = list(synth.stream("sphere_room", n + 10, W, H))
for lo in range(0, n + 10, 10):  # the last ten frames and their recompute are the measured ones (everything before warms the code up)
    part = frames[lo:lo + 10]
    gm.IntegrateBatch(integ, [(d, p, cam) for d, p in part], [(color, p, cam) for _, p in part])
    gm.synchronize()
    gm.counters(reset=True)  # prints and zeroes the diagnostics so far
    gm.UpdateMeshes(force=True)
    gm.synchronize()
    print("--- recompute after frame %d" % (lo + 10), file=sys.stderr)
    gm.counters(reset=True)
