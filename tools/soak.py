"""Soak: a long stream with recomputes, garbage collection, mesh downloads and resets; free device memory and pool state must settle.
python3 tools/soak.py [frames] [group shards] [grow]     grow: the pool starts at 128 chunks and grows on demand (chisel_hip_config.max_chunks < 0)"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvids_amd import chisel as ch, synth

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
n_group = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # > 0: an in-library group of that many shards on device 0
grow = len(sys.argv) > 3 and sys.argv[3] == "grow"
W, H, N, res = 320, 240, 16, 0.02
intr = synth.intrinsics(W, H)
cam = ch.PinholeCamera(*intr, W, H, 0.05, 5.0)
integ = ch.ProjectionIntegrator(ch.InverseTruncator(2.0), ch.ConstantWeighter(1.0), 0.05, True)
color = synth.render_color(W, H, 3)
m = ch.Chisel((N,) * 3, res, True, max_chunks=-128 if grow else 1 << 14, **({"devices": [0] * n_group} if n_group else {}))
frames = list(synth.stream("sphere_room", 300, W, H, agents=2))
dev = torch.device("cuda:0")
free0 = None
t0 = time.perf_counter()
log = []
for k in range(0, n_frames, 10):
    part = [frames[(k + j) % len(frames)] for j in range(10)]
    m.IntegrateBatch(integ, [(d, p, cam) for d, p in part], [(color, p, cam) for _, p in part])
    m.UpdateMeshes(force=True)
    if k % 50 == 40:
        ids = np.asarray(m.GetChunkIDs()).reshape(-1, 3)
        if len(ids):
            m.GarbageCollect(ids[:: 7])          # every seventh chunk goes (and comes back with later frames)
    if k % 100 == 90:
        meshes = m.GetMeshIDs() if hasattr(m, "GetMeshIDs") else []
        for cid in list(map(tuple, np.asarray(meshes).reshape(-1, 3)[:8].tolist())):
            m.GetMesh(cid)
    if k % 300 == 290:
        m.synchronize()
        free, total = torch.cuda.mem_get_info(dev)
        if free0 is None:
            free0 = free
        log.append((k + 10, m.NumChunks(), free / 2**20))
        m.Reset()
m.synchronize()
dt = time.perf_counter() - t0
print("soak: %d frames in %.1f s (%.0f frames/s incl. recompute every 10 and host-side work)" % (n_frames, dt, n_frames / dt))
for row in log:
    print("   after %5d frames: %5d chunks resident, %.0f MiB free" % row)
warm = log[0][2] - log[min(3, len(log) - 1)][2]
drift = log[min(3, len(log) - 1)][2] - log[-1][2]
print("warm-up (arenas, pools, staging) first -> fourth checkpoint: %.1f MiB; drift fourth -> last checkpoint: %.1f MiB" % (warm, drift))
assert abs(drift) < 16.0, "device memory keeps growing"
# the stream repeats every 600 frames (and the listing the garbage collection picks from is in ascending id order): so do the counts
for i in range(2, len(log)):
    assert log[i][1] == log[i - 2][1], "chunk counts of identical stretches of the stream differ: %d vs %d after %d frames" % (log[i][1], log[i - 2][1], log[i][0])
if grow and not n_group:
    info = m.pool_info()
    print("pool:", info)
    assert info["growable"] and info["committed"] > 128 and info["grown"] > 0, "the pool never grew"
print("soak ok")
