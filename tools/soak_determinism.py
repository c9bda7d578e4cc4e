"""Is the soak's stream deterministic?  The first 300 frames of tools/soak.py several times over in one process, with parts of the loop
switched off, comparing the resident chunk ids and a checksum of the voxel arrays between repetitions.
python3 tools/soak_determinism.py"""
import os, sys, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvids_amd import chisel as ch, synth

W, H, N, res = 320, 240, 16, 0.02
intr = synth.intrinsics(W, H)
cam = ch.PinholeCamera(*intr, W, H, 0.05, 5.0)
integ = ch.ProjectionIntegrator(ch.InverseTruncator(2.0), ch.ConstantWeighter(1.0), 0.05, True)
color = synth.render_color(W, H, 3)
frames = list(synth.stream("sphere_room", 150, W, H, agents=2))

def run(mesh, gc, getmesh, wait):
    m = ch.Chisel((N,) * 3, res, True, max_chunks=1 << 14)
    for k in range(0, 300, 10):
        part = [frames[(k + j) % len(frames)] for j in range(10)]
        m.IntegrateBatch(integ, [(d, p, cam) for d, p in part], [(color, p, cam) for _, p in part])
        if wait:
            m.synchronize()
        if mesh:
            m.UpdateMeshes(force=True)
        if gc and k % 50 == 40:
            ids = np.asarray(m.GetChunkIDs()).reshape(-1, 3)
            if len(ids):
                m.GarbageCollect(ids[::7])
        if getmesh and k % 100 == 90:
            for cid in list(map(tuple, np.asarray(m.GetMeshIDs()).reshape(-1, 3)[:8].tolist())):
                m.GetMesh(cid)
    m.synchronize()
    f = m.fields()
    crc = 0
    for cid in sorted(f):
        for a in f[cid]:
            if a is not None:
                crc = zlib.crc32(np.ascontiguousarray(a).tobytes(), crc)
    return len(f), crc

for name, kw in (("all", dict(mesh=True, gc=True, getmesh=True, wait=False)), ("no gc", dict(mesh=True, gc=False, getmesh=True, wait=False)),
                 ("no mesh", dict(mesh=False, gc=True, getmesh=False, wait=False)), ("gc only, waiting", dict(mesh=False, gc=True, getmesh=False, wait=True)),
                 ("integration only", dict(mesh=False, gc=False, getmesh=False, wait=False)), ("all, waiting", dict(mesh=True, gc=True, getmesh=True, wait=True))):
    out = [run(**kw) for _ in range(5)]
    print("%-20s %s  -> %s" % (name, out, "deterministic" if len(set(out)) == 1 else "DIFFERS"))
