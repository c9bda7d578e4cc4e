#!/usr/bin/env python3
"""RCCL code path of the N > 1 bench on ONE GPU (world size 1): FrameExchange over backend "nccl" on its own stream,
ordered against the map with events (PipelinedExchange), checked against direct integration of the same frames; then the sharded
mesher's exchange (ShardedChisel.UpdateMeshes: all_gather_into_tensor of the dirty ids, all_to_all_single of the -- here empty --
shell payloads, event ordering both ways) against the plain recompute.
    python3 tools/nccl_world1_check.py        (inside gpurun)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch, torch.distributed as dist
from cvids_amd import synth
from cvids_amd.chisel import Chisel, ConstantWeighter, InverseTruncator, PinholeCamera, ProjectionIntegrator
from cvids_amd.sharded import FrameExchange, PipelinedExchange, ShardedChisel, pack_meta

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
W, H, K = 320, 240, 8
intr = synth.intrinsics(W, H)
cam = PinholeCamera(*intr, W, H, 0.05, 5.0)
integ = ProjectionIntegrator(InverseTruncator(2.0), ConstantWeighter(1.0), 0.05, True)
frames = list(synth.stream("sphere_room", 4 * K, W, H))
stack = [torch.from_numpy(np.stack([frames[b * K + j][0] for j in range(K)])).to(dev) for b in range(4)]
meta = [torch.from_numpy(np.stack([pack_meta(frames[b * K + j][1], cam) for j in range(K)])).to(dev) for b in range(4)]

ref = Chisel((16,) * 3, 0.02, False)
for b in range(4):
    ref.IntegrateBatch(integ, [(stack[b][j], frames[b * K + j][1], cam) for j in range(K)])
want = ref.fields()

m = Chisel((16,) * 3, 0.02, False)
xch = FrameExchange(W, H, K, dev, dist, channels=0)
xch.world = 2  # take the collective branch of exchange() although there is one rank (send and receive buffers of equal size)
xch.per = K
xch.send = [torch.zeros((K, xch.row), dtype=torch.float32, device=dev) for _ in range(2)]
px = PipelinedExchange(xch, m)
for b in range(4):
    depth, _, _ = px.exchange(b, stack[b], meta[b])
    m.IntegrateBatch(integ, [(depth[j], frames[b * K + j][1], cam) for j in range(K)])
    px.consumed(b)
got = m.fields()
assert set(got) == set(want) and len(got) > 50, (len(got), len(want))
for cid in want:
    assert np.array_equal(want[cid][0].view(np.uint32), got[cid][0].view(np.uint32)) and np.array_equal(want[cid][1].view(np.uint32), got[cid][1].view(np.uint32)), cid
# the sharded mesher's collectives with one rank: every chunk is a job of rank 0, no ghost is needed, the payloads are empty
xch.world = 1
sh = ShardedChisel(m, xch, integ)
sh.force_collectives = True
for mm in (ref, m):
    assert len(mm.GetMeshesToUpdate()) > 50
moved = sh.UpdateMeshes(force=True)
ref.UpdateMeshes(force=True)
assert moved == 0 and len(m.GetMeshesToUpdate()) == 0
ids_a, ids_b = sorted(map(tuple, ref.GetMeshIDs().tolist())), sorted(map(tuple, m.GetMeshIDs().tolist()))
assert ids_a == ids_b and len(ids_a) > 20, (len(ids_a), len(ids_b))
nv = 0
for cid in ids_a:
    a, b = ref.GetMesh(cid), m.GetMesh(cid)
    for key in ("vertices", "normals", "grids"):
        assert np.array_equal(np.asarray(a[key]).view(np.uint32), np.asarray(b[key]).view(np.uint32)), (cid, key)
    nv += len(a["vertices"])
print("sharded mesher over nccl, one rank: %d meshes, %d vertices identical" % (len(ids_a), nv))
# ... and its wait-free form (fixed segments, all_reduce of the status, all_to_all of equal splits, nothing read in between): a second
# stretch of frames into both maps, the first recompute above left the sizes behind
moved = sh.UpdateMeshes(force=True, wait_free=True)  # (the blocking form once more: it is the one that posts the sizes)
ref.UpdateMeshes(force=True)
for b in range(4):
    for mm in (ref, m):
        mm.IntegrateBatch(integ, [(torch.from_numpy(frames[(b * K + j + 3) % len(frames)][0]).to(dev), frames[(b * K + j + 3) % len(frames)][1], cam) for j in range(K)])
sh.Settle()
assert sh._est is not None
sh.UpdateMeshes(force=True, wait_free=True)
ref.UpdateMeshes(force=True)
assert sh.wait_free_recomputes == 1
sh.Settle()
assert getattr(sh, "wait_free_aborts", 0) == 0 and len(m.GetMeshesToUpdate()) == 0
ids_a, ids_b = sorted(map(tuple, ref.GetMeshIDs().tolist())), sorted(map(tuple, m.GetMeshIDs().tolist()))
assert ids_a == ids_b and len(ids_a) > 20, (len(ids_a), len(ids_b))
for cid in ids_a:
    a, b = ref.GetMesh(cid), m.GetMesh(cid)
    for key in ("vertices", "normals", "grids"):
        assert np.array_equal(np.asarray(a[key]).view(np.uint32), np.asarray(b[key]).view(np.uint32)), (cid, key)
print("wait-free form over nccl, one rank: %d meshes identical, status %s" % (len(ids_a), sh._est))
t = torch.tensor([1.0], device=dev)
dist.all_reduce(t)
dist.barrier()
print("nccl world-1 check ok: %d chunks identical, backend %s" % (len(got), dist.get_backend()))
dist.destroy_process_group()
