#!/usr/bin/env python3
"""RCCL code path of the N > 1 bench on ONE GPU (world size 1): FrameExchange over backend "nccl" on its own stream,
ordered against the map with events (PipelinedExchange), checked against direct integration of the same frames.
    python3 tools/nccl_world1_check.py        (inside gpurun)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch, torch.distributed as dist
from cvids_amd import synth
from cvids_amd.chisel import Chisel, ConstantWeighter, InverseTruncator, PinholeCamera, ProjectionIntegrator
from cvids_amd.sharded import FrameExchange, PipelinedExchange, pack_meta

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
t = torch.tensor([1.0], device=dev)
dist.all_reduce(t)
dist.barrier()
print("nccl world-1 check ok: %d chunks identical, backend %s" % (len(got), dist.get_backend()))
dist.destroy_process_group()
