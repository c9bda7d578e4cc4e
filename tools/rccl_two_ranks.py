#!/usr/bin/env python3
"""The collectives the N > 1 path issues, on their own, between real ranks: all_gather_into_tensor of a fixed-capacity int32 block (the
dirty lists, the frames), all_to_all_single of uint8 segments with uneven splits (the shell segments), an all_reduce.  Started by
torchrun (RANK / LOCAL_RANK / WORLD_SIZE from the environment):
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/rccl_two_ranks.py
    ... tools/rccl_two_ranks.py --backend gloo      (CPU tensors: the form that runs in the build container)"""
import argparse, os, sys
import torch, torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("--backend", default="nccl")
args = ap.parse_args()
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", 0))
if args.backend == "nccl":
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
else:
    dev = torch.device("cpu")
    dist.init_process_group(args.backend)
# 1. all_gather_into_tensor, int32, fixed capacity
cap = 1 + 4 * 4096
mine = torch.full((cap,), rank + 1, dtype=torch.int32, device=dev)
gathered = torch.zeros((world * cap,), dtype=torch.int32, device=dev)
dist.all_gather_into_tensor(gathered, mine)
for r in range(world):
    assert int(gathered[r * cap]) == r + 1 and int(gathered[(r + 1) * cap - 1]) == r + 1, "all_gather_into_tensor: block %d" % r
# 2. all_to_all_single, uint8, uneven splits: rank s sends (s + 1) * (r + 3) * 1000 bytes of value 16 s + r to rank r
send_sizes = [(rank + 1) * (r + 3) * 1000 for r in range(world)]
recv_sizes = [(s + 1) * (rank + 3) * 1000 for s in range(world)]
send = torch.cat([torch.full((n,), 16 * rank + r, dtype=torch.uint8, device=dev) for r, n in enumerate(send_sizes)])
recv = torch.zeros((sum(recv_sizes),), dtype=torch.uint8, device=dev)
dist.all_to_all_single(recv, send, recv_sizes, send_sizes)
at = 0
for s, n in enumerate(recv_sizes):
    seg = recv[at:at + n]
    assert int(seg.min()) == int(seg.max()) == 16 * s + rank, "all_to_all_single: segment from rank %d" % s
    at += n
# 3. on a side stream, ordered by events against the current one (what PipelinedExchange does)
if dev.type == "cuda":
    side = torch.cuda.Stream(device=dev)
    ready, done = torch.cuda.Event(), torch.cuda.Event()
    x = torch.ones((1 << 20,), dtype=torch.float32, device=dev) * (rank + 1)
    ready.record()
    with torch.cuda.stream(side):
        side.wait_event(ready)
        dist.all_reduce(x)
        done.record(side)
    torch.cuda.current_stream(dev).wait_event(done)
    assert float(x[0]) == world * (world + 1) / 2
    torch.cuda.synchronize()
else:
    x = torch.ones((1024,)) * (rank + 1)
    dist.all_reduce(x)
    assert float(x[0]) == world * (world + 1) / 2
dist.barrier()
if rank == 0:
    print("collectives ok: backend %s, %d ranks (all_gather_into_tensor int32, all_to_all_single uint8 with uneven splits, all_reduce on a side stream)" % (dist.get_backend(), world))
dist.destroy_process_group()
