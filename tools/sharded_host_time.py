"""What a wait-free sharded recompute (cvids_amd/sharded.py: ShardedChisel._recompute_wait_free) costs the host, call by call: one process, two
shards of the driver's stream (640x480 @ 1 cm, a recompute every 10 frames) on one GPU, the collectives stood in for by slice copies and
left out of the timing -- what tools/n_ranks_one_gpu.sh cannot show, because there two processes and gloo's threads share the host's cores.
    python3 tools/sharded_host_time.py [shards]        (inside gpurun)
Under `rocprofv3 --kernel-trace --stats -d DIR -o t -- python3 tools/sharded_host_time.py 8` the kernel trace gives what ONE rank's GPU does per
recompute (tools/sharded_kernel_time.py DIR sums it up by step)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from cvids_amd import chisel as ch, synth
W, H, N, res, n_shards = 640, 480, 16, 0.01, (int(sys.argv[1]) if len(sys.argv) > 1 else 2)
intr = synth.intrinsics(W, H)
cam = ch.PinholeCamera(*intr, W, H, 0.05, 5.0)
integ = ch.ProjectionIntegrator(ch.InverseTruncator(1.0), ch.ConstantWeighter(1.0), 0.05, True)
color = synth.render_color(W, H, 3)
shards = [ch.Chisel((N,) * 3, res, True, n_shards=n_shards, shard_rank=r) for r in range(n_shards)]
frames = list(synth.stream("sphere_room", 120, W, H))
dev = torch.device("cuda", 0)
cap = 1 << 12
acc = {}
def T(name, f):
    t0 = time.perf_counter(); r = f(); acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e6); return r
stride = None
for k in range(0, 120, 10):
    part = frames[k:k + 10]
    for s_ in shards:
        s_.IntegrateBatch(integ, [(d, p, cam) for d, p in part], [(color, p, cam) for _, p in part])
    gathered = torch.zeros((n_shards, 1 + 4 * cap), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for r, s_ in enumerate(shards):
        T("dirty ids", lambda: s_.DirtyIdsDevice(gathered[r]))
        s_.synchronize()
    if stride is None:
        plans = [s_.PlanShellsDevice(gathered.view(-1), n_shards, cap) for s_ in shards]
        sizes = [[s_.ShellSegmentBytes(*plans[r]["send"][p]) for p in range(n_shards)] for r, s_ in enumerate(shards)]
        send = [torch.empty((sum(sz),), dtype=torch.uint8, device=dev) for sz in sizes]
        torch.cuda.synchronize()
        for r, s_ in enumerate(shards): s_.ExportShellsPacked(send[r]); s_.synchronize()
        offs = [np.concatenate([[0], np.cumsum(sz)]) for sz in sizes]
        recv = [torch.cat([send[o][int(offs[o][r]):int(offs[o][r + 1])] for o in range(n_shards)]) for r in range(n_shards)]
        torch.cuda.synchronize()
        for r, s_ in enumerate(shards): s_.ImportShellsPacked(recv[r]); s_.UpdateMeshesPlanned(); s_.DropGhostChunks()
        need = max(max(sz) for sz in sizes)
        stride = (need * 2 + 4096 + 15) // 16 * 16
        jobs, items, sent = max(p["jobs"] for p in plans), max(int(p["recv"][:, 0].sum()) for p in plans), max(p["send_items"] for p in plans)
        continue
    status = torch.zeros((n_shards, 8), dtype=torch.int32, device=dev)
    send = [torch.zeros((n_shards * stride,), dtype=torch.uint8, device=dev) for _ in range(n_shards)]
    torch.cuda.synchronize()
    for r, s_ in enumerate(shards):
        T("plan + export", lambda: s_.PlanShellsQueue(gathered.view(-1), n_shards, cap, stride, status[r], send[r], sent))
        s_.synchronize()  # (one shard at a time: kernels of different shards would run beside each other and stretch each other's trace entries)
    red = status.max(dim=0).values.contiguous()
    recv = [torch.cat([send[o][r * stride:(r + 1) * stride] for o in range(n_shards)]) for r in range(n_shards)]
    torch.cuda.synchronize()
    for r, s_ in enumerate(shards):
        T("import fixed", lambda: s_.ImportShellsFixed(recv[r], stride, red, jobs, items))
        T("mesh planned", lambda: s_.UpdateMeshesPlanned())
        T("drop", lambda: s_.DropGhostChunks())
        s_.synchronize()
    st = red.tolist()
    for s_ in shards:
        T("commit", lambda: s_.ShellCommit(st[0] != 0))
    ev = torch.cuda.Event(); ev.record()
    T("torch ev.record", lambda: ev.record(torch.cuda.current_stream()))
    T("map.record_event", lambda: shards[0].record_event(ev.cuda_event))
    T("torch wait_event", lambda: torch.cuda.current_stream().wait_event(ev))
    T("map.wait_event", lambda: shards[0].wait_event(ev.cuda_event))
    T("order_stream_after_map", lambda: shards[0].order_stream_after_map(torch.cuda.current_stream().cuda_stream))
    T("order_map_after_stream", lambda: shards[0].order_map_after_stream(torch.cuda.current_stream().cuda_stream))
    jobs, items, sent = st[3], st[4], st[5]
    keep = (gathered, status, send, recv, red)
print("status", st, "stride", stride)
for k, v in acc.items():
    print("%-18s median %7.1f us   min %7.1f   (n %d)" % (k, float(np.median(v)), min(v), len(v)))
