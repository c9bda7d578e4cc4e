#!/usr/bin/env python3
"""bench.py -- depth frames/s integrated into the TSDF (+ Mvoxel-updates/s), 640x480 @ 1 cm.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one depth(+colour) frame pushed through the hot path (pyramid -> cull -> resolve -> integrate kernels).
The default at 1 GPU is BASELINE.json's configuration 3: 640x480 depth + BGR colour, 1 cm voxels, a marching-cubes
recompute on every 10th frame (the reference's keyframe cadence, Chisel.cpp:54), 10 frames per call; `integration_only`
in the JSON line is the same stream without the recomputes.  Frames are synthetic (cvids_amd.synth: sphere room,
0.5 deg + 1 cm per frame) and already resident in HBM when the timed region starts.  Frames are handed to the library
--batch at a time (chisel_hip_integrate_batch: one launch set applies up to 16 frames to every voxel in frame order;
--batch 1 is the reference's frame-by-frame call pattern).  N > 1: one process per GPU, the chunk hash is sharded
spatially (chisel_hip_config.n_shards); the frames of a batch are ingested round-robin (frame j on rank j * N / batch)
and one RCCL all-gather per batch hands every rank the whole batch (cvids_amd.sharded.FrameExchange, on its own
stream, ordered against the map with events), then each rank integrates the chunks it owns -> total work is fixed:
"scaling": "strong".  N > 1 runs the same workload, the mesh recompute included (cvids_amd.sharded.ShardedChisel.UpdateMeshes:
every rank meshes the chunks it owns, neighbour chunks of other ranks arrive as ghosts).  Without WORLD_SIZE in the
environment `--gpus N` with N > 1 starts the N ranks itself (torch.distributed.run as a child process, before anything
touches a GPU) and relays rank 0's line.  --config picks one of BASELINE.json's configurations (3 is the default; 5 ends
with a garbage collection and a full mesh extraction inside the timed region).

The timed region (W warm-up steps, then exactly K steps between barrier + synchronize) is repeated --repeats times, each
time from an empty map, and `value` is K / the MEDIAN of those times (p10 / p90 beside it): the driver's default region is
20 frames = two launch sets = well under a millisecond, a single sample of that is at the mercy of one scheduling hiccup.

One JSON line on rank 0.  `roofline` prices the integration kernel: algorithmic bytes per frame
(DESIGN.md "Measurement": 16 B per integrated voxel + 8 B per carve probe + 8 B per carve + colour
bytes + the images once) divided by the kernel's mean duration from hipEvents recorded on the map's
stream during instrumented passes over the same frames from the same initial map state (median over the passes).
`roofline.traffic` / `hbm_frac_measured` come from the committed rocprofv3 PMC passes of exactly this command
(tools/profile.sh -> profiles/*_digest.json, matched on steps / warmup / frames per launch / meshing), null otherwise.
`cpu_baseline` = the oracle (oracle/liboracle.so: the reference algorithm restated, "faithful mode",
16 threads as the reference hard-codes) on a bounded sample of the same frames.
"""
import argparse
import datetime
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "depth frames/sec integrated + Mvoxel-updates/sec, 640x480 @ 1 cm, 1/2/4/8 GPU"
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--res", type=float, default=0.01)
    ap.add_argument("--chunk", type=int, default=16)
    ap.add_argument("--scene", default="sphere_room")
    ap.add_argument("--noise", action="store_true", help="depth noise 0.002 d^2 (SURVEY.md 8d)")
    ap.add_argument("--nan-fraction", type=float, default=0.0, help="share of the depth pixels that are NaN (SURVEY.md 8d: 0.02)")
    ap.add_argument("--no-color", action="store_true", help="depth-only path (IntegrateDepthScan)")
    ap.add_argument("--agents", type=int, default=1)
    ap.add_argument("--trunc-scale", type=float, default=None, help="InverseTruncator scale (default: 100*res)")
    ap.add_argument("--mesh-every", type=int, default=None,
                    help="marching-cubes recompute (UpdateMeshes) every M frames inside the timed region; default: 10 at 1 GPU (the "
                         "reference's keyframe cadence, Chisel.cpp:54 -- BASELINE config 3), 0 = off (N > 1: a sharded map is not meshed yet)")
    ap.add_argument("--shard-block", type=int, default=0,
                    help="N > 1 / --sim-shards: edge of the ownership blocks in chunks (chunk_owner; 0 = 8 between two ranks, where every edge "
                         "balances alike, else the library's 2).  Larger blocks: fewer shells cross between ranks when the map is meshed, coarser "
                         "balance of the integration from four ranks on (tools/shard_balance.py)")
    ap.add_argument("--mesh-checksum", action="store_true",
                    help="after the last pass: meshes, vertices and a checksum over every mesh array of the final map (summed over the ranks), as "
                         "`mesh_checksum` -- equal at every N when the sharded mesher is right (tests/test_gpu_bench.py)")
    ap.add_argument("--blocking-mesh", action="store_true",
                    help="N > 1: the sharded recompute in its blocking form (the host reads the plan's sizes in the middle of it) instead of the "
                         "wait-free one (cvids_amd/sharded.py: ShardedChisel._recompute_wait_free)")
    ap.add_argument("--batch", type=int, default=None,
                    help="frames per chisel_hip_integrate_batch call (<= 16 share one launch set); default: the keyframe interval when "
                         "meshing (10 frames = one launch set, the recompute falls exactly on every 10th frame), else 8")
    ap.add_argument("--repeats", type=int, default=9, help="timed passes (each from an empty map, with its own warm-up); `value` is the median")
    ap.add_argument("--max-chunks", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=30, help="frames of the same stream the CPU oracle is timed on (about 1 s each at 1 cm)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-pcie-leg", action="store_true", help="skip the extra pass over page-locked HOST depth frames (`pcie_inclusive`, 1 GPU only)")
    ap.add_argument("--host-frames", action="store_true", help="hand host (pageable) depth buffers to the library: the PCIe-inclusive rate (never `value`)")
    ap.add_argument("--pinned", action="store_true", help="with --host-frames: the host buffers are page-locked (what a capture pipeline would hand over)")
    ap.add_argument("--sim-shards", type=int, default=0, help="diagnostic, 1 GPU: integrate only the chunks of one shard of an N-way sharded map "
                                                               "(what one rank of an N-GPU run computes; every rank sees every frame)")
    ap.add_argument("--sim-rank", type=int, default=0)
    ap.add_argument("--exchange-color", action="store_true", default=None,
                    help="N > 1: every frame's colour image travels with its depth (a second all-gather per batch), as the caller delivers it "
                         "(ChiselServer.cpp:379-421); the default with --config 4 (its four agents deliver depth AND colour per frame).  Otherwise: depth "
                         "only, one static colour image resident on every rank")
    ap.add_argument("--no-exchange-color", dest="exchange_color", action="store_false", help="N > 1: depth only (the letter of BASELINE config 4: \"RCCL depth broadcast\")")
    ap.add_argument("--group", type=int, default=0, help="1 process: one map over N shards through the in-library group handle (chisel_hip_create_group: what a "
                                                          "single-process C++ caller like chisel_ros gets) -- on devices 0..N-1 when the node has them, else N shards on device 0")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the product path) | gloo (functional check of the N > 1 logic on one GPU)")
    ap.add_argument("--config", type=int, default=3, choices=(2, 3, 4, 5),
                    help="BASELINE.json configuration: 2 = depth only @ 2 cm; 3 = depth + colour @ 1 cm, meshes every 10th frame (the metric's); "
                         "4 = four interleaved agents; 5 = 1280x720 @ 0.5 cm, garbage collection + full mesh extraction at the end of the timed region")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="seconds after which a self-launched N > 1 run is killed")
    ap.add_argument("--no-e2e-leg", action="store_true", help="skip the end-to-end pass (per-frame depth AND colour from page-locked host buffers, `e2e`, 1 GPU only)")
    args = ap.parse_args()
    given = {a.split("=")[0] for a in sys.argv[1:] if a.startswith("--")}
    preset = {2: {"res": 0.02, "no_color": True, "mesh_every": 0, "batch": 16},  # (16 frames per call: at 2 cm a launch is as long as its frames' chain, the front half per launch the rest)
              3: {},
              4: {"agents": 4},
              5: {"width": 1280, "height": 720, "res": 0.005, "mesh_every": 0, "batch": 16, "max_chunks": 1 << 18}}[args.config]
    for k, v in preset.items():
        if "--" + k.replace("_", "-") not in given:
            setattr(args, k, v)
    if args.config == 5 and "--steps" not in given:
        args.steps, args.warmup = 48, 16
    if args.exchange_color is None:
        args.exchange_color = args.config == 4  # config 4's agents hand over depth and colour per frame: both travel (the N > 1 line says which)
    return args


def self_launch(args):
    """`python3 bench.py --gpus N` from a plain shell: this process touches no GPU; it starts the N ranks as children
    (python -m torch.distributed.run, rendezvous on 127.0.0.1), lets their output through (rank 0 prints the JSON line) and
    exits with their status.  Children that outlive --launch-timeout are killed (the whole process group) and the exit code is 124."""
    import signal
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    child = subprocess.Popen(cmd, env=env, start_new_session=True)

    def kill_tree():
        # the launcher puts every rank into a process group of its own: walk the tree (exact pids, never a pattern)
        victims = []
        try:
            import psutil
            victims = psutil.Process(child.pid).children(recursive=True)
        except Exception:
            pass
        for p in victims:
            try:
                p.send_signal(signal.SIGKILL)
            except Exception:
                pass
        try:
            os.killpg(child.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass

    try:
        rc = child.wait(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        print("bench.py: the %d-rank run did not finish within %.0f s: killing it" % (args.gpus, args.launch_timeout), file=sys.stderr)
        kill_tree()
        child.wait()
        rc = 124
    except KeyboardInterrupt:
        kill_tree()
        raise
    sys.exit(rc)


def algorithmic_bytes(cnt, n_frames, W, H, channels):
    """B = 16 N_sdf + 8 N_col + 4 N_colsat + 8 N_probe + 8 N_carved + (4 + C) W H per frame."""
    b = 16 * cnt["sdf"] + 8 * cnt["col"] + 4 * cnt["col_sat"] + 8 * cnt["probe"] + 8 * cnt["carved"]
    b += n_frames * (4 + channels) * W * H
    return b


def pmc_traffic(kernel_key, args, frames_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of THIS command
    (tools/profile.sh -> profiles/*_digest.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs).  FETCH_SIZE is doubled
    as MI355X_MICROARCH.md prescribes for gfx950 (it tallies 128-byte requests at 64 bytes).  A digest counts only if it
    was taken with the same steps, warm-up, frames per launch, meshing cadence, image and voxel size; (None, None) otherwise."""
    import glob
    want = {"steps": args.steps, "warmup": args.warmup, "frames_per_launch": frames_per_launch, "mesh_every": args.mesh_every,
            "image": "%dx%d" % (args.width, args.height), "voxel_m": args.res, "chunk": args.chunk, "color": not args.no_color}
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_digest.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        # the kernel has one instantiation per granularity (..., 2> / ..., 4>: voxels per lane, picked per launch): all of them count
        ks = [v for n, v in d.get("kernels", {}).items() if n.startswith(kernel_key) and "fetch_kib" in v and "write_kib" in v]
        if ks and all(d.get("bench", {}).get(key) == val for key, val in want.items()):
            calls = sum(v.get("calls", 1) for v in ks) or 1
            best = (f, {"fetch_kib": sum(v["fetch_kib"] * v.get("calls", 1) for v in ks) / calls,
                        "write_kib": sum(v["write_kib"] * v.get("calls", 1) for v in ks) / calls})
    if not best:
        return None, None
    f, k = best
    return (2.0 * k["fetch_kib"] + k["write_kib"]) * 1024.0, os.path.relpath(f, ROOT)


def cpu_baseline(args, frames, color_img, intr, scale):
    import oracle
    om = oracle.OracleMap(args.chunk, args.res, not args.no_color, threads=16)
    om.set_integrator(oracle.TRUNC_INVERSE, scale, 1.0, True, 0.05)
    n = min(args.cpu_frames, len(frames))
    upd = 0
    t0 = time.perf_counter()
    for depth, pose in frames[:n]:
        if args.no_color:
            om.integrate_depth(depth, pose, intr, 0.05, 5.0)
        else:
            om.integrate_depth_color(depth, pose, intr, color_img, near=0.05, far=5.0)
        c = om.counters()
        upd += c["sdf"] + c["carved"]
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": 16 if not args.no_color else 1, "kind": "port",
            "host_cpus": os.cpu_count(), "mvoxel_updates_per_s": upd / dt / 1e6,
            "sample": "the first %d of the timed frames, integrated into an EMPTY oracle map (the GPU's timed region starts from the map the "
                      "warm-up frames left; the oracle's cost per frame is set by the %d candidate chunks it visits, not by the map's state); "
                      "oracle faithful mode, %s; %.1f s"
                      % (n, c["candidates"], "16 std::threads as Chisel.h:150" if not args.no_color else "serial as Chisel.h:71", dt)}


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)  # does not return
    import torch
    import torch.distributed as dist
    from cvids_amd import synth
    from cvids_amd.chisel import Chisel, ConstantWeighter, InverseTruncator, PinholeCamera, ProjectionIntegrator

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if args.dist_backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)  # functional check: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            # (a collective that does not complete aborts the run after three minutes instead of the default ten)
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=180))
        else:
            dist.init_process_group(args.dist_backend, timeout=datetime.timedelta(seconds=180))

    if args.mesh_every is None:
        args.mesh_every = 10  # the reference's keyframe cadence (Chisel.cpp:54), at every N
    if args.batch is None:
        # 1 GPU: the keyframe interval; N GPUs: 16 frames per all-gather (every rank contributes 16 / N frame slots)
        args.batch = args.mesh_every if 0 < args.mesh_every <= 16 else (16 if world > 1 else 8)
    W, H = args.width, args.height
    intr = synth.intrinsics(W, H)
    cam = PinholeCamera(*intr, W, H, 0.05, 5.0)
    scale = args.trunc_scale if args.trunc_scale is not None else 100.0 * args.res  # 2.0 @2 cm, 1.0 @1 cm, 0.5 @0.5 cm
    integ = ProjectionIntegrator(InverseTruncator(scale), ConstantWeighter(1.0), 0.05, True)
    use_color = not args.no_color
    channels = 3 if use_color else 0
    total = args.warmup + args.steps
    n_traj = (total + args.agents - 1) // args.agents
    frames = list(synth.stream(args.scene, n_traj, W, H, agents=args.agents, noise=args.noise, nan_fraction=args.nan_fraction))[:total]
    color_img = synth.render_color(W, H, 3) if use_color else None

    # frames are handed over a batch at a time; with N ranks frame j of a batch is ingested by rank j * N / K and its
    # pixels live in that rank's HBM before timing starts
    import ctypes as C
    from cvids_amd import capi
    from cvids_amd.chisel import color_frame, depth_frame
    from cvids_amd.sharded import FrameExchange, PipelinedExchange, ShardedChisel, frames_of_rank, pack_meta
    K = max(1, args.batch)
    if world > 1 and K % world:
        K = max(world, (K // world) * world)  # every rank contributes K / world frame slots to each all-gather
    bounds = [(lo, min(lo + K, args.warmup)) for lo in range(0, args.warmup, K)] + \
             [(lo, min(lo + K, total)) for lo in range(args.warmup, total, K)]
    c_dev = torch.from_numpy(color_img).to(dev) if use_color else None  # static colour pattern, resident on every rank
    xcolor = bool(args.exchange_color and use_color and world > 1)
    xch = FrameExchange(W, H, K, dev, dist, channels=3 if xcolor else 0) if world > 1 else None
    if world == 1:
        stack = [torch.from_numpy(np.stack([frames[i][0] for i in range(lo, hi)])).to(dev) for lo, hi in bounds]
    else:
        # a short (last) batch leaves the trailing slots empty: the collective always moves K slots, only hi - lo are integrated
        mine = frames_of_rank(K, world, rank)
        blank = np.zeros((H, W), np.float32)
        stack = [torch.from_numpy(np.stack([frames[lo + j][0] if lo + j < hi else blank for j in mine])).to(dev) for lo, hi in bounds]
        meta = [torch.from_numpy(np.stack([pack_meta(frames[min(lo + j, hi - 1)][1], cam) for j in mine])).to(dev) for lo, hi in bounds]
        color_slots = c_dev.unsqueeze(0).repeat(len(mine), 1, 1, 1).contiguous() if xcolor else None  # this rank's frames' colour images

    host_src = None
    if args.host_frames and world == 1:
        if args.pinned:
            pins = [torch.from_numpy(f[0]).pin_memory() for f in frames]
            keep_pins = pins
            host_src = [p.numpy() for p in pins]  # same page-locked memory, handed over as a host pointer
        else:
            host_src = [f[0] for f in frames]
    # the C structs of every batch are built once, outside the timed region (device addresses are fixed)
    keep = []
    calls = []
    for b, (lo, hi) in enumerate(bounds):
        n = hi - lo
        src = stack[b] if world == 1 else xch.depth_view(b & 1)
        fa = (capi.DepthFrame * n)()
        ca = (capi.ColorFrame * n)() if use_color else None
        for j in range(n):
            fa[j], k1 = depth_frame(host_src[lo + j] if (args.host_frames and world == 1) else src[j], frames[lo + j][1], cam)
            keep.append(k1)
            if use_color:
                ca[j], k2 = color_frame(xch.color[b & 1][j] if xcolor else c_dev, frames[lo + j][1], cam)
                keep.append(k2)
        calls.append((n, fa, ca))
    first_timed = next(b for b, (lo, hi) in enumerate(bounds) if lo >= args.warmup) if args.steps else len(bounds)
    calls_ref = [calls]

    group_devices = None
    if args.group and world == 1:
        group_devices = list(range(args.group)) if torch.cuda.device_count() >= args.group else [local_rank] * args.group

    def new_map():
        if group_devices:
            m = Chisel((args.chunk,) * 3, args.res, use_color, max_chunks=args.max_chunks, devices=group_devices)
        else:
            ns = args.sim_shards if (args.sim_shards and world == 1) else world
            # ownership blocks: between TWO ranks the owner is the parity of bx + by + bz -- a 3-D checkerboard that deals every launch set
            # evenly whatever the blocks' edge (tools/shard_balance.py: max / mean 1.00), so the largest edge measured is taken there: a
            # quarter of the shells cross between the ranks per recompute.  From four ranks on larger blocks cost more than they save.
            blk = args.shard_block or (8 if ns == 2 else 0)
            m = Chisel((args.chunk,) * 3, args.res, use_color, device_id=local_rank, max_chunks=args.max_chunks,
                       n_shards=ns,
                       shard_rank=(args.sim_rank % args.sim_shards) if (args.sim_shards and world == 1) else rank, shard_block=blk)
        m._use(integ)
        m.px = PipelinedExchange(xch, m) if world > 1 else None  # RCCL -> integrate ordering: events, no host wait
        m.sharded = ShardedChisel(m, xch, integ) if world > 1 else None  # Chisel::UpdateMeshes of the sharded map
        return m

    mesh_stats = {"recomputes": 0, "ghost_bytes": 0}

    def update_meshes(m, ids=None):
        mesh_stats["recomputes"] += 1
        if world > 1:
            mesh_stats["ghost_bytes"] += m.sharded.UpdateMeshes(force=True, ids=ids, wait_free=not args.blocking_mesh) or 0
        elif ids is None:
            m.UpdateMeshes(force=True)
        else:
            m.UpdateMeshesOf(ids)

    def finish_config5(m):
        """BASELINE config 5's "chunk GC + full mesh extraction": the chunks that lie behind the camera of the last frame (centre's
        camera z < 0: the map forgets what it has turned away from) are garbage-collected (Chisel::GarbageCollect, Chisel.cpp:61-67),
        then every remaining chunk is meshed.  Each rank handles the chunks it owns."""
        pose = np.asarray(frames[total - 1][1], np.float64)
        ids = np.asarray(m.GetChunkIDs(), np.int64).reshape(-1, 3)
        if len(ids):
            centre = (ids + 0.5) * (args.chunk * args.res)
            z_cam = (centre - pose[:3, 3]) @ pose[:3, 2]
            behind = ids[z_cam < 0.0]
            if len(behind):
                m.GarbageCollect(behind.astype(np.int32))
            ids = ids[z_cam >= 0.0]
        update_meshes(m, ids.astype(np.int32))
        return len(ids)

    def run(m, b_lo, b_hi):
        L, h = m.L, m.h
        for b in range(b_lo, b_hi):
            n, fa, ca = calls_ref[0][b]
            if world > 1:
                m.px.exchange(b, stack[b], meta[b], color_slots)  # RCCL all-gather on the communication stream; the map waits for its event
            if world > 1:
                m.sharded.Settle()  # a wait-free recompute in flight: its status (by now on the host) before the map changes again
            rc = L.chisel_hip_integrate_batch(h, n, fa, ca)
            if rc:
                capi.check(rc)
            if world > 1:
                m.px.consumed(b)
            if args.mesh_every and (bounds[b][1] // args.mesh_every) > (bounds[b][0] // args.mesh_every):
                update_meshes(m)
        if args.config == 5 and b_hi == len(bounds) and b_hi > b_lo:
            finish_config5(m)
        if world > 1:
            m.sharded.Settle()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(m, instrumented=False):
        """W warm-up steps, then exactly K steps between fences; returns (seconds, host issue seconds, counters, chunks, profile)"""
        run(m, 0, first_timed)
        m.synchronize()
        m.counters(reset=True)
        m.launch_stats(reset=True)
        if instrumented:
            m.set_profiling(True)
            if m.px is not None:
                m.px.measure(True)
        fence()
        t0 = time.perf_counter()
        run(m, first_timed, len(bounds))
        t_issue = time.perf_counter() - t0  # host time to enqueue everything (no synchronisation yet)
        fence()
        m.synchronize()  # inside the timed region: a launch queued behind a recompute that did not fit is replayed by this call (normally the map's stream is idle by now: a few microseconds); it also surfaces pool exhaustion
        dt = time.perf_counter() - t0
        prof = m.profile(reset=True) if instrumented else None
        if instrumented and m.px is not None:
            prof["allgather_us"] = m.px.measure(False)  # per batch, events on the communication stream
        cnt = m.counters()
        cnt["launch_shapes"] = {k: v for k, v in m.launch_stats().items() if v}  # which shapes the launch heuristics picked in the timed region
        if instrumented:
            m.set_profiling(False)
        n_chunks = m.NumChunks()
        t_all = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        return float(t_all.item()), t_issue, cnt, n_chunks, prof

    def median(v):
        v = sorted(v)
        return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])

    def pct(v, q):
        v = sorted(v)
        return v[min(len(v) - 1, max(0, int(round(q * (len(v) - 1)))))]

    # A process that has just started sees the GPU at idle clocks, and the first pass of a cold process is up to 15 % slower
    # than the steady state the metric is about: a throw-away pass over the whole stream comes first.  Then, every pass from
    # an EMPTY map with its own W warm-up frames: the integration-only passes (C), the timed passes (A: `value`), the
    # instrumented passes (B: hipEvents around every kernel), the PCIe-inclusive pass (D).  One map serves all of them
    # (chisel_hip_reset between passes: the pool is not re-allocated).
    repeats = max(1, args.repeats)
    m = new_map()
    t_warm, warm_passes = time.perf_counter(), 0
    while True:  # at least one pass, and at least a second of work: a fresh box needs that long to reach its steady clocks
        run(m, 0, len(bounds))
        m.synchronize()
        warm_passes += 1
        # (N > 1: a fixed number of passes -- every rank must take the same turn, their clocks do not agree)
        if (world == 1 and time.perf_counter() - t_warm > 1.0) or (world > 1 and warm_passes >= 3):
            break
        (m.sharded or m).Reset()
    fence()
    # ---- passes C (1 GPU, meshing on): the same stream without the mesh recomputes, for reference ----------------
    no_mesh = None
    if args.mesh_every and world == 1 and not args.no_roofline:
        every, args.mesh_every = args.mesh_every, 0
        ts = []
        for _ in range(min(repeats, 5)):
            (m.sharded or m).Reset()
            ts.append(timed(m)[0])
        args.mesh_every = every
        no_mesh = {"value": args.steps / median(ts), "unit": "frames/s", "ms_per_step": median(ts) / args.steps * 1e3, "repeats": len(ts)}

    # ---- passes A: the timed region ---------------------------------------------------------------------
    ts, issues = [], []
    for _ in range(repeats):
        (m.sharded or m).Reset()
        dt_r, t_issue_r, cnt, n_chunks, _ = timed(m)
        ts.append(dt_r)
        issues.append(t_issue_r)
    dt, t_issue = median(ts), median(issues)
    vals = torch.tensor([cnt["sdf"] + cnt["carved"], cnt["sdf"], cnt["col"], cnt["col_sat"], cnt["probe"], cnt["carved"],
                         cnt["work_chunks"], n_chunks], dtype=torch.float64, device=dev)
    per_rank_sdf = None
    if world > 1:
        mine_sdf = torch.tensor([float(cnt["sdf"])], dtype=torch.float64, device=dev)
        all_sdf = [torch.zeros_like(mine_sdf) for _ in range(world)]
        dist.all_gather(all_sdf, mine_sdf)
        per_rank_sdf = [float(t.item()) for t in all_sdf]
        dist.all_reduce(vals, op=dist.ReduceOp.SUM)
    vals = [float(v) for v in vals.tolist()]

    # ---- passes B: same frames from the same initial state, hipEvents around every kernel ---------------------
    roof = None
    if not args.no_roofline:
        avgs, others, dts_b = [], [], []
        for _ in range(min(repeats, 5)):
            (m.sharded or m).Reset()
            dt_b, _, cb, _, prof = timed(m, instrumented=True)
            k = prof["integrate"]
            avgs.append(k["ms"] / max(k["launches"], 1))
            others.append({n: (prof[n]["ms"] / max(prof[n]["launches"], 1)) * 1e3 for n in ("pyramid", "cull", "resolve", "mesh") if prof[n]["launches"]})
            if prof.get("allgather_us"):
                others[-1]["allgather"] = median(prof["allgather_us"])
            dts_b.append(dt_b)
        avg_ms = median(avgs)
        launches = k["launches"]
        bytes_per_launch = algorithmic_bytes(cb, args.steps, W, H, channels) / max(launches, 1)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        kernel_key = "integrate_kernel<%d, %s, %s" % (args.chunk, "true" if use_color else "false", "true" if use_color else "false")
        traffic, traffic_src = pmc_traffic(kernel_key, args, args.steps / max(launches, 1)) if world == 1 else (None, None)
        roof = {"bound": "hbm", "kernel": kernel_key + ", 2|4>",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_src,
                "hbm_frac_measured": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and avg_ms > 0) else None,
                "avg_kernel_us": avg_ms * 1e3, "avg_kernel_us_min_max": [min(avgs) * 1e3, max(avgs) * 1e3], "instrumented_passes": len(avgs),
                "launches": launches, "frames_per_launch": args.steps / max(launches, 1),
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "other_kernels_us": {n: median([o[n] for o in others if n in o]) for n in others[0]},
                "instrumented_ms_per_step": median(dts_b) / args.steps * 1e3,
                "note": "rank 0 shard; algorithmic bytes count every frame's voxel updates (16 B each) although a launch moves a voxel once: "
                        "`frac` can exceed `hbm_frac_measured`; traffic (PMC FETCH_SIZE/WRITE_SIZE) is collected by tools/profile.sh into profiles/"}

    # ---- pass D (1 GPU): the same stream handed over as page-locked HOST depth frames (what a capture pipeline holds): the
    # PCIe-inclusive rate.  Never `value`.
    pcie = None
    if world == 1 and not args.no_pcie_leg and not args.host_frames and not args.no_roofline:
        pins = [torch.from_numpy(f[0]).pin_memory() for f in frames]
        saved = calls
        calls_d = []
        for b, (lo, hi) in enumerate(bounds):
            n = hi - lo
            fa = (capi.DepthFrame * n)()
            for j in range(n):
                fa[j], k1 = depth_frame(pins[lo + j].numpy(), frames[lo + j][1], cam)
                keep.append(k1)
            calls_d.append((n, fa, saved[b][2]))
        calls_ref[0] = calls_d
        ts_d = []
        for _ in range(min(repeats, 3)):
            (m.sharded or m).Reset()
            ts_d.append(timed(m)[0])
        calls_ref[0] = saved
        pcie = {"value": args.steps / median(ts_d), "unit": "frames/s", "ms_per_step": median(ts_d) / args.steps * 1e3,
                "source": "page-locked host depth frames read over PCIe during the call (colour image resident)", "repeats": len(ts_d)}
    # ---- pass E (1 GPU): end to end as SURVEY.md 8(d) defines it -- every frame's depth AND colour image start in page-locked host
    # memory (a new colour image per frame, as the caller delivers it: ChiselServer.cpp:379-421), are copied to one of two device
    # buffer sets on a copy stream while the previous batch is integrated (events both ways, no host wait), then integrated.
    e2e = None
    if world == 1 and not args.no_e2e_leg and not args.host_frames and not args.no_roofline:
        nb = len(bounds)
        kmax = max(hi - lo for lo, hi in bounds)
        h_depth = [torch.from_numpy(np.stack([frames[i][0] for i in range(lo, hi)])).pin_memory() for lo, hi in bounds]
        h_color = None
        if use_color:
            uu, vv = np.meshgrid(np.arange(W), np.arange(H))
            pat = lambda k: np.stack([(uu + k) % 256, vv % 256, (uu + vv + k) % 256], axis=-1).astype(np.uint8)  # (u + k, v, u + v + k) mod 256
            h_color = [torch.from_numpy(np.stack([pat(i) for i in range(lo, hi)])).pin_memory() for lo, hi in bounds]
        d_depth = [torch.empty((kmax, H, W), dtype=torch.float32, device=dev) for _ in range(2)]
        d_color = [torch.empty((kmax, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(2)] if use_color else None
        calls_e = []
        for b, (lo, hi) in enumerate(bounds):
            n = hi - lo
            fa = (capi.DepthFrame * n)()
            ca = (capi.ColorFrame * n)() if use_color else None
            for j in range(n):
                fa[j], k1 = depth_frame(d_depth[b & 1][j], frames[lo + j][1], cam)
                keep.append(k1)
                if use_color:
                    ca[j], k2 = color_frame(d_color[b & 1][j], frames[lo + j][1], cam)
                    keep.append(k2)
            calls_e.append((n, fa, ca))
        copy_stream = torch.cuda.Stream(device=dev)
        ready = [torch.cuda.Event() for _ in range(2)]
        free = [torch.cuda.Event() for _ in range(2)]
        for e in ready + free:
            e.record(copy_stream)

        def run_e2e(m, b_lo, b_hi):
            for b in range(b_lo, b_hi):
                n, fa, ca = calls_e[b]
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(free[b & 1])  # the batch that last used this buffer set has been integrated
                    d_depth[b & 1][:n].copy_(h_depth[b], non_blocking=True)
                    if use_color:
                        d_color[b & 1][:n].copy_(h_color[b], non_blocking=True)
                    ready[b & 1].record(copy_stream)
                m.wait_event(ready[b & 1].cuda_event)
                rc = m.L.chisel_hip_integrate_batch(m.h, n, fa, ca)
                if rc:
                    capi.check(rc)
                m.record_event(free[b & 1].cuda_event)
                if args.mesh_every and (bounds[b][1] // args.mesh_every) > (bounds[b][0] // args.mesh_every):
                    update_meshes(m)

        ts_e = []
        for _ in range(min(repeats, 3)):
            (m.sharded or m).Reset()
            run_e2e(m, 0, first_timed)
            m.synchronize()
            fence()
            t0 = time.perf_counter()
            run_e2e(m, first_timed, nb)
            fence()
            ts_e.append(time.perf_counter() - t0)
            m.synchronize()
        # the bus rate this box gives a plain page-locked copy (64 MiB, best of 5)
        big_h = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
        big_d = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            big_d.copy_(big_h, non_blocking=True)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        frame_bytes = W * H * (4 + (3 if use_color else 0))
        e2e = {"value": args.steps / median(ts_e), "unit": "frames/s", "ms_per_step": median(ts_e) / args.steps * 1e3, "repeats": len(ts_e),
               "host_bytes_per_frame": frame_bytes, "h2d_gb_per_s": args.steps * frame_bytes / median(ts_e) / 1e9,
               "pcie_copy_gb_per_s_measured": (64 << 20) / best / 1e9,
               "source": "every frame's depth and colour image ((u + k, v, u + v + k) mod 256) copied from page-locked host memory into one of two "
                         "device buffer sets on a copy stream, overlapped with the previous batch's integration (events, no host wait)"}
    mesh_checksum = None
    if args.mesh_checksum:
        import zlib
        if world > 1:
            m.sharded.Settle()
        acc = np.zeros(3, np.int64)
        for cid in map(tuple, np.asarray(m.GetMeshIDs()).reshape(-1, 3).tolist()):  # (a rank lists the meshes of the chunks it owns)
            mesh = m.GetMesh(cid)
            h = zlib.crc32(np.asarray(cid, np.int32).tobytes())
            for key in ("vertices", "normals", "colors", "grids"):
                if mesh[key] is not None:
                    h = zlib.crc32(np.ascontiguousarray(mesh[key]).tobytes(), h)
            acc += np.array([1, len(mesh["vertices"]), h], np.int64)
        t_sum = torch.from_numpy(acc).to(dev if (world > 1 and dist.get_backend() == "nccl") else torch.device("cpu"))
        if world > 1:
            dist.all_reduce(t_sum)
        mesh_checksum = {"meshes": int(t_sum[0]), "vertices": int(t_sum[1]), "crc_sum": int(t_sum[2])}
    sharded_totals = (getattr(m.sharded, "shell_bytes", 0), getattr(m.sharded, "whole_chunk_bytes", 0)) if world > 1 else (0, 0)
    wait_free_totals = (getattr(m.sharded, "wait_free_recomputes", 0), getattr(m.sharded, "wait_free_aborts", 0), getattr(m.sharded, "wire_bytes", 0), {str(k): v for k, v in getattr(m.sharded, "abort_bits", {}).items()}, getattr(m.sharded, "last_abort_status", None), getattr(m.sharded, "_est", None)) if world > 1 else (0, 0, 0, {}, None, None)
    m.close()

    if rank == 0:
        out = {
            "metric": METRIC, "value": args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic" + ((" (host depth buffers, %s: PCIe-inclusive)" % ("page-locked" if args.pinned else "pageable")) if args.host_frames else ""),
            "mvoxel_updates_per_s": vals[0] / dt / 1e6, "host_issue_ms_per_step": t_issue / args.steps * 1e3,
            "repeats": repeats, "value_p10_p90": [args.steps / pct(ts, 0.9), args.steps / pct(ts, 0.1)],
            "ms_per_step_min_max": [min(ts) / args.steps * 1e3, max(ts) / args.steps * 1e3],
            "config": {"workload": "%dx%d %s stream (%s, %d agent%s), %g cm voxels, chunk %d^3, InverseTruncator(%g), carving 0.05 m, "
                                   "projective TSDF integration%s" % (W, H, "depth+BGR colour" if use_color else "depth-only",
                                                                        args.scene, args.agents, "" if args.agents == 1 else "s",
                                                                        args.res * 100, args.chunk, scale,
                                                                        (" + marching cubes every %d frames" % args.mesh_every) if args.mesh_every else ""),
                       "baseline_config": args.config, "image": "%dx%d" % (W, H), "voxel_m": args.res, "chunk": args.chunk, "color": use_color, "frames_per_call": K, "mesh_every": args.mesh_every,
                       "parallelism": "spatial chunk-hash shards x%d, RCCL all-gather of each %d-frame batch" % (world, K) if world > 1 else
                                      ("1 process, in-library group of %d shards on devices %s (one issuing host thread per shard)" % (args.group, group_devices) if group_devices else "1 GPU")},
            "per_frame": {"voxel_updates": vals[0] / args.steps, "n_sdf": vals[1] / args.steps, "n_col": vals[2] / args.steps,
                          "n_probe": vals[4] / args.steps, "n_carved": vals[5] / args.steps,
                          "work_chunks": vals[6] / args.steps, "resident_chunks_end": vals[7]},
            "launch_shapes": cnt.get("launch_shapes"),
        }
        if mesh_checksum is not None:
            out["mesh_checksum"] = mesh_checksum
        if args.noise or args.nan_fraction:
            out["config"]["workload"] += " (depth noise 0.002 d^2: %s, NaN pixels: %g)" % ("yes" if args.noise else "no", args.nan_fraction)
        if world > 1:
            out["sharded_meshing"] = {"recomputes": mesh_stats["recomputes"],
                                      "ghost_bytes_per_recompute_rank0": mesh_stats["ghost_bytes"] / max(1, mesh_stats["recomputes"]),
                                      "whole_ghost_chunks_would_be": sharded_totals[1] / max(1, mesh_stats["recomputes"]),
                                      "shells_over_whole_chunks": sharded_totals[0] / max(1, sharded_totals[1]),
                                      # (of the last timed pass) recomputes queued without a host wait, how many of them were called off on the
                                      # device and made again the blocking way, and the fixed-size segments they sent per recompute
                                      "wait_free": {"recomputes": wait_free_totals[0], "called_off": wait_free_totals[1], "made_again_wait_free": getattr(m.sharded, "wait_free_retries", 0), "called_off_by_status_bits": wait_free_totals[3], "last_called_off_status": wait_free_totals[4], "sizes_now": wait_free_totals[5],
                                                    "wire_bytes_per_recompute_rank0": wait_free_totals[2] / max(1, wait_free_totals[0])}}
            if getattr(m.sharded, "phase_us", None):  # CHISEL_HIP_HOST_TIMING=1: rank 0's host time per sharded recompute, by phase
                n_rc = max(1, m.sharded.phase_us.get("recomputes", 1))
                out["sharded_meshing"]["host_us_per_recompute_rank0"] = {k: round(v / n_rc, 1) for k, v in m.sharded.phase_us.items() if k != "recomputes"}
            out["load_balance"] = {"n_sdf_per_rank_max_over_mean": max(per_rank_sdf) / max(1e-9, sum(per_rank_sdf) / world), "n_sdf_per_rank": per_rank_sdf}
            out["exchange"] = {"payload": "depth + colour" if xcolor else "depth only (one static colour image resident on every rank)",
                               "bytes_per_batch": K * (W * H * 4 + 128 + (W * H * 3 if xcolor else 0)),
                               "allgather_us_per_batch_rank0": (roof or {}).get("other_kernels_us", {}).get("allgather"),
                               "hardware_curve": "none yet: no run of this path on two or more physical GPUs exists (tools/first_contact.sh is the list of first commands)"}
            out["note"] = ("N > 1: one map sharded over the ranks (total work fixed); every batch is all-gathered, every rank integrates the chunks it "
                           "owns and meshes them with ghost copies of the neighbours other ranks own")
        if roof:
            out["roofline"] = roof
        if no_mesh:
            out["integration_only"] = no_mesh
        if pcie:
            out["pcie_inclusive"] = pcie
        if e2e:
            out["e2e"] = e2e
        if args.config == 5:
            out["config"]["end_of_region"] = "garbage collection of the chunks behind the last camera + mesh extraction of every remaining chunk (inside the timed region)"
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 16, "kind": "port",
                                       "sample": "infeasible as specified: the reference allocates every chunk of the frustum's bounding box per frame, "
                                                 "about 250 k chunks = 33 GB at 1280x720 / 0.5 cm / 5 m (BASELINE.md section 2)"}
        elif not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, frames[args.warmup:], color_img, intr, scale)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
