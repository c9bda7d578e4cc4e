"""Spatially sharded TSDF map: one process per GPU, every process owns the chunks `chunk_owner(id) == rank`.

Voxels have exactly one owner and an update depends only on (the voxel, the frame, the pose), so the shards never
exchange voxel data and no reduction exists on this path.  The only exchange step is the frame
payload: every rank needs every depth(+colour) frame.  Frames are ingested round-robin -- frame j of a batch enters
on rank j * world / K -- and one `all_gather` per batch hands every rank the whole batch: with K a multiple of the
world size each GPU sends K/world frames over all of its xGMI links at once instead of one root pushing K frames
through its own links (RCCL `ncclAllGather`; `gloo` in the CPU tests).  Poses / intrinsics travel the same way as
a small float tensor.

Nothing here computes on voxels: integration is `Chisel.IntegrateBatch` of the local shard (libchisel_hip.so).
The local map is injected (`local_map`) so the distribution logic is testable on CPU with a recording stand-in.
"""
import numpy as np

META_FLOATS = 20  # pose 12 + fx, fy, cx, cy + near, far + 2 spare


def pack_meta(pose, camera):
    m = np.zeros(META_FLOATS, np.float32)
    m[:12] = np.asarray(pose, np.float32)[:3, :4].reshape(12)
    m[12:18] = (camera.fx, camera.fy, camera.cx, camera.cy, camera.near_plane, camera.far_plane)
    return m


def unpack_meta(m, width, height):
    from .chisel import PinholeCamera
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :4] = np.asarray(m[:12], np.float32).reshape(3, 4)
    cam = PinholeCamera(float(m[12]), float(m[13]), float(m[14]), float(m[15]), width, height, float(m[16]), float(m[17]))
    return pose, cam


def frames_of_rank(n_frames, world, rank):
    """indices (within a batch of n_frames) that `rank` ingests: contiguous blocks, so that all_gather output order is frame order"""
    if n_frames % world:
        raise ValueError("batch of %d frames does not split over %d ranks" % (n_frames, world))
    per = n_frames // world
    return list(range(rank * per, (rank + 1) * per))


class FrameExchange:
    """all_gather of one batch: every rank contributes the frames it ingested, every rank receives all of them in order.

    One collective per batch: a frame travels as one fp32 row [H*W depth | META_PAD metadata] (colour, when exchanged,
    as a second uint8 collective)."""

    META_PAD = 32  # floats per frame after the pixels: keeps every frame's depth 16-byte aligned

    def __init__(self, width, height, batch, device, dist_module=None, channels=0):
        import torch
        self.torch = torch
        self.dist = dist_module
        self.device = torch.device(device)
        self.world = dist_module.get_world_size() if dist_module is not None else 1
        self.rank = dist_module.get_rank() if dist_module is not None else 0
        if batch % self.world:
            raise ValueError("batch %d must be a multiple of the world size %d" % (batch, self.world))
        self.batch, self.per = batch, batch // self.world
        self.W, self.H, self.C = width, height, channels
        self.row = width * height + self.META_PAD
        # double-buffered on both sides: the kernels of batch b may still read buffer b & 1 while batch b+1 arrives
        self.recv = [torch.zeros((batch, self.row), dtype=torch.float32, device=device) for _ in range(2)]
        self.send = [torch.zeros((self.per, self.row), dtype=torch.float32, device=device) for _ in range(2)]
        self.color = [torch.empty((batch, height, width, channels), dtype=torch.uint8, device=device) for _ in range(2)] if channels else None
        self.turn = 0

    def depth_view(self, b):
        return self.recv[b][:, :self.W * self.H].unflatten(1, (self.H, self.W))

    def meta_view(self, b):
        return self.recv[b][:, self.W * self.H:self.W * self.H + META_FLOATS]

    def exchange(self, local_depth, local_meta, local_color=None, buffer=None):
        """local_*: this rank's `per` frame slots ([per, H, W] float32, [per, META_FLOATS], [per, H, W, C] uint8).
        `buffer` (0 / 1) selects the receive buffer; by default the two alternate."""
        b = (self.turn & 1) if buffer is None else (buffer & 1)
        self.turn += 1
        s = self.send[b]
        s[:, :self.W * self.H].copy_(local_depth.reshape(self.per, -1), non_blocking=True)
        s[:, self.W * self.H:self.W * self.H + META_FLOATS].copy_(local_meta, non_blocking=True)
        if self.world == 1:
            self.recv[b].copy_(s, non_blocking=True)
            if self.C:
                self.color[b].copy_(local_color, non_blocking=True)
        elif self.dist.get_backend() == "gloo" and s.is_cuda:
            # functional check only (gloo has no device all-gather): bounce through the host
            host = self.torch.empty(self.recv[b].shape, dtype=self.torch.float32)
            self.dist.all_gather_into_tensor(host.view(-1), s.cpu().view(-1))
            self.recv[b].copy_(host)
            if self.C:
                hc = self.torch.empty(self.color[b].shape, dtype=self.torch.uint8)
                self.dist.all_gather_into_tensor(hc.view(-1), local_color.contiguous().cpu().view(-1))
                self.color[b].copy_(hc)
        else:
            self.dist.all_gather_into_tensor(self.recv[b].view(-1), s.view(-1))
            if self.C:
                self.dist.all_gather_into_tensor(self.color[b].view(-1), local_color.contiguous().view(-1))
        return self.depth_view(b), self.meta_view(b), (self.color[b] if self.C else None)


class PipelinedExchange:
    """FrameExchange on its own stream, ordered against the map with events only (no host synchronisation):
    the all-gather of batch b+1 runs while batch b is being integrated.

        px = PipelinedExchange(xch, local_map)
        for b, (local_depth, local_meta) in enumerate(batches):
            depth, meta, _ = px.exchange(b, local_depth, local_meta)   # queued on the communication stream
            local_map.IntegrateBatch(...)                               # waits (on the device) for that all-gather
            px.consumed(b)                                              # buffer b & 1 may be refilled after this batch

    Buffer b & 1 is reused by batch b + 2: its all-gather waits for the event recorded by consumed(b).
    """

    def __init__(self, exchange, local_map):
        torch = exchange.torch
        self.x, self.map, self.torch = exchange, local_map, torch
        dev = exchange.recv[0].device
        self.comm = torch.cuda.Stream(device=dev)
        self.ready = [torch.cuda.Event() for _ in range(2)]
        self.free = [torch.cuda.Event() for _ in range(2)]
        for e in self.ready + self.free:  # a first record creates the hipEvent_t behind the torch object
            e.record(self.comm)
        self.timing = False   # measure(): pairs of timing events around every collective, on the communication stream
        self.timed = []

    def exchange(self, b, local_depth, local_meta, local_color=None):
        torch = self.torch
        self.comm.wait_stream(torch.cuda.current_stream())  # the local slots were produced on the caller's stream
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(self.free[b & 1])
            if self.timing:
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record(self.comm)
            out = self.x.exchange(local_depth, local_meta, local_color, buffer=b & 1)
            if self.timing:
                t1.record(self.comm)
                self.timed.append((t0, t1))
            self.ready[b & 1].record(self.comm)
        self.map.wait_event(self.ready[b & 1].cuda_event)
        return out

    def measure(self, on):
        """switch the timing of the collectives on / off; switching off returns their durations in microseconds (after a device sync)"""
        self.timing = bool(on)
        if on:
            self.timed = []
            return None
        self.torch.cuda.synchronize()
        us = [a.elapsed_time(b) * 1e3 for a, b in self.timed]
        self.timed = []
        return us

    def consumed(self, b):
        self.map.record_event(self.free[b & 1].cuda_event)


# ---- meshing a sharded map -----------------------------------------------------------------------------------------
# A chunk's mesh reads its 26 neighbours (cube corners, gradient normals, vertex colours), most of which live on other
# shards.  Per recompute: (1) the union of all shards' meshesToUpdate is formed, (2) every shard takes the ids it owns as
# its jobs, (3) asks the owners for the neighbours of its jobs it does not own, (4) imports what they hold as ghost
# chunks, (5) recomputes its jobs, (6) drops the ghosts.  mesh_plan() is the pure part of that.
def mesh_plan(union_ids, rank, world, owner_of):
    """union_ids: iterable of (x, y, z); owner_of(id) -> rank.
    -> (jobs: sorted list of ids this rank meshes, requests: {owner: sorted list of ids to ask that owner for})"""
    jobs = sorted({tuple(int(v) for v in i) for i in union_ids if owner_of(i) == rank})
    want = set()
    for x, y, z in jobs:
        for dz in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    if dx or dy or dz:
                        want.add((x + dx, y + dy, z + dz))
    requests = {}
    for i in want:
        o = owner_of(i)
        if o != rank:
            requests.setdefault(o, []).append(i)
    return jobs, {o: sorted(v) for o, v in requests.items()}


# ---- the device plan restated in numpy (tests: CPU stand-in maps, and the check of chisel_hip_shell_plan_device on the GPU) ----------
def shell_box_of(d):
    """box code of direction d = G - J (what job J reads of its neighbour G): per axis d > 0 -> {0, 1} (1), d < 0 -> {N - 1} (2), 0 -> all (0)"""
    return sum((1 if v > 0 else (2 if v < 0 else 0)) << (2 * a) for a, v in enumerate(d))


def shell_box_coords(code, n):
    rng = lambda c: range(n) if c == 0 else (range(0, 2) if c == 1 else (range(n - 1, n) if c == 2 else (0, 1, n - 1)))
    return [(x, y, z) for z in rng((code >> 4) & 3) for y in rng((code >> 2) & 3) for x in rng(code & 3)]


def shell_item_covered(j, d, r, jobset, owner_of):
    """job j of rank r reads box(d) of its neighbour g = j + d: is that box part of a box another job of r reads of g anyway?  A box holds
    another when it is "all" wherever it differs: the job at g - d', d' = d with some (not all) of its non-zero components zeroed."""
    g = (j[0] + d[0], j[1] + d[1], j[2] + d[2])
    nz = [a for a in range(3) if d[a]]
    for mask in range(1, (1 << len(nz)) - 1):
        d2 = list(d)
        for k, a in enumerate(nz):
            if (mask >> k) & 1:
                d2[a] = 0
        j2 = (g[0] - d2[0], g[1] - d2[1], g[2] - d2[2])
        if j2 in jobset and owner_of(j2) == r:
            return True
    return False


def plan_shells_reference(entries, world, rank, owner_of):
    """entries: (n, 4) (x, y, z, flag) of ALL ranks.  -> (jobs: sorted ids this rank owns, send: {peer: [(x, y, z, box), ...]} -- what this
    rank owns and the jobs of `peer` read --, recv: {owner: [(x, y, z, box), ...]}): one item per (job, direction), less the items whose box
    is part of another item's of the same (rank, ghost) (shell_item_covered: an edge or corner box beside the face box that holds it)."""
    jobset = set()
    for x, y, z, flag in np.asarray(entries, np.int64).reshape(-1, 4).tolist():
        r = 0 if flag else 1
        jobset.update((x + dx, y + dy, z + dz) for dx in range(-r, r + 1) for dy in range(-r, r + 1) for dz in range(-r, r + 1))
    jobs = sorted(j for j in jobset if owner_of(j) == rank)
    send, recv = {}, {}
    for j in sorted(jobset):
        r = owner_of(j)
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    if not (dx or dy or dz):
                        continue
                    g = (j[0] + dx, j[1] + dy, j[2] + dz)
                    o = owner_of(g)
                    if o == r:
                        continue
                    if (o == rank or r == rank) and shell_item_covered(j, (dx, dy, dz), r, jobset, owner_of):
                        continue
                    item = g + (shell_box_of((dx, dy, dz)),)
                    if o == rank:
                        send.setdefault(r, []).append(item)
                    elif r == rank:
                        recv.setdefault(o, []).append(item)
    return jobs, send, recv


def segment_bytes(items, voxels, color):
    return 16 + 32 * items + (12 if color else 8) * voxels


def pack_segment(items, found, sdf, wgt, col):
    """one segment (bytes): items (n, 4), found (n,), the boxes' voxels back to back in item order (sdf, wgt: float32; col: uint32 or None)"""
    items = np.asarray(items, np.int32).reshape(-1, 4)
    n, vox = len(items), len(sdf)
    head = np.array([n, vox, 0, 0], np.int32)
    rec = np.zeros((n, 8), np.int32)
    rec[:, :4] = items
    rec[:, 4] = found
    return head.tobytes() + rec.tobytes() + np.asarray(sdf, np.float32).tobytes() + np.asarray(wgt, np.float32).tobytes() + \
        (np.asarray(col, np.uint32).tobytes() if col is not None else b"")


def unpack_segment(buf, color):
    """-> (rec (n, 8) int32: x, y, z, box, found, first voxel, 0, 0; sdf, wgt, col or None) of one segment given as a uint8 array"""
    buf = np.ascontiguousarray(buf, np.uint8)
    head = buf[:16].view(np.int32)
    n, vox = int(head[0]), int(head[1])
    rec = buf[16:16 + 32 * n].view(np.int32).reshape(n, 8)
    at = 16 + 32 * n
    sdf = buf[at:at + 4 * vox].view(np.float32)
    wgt = buf[at + 4 * vox:at + 8 * vox].view(np.float32)
    col = buf[at + 8 * vox:at + 12 * vox].view(np.uint32) if color else None
    return rec, sdf, wgt, col


class LocalShardGroup:
    """All shards of a map in ONE process (tests; a single GPU holding several shards): the same protocol as
    ShardedChisel.UpdateMeshes with the exchange done by direct calls."""

    def __init__(self, shards, device_payload=True):
        self.shards = shards
        self.world = len(shards)
        self.calls = 0
        self.device_payload = device_payload

    def IntegratePointCloud(self, integrator, cloud, extrinsic, truncation, max_dist):
        """Chisel::IntegratePointCloud on every shard: each lists and updates only the chunks it owns (cloud_prepare_kernel)."""
        for s_ in self.shards:
            s_.IntegratePointCloud(integrator, cloud, extrinsic, truncation, max_dist)

    def UpdateMeshes(self, force=False, wait_free=False, stride=None):
        """the protocol of ShardedChisel.UpdateMeshes with the two collectives done by copies between the shards' buffers.
        wait_free: its wait-free form (fixed segments of `stride` bytes -- default: half as much again as the largest segment of the
        previous recompute --, nothing read between the steps; a recompute that does not fit is called off on the device and made again
        the blocking way).  -> the all-reduced status of a wait-free recompute, else None"""
        import torch
        self.calls += 1
        if not force and (self.calls - 1) % 10:  # Chisel.cpp:53-58: every 10th call
            return None
        if wait_free and (stride or getattr(self, "_seg_need", 0)):
            return self._update_wait_free(stride or (self._seg_need + self._seg_need // 2 + 4096 + 15) // 16 * 16)
        return self._update_blocking()

    def _update_wait_free(self, stride):
        import torch
        W = self.world
        dev = torch.device("cuda", torch.cuda.current_device())
        cap = getattr(self, "_dirty_cap", 1 << 12)
        gathered = torch.zeros((W, 1 + 4 * cap), dtype=torch.int32, device=dev)
        status = torch.zeros((W, 8), dtype=torch.int32, device=dev)
        send = [torch.zeros((W * stride,), dtype=torch.uint8, device=dev) for _ in range(W)]
        torch.cuda.current_stream().synchronize()
        for r, s_ in enumerate(self.shards):
            s_.DirtyIdsDevice(gathered[r])
            s_.synchronize()  # (stands for the all-gather's event)
        for r, s_ in enumerate(self.shards):
            s_.PlanShellsQueue(gathered.view(-1), W, cap, stride, status[r], send[r])
            s_.synchronize()  # (... for the events in front of the all-reduce and the all-to-all)
        reduced = status.max(dim=0).values.contiguous()  # the all-reduce (MAX)
        recv = [torch.cat([send[o][r * stride:(r + 1) * stride] for o in range(W)]) for r in range(W)]  # the all-to-all of equal splits
        torch.cuda.current_stream().synchronize()
        for r, s_ in enumerate(self.shards):
            s_.ImportShellsFixed(recv[r], stride, reduced)
            s_.UpdateMeshesPlanned()
            s_.DropGhostChunks()
        self._keep = (gathered, status, reduced, send, recv)  # (the device reads them until the commit below has settled the mesh step)
        st = [int(v) for v in reduced.tolist()]  # (a host wait here: this class is a test vehicle; ShardedChisel reads it in Settle())
        for s_ in self.shards:
            s_.ShellCommit(st[0] != 0)
        self.last_status = st
        self._seg_need = st[2]
        self.ghost_bytes = st[6] * (12 if self.shards[0].use_color else 8)
        if st[0] & 1:
            self._dirty_cap = 2 * st[1]
        if st[0]:
            self.wait_free_aborts = getattr(self, "wait_free_aborts", 0) + 1
            self._update_blocking()
        return st

    def _update_blocking(self):
        import torch
        W = self.world
        dev = torch.device("cuda", torch.cuda.current_device())
        cap = getattr(self, "_dirty_cap", 1 << 12)
        while True:
            gathered = torch.zeros((W, 1 + 4 * cap), dtype=torch.int32, device=dev)
            torch.cuda.current_stream().synchronize()
            for r, s_ in enumerate(self.shards):
                s_.DirtyIdsDevice(gathered[r])
                s_.synchronize()
            plans = [s_.PlanShellsDevice(gathered.view(-1), W, cap) for s_ in self.shards]
            mx = max(p["max_count"] for p in plans)
            if mx <= cap:
                break
            cap = self._dirty_cap = 2 * mx
        self.plans = plans
        send = []
        for r, s_ in enumerate(self.shards):
            sizes = [s_.ShellSegmentBytes(*plans[r]["send"][p]) for p in range(W)]
            buf = torch.empty((sum(sizes),), dtype=torch.uint8, device=dev)
            torch.cuda.current_stream().synchronize()
            s_.ExportShellsPacked(buf)
            s_.synchronize()
            send.append((buf, np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)))
            self._seg_need = max([getattr(self, "_seg_need", 0) if r else 0] + sizes)
        self.ghost_bytes = 0
        for r, s_ in enumerate(self.shards):
            sizes = [s_.ShellSegmentBytes(*plans[r]["recv"][o]) for o in range(W)]
            # what the all-to-all does: segment (o -> r) of every owner's send buffer, in owner order
            parts = [send[o][0][int(send[o][1][r]):int(send[o][1][r + 1])] for o in range(W)]
            assert [int(p.numel()) for p in parts] == sizes, (r, [int(p.numel()) for p in parts], sizes)
            recv = torch.cat(parts) if parts else torch.empty((0,), dtype=torch.uint8, device=dev)
            torch.cuda.current_stream().synchronize()
            s_.ImportShellsPacked(recv)
            self.ghost_bytes += int(plans[r]["recv"][:, 1].sum()) * (12 if s_.use_color else 8)
        for s_ in self.shards:
            s_.UpdateMeshesPlanned()
            s_.DropGhostChunks()


class ShardedChisel:
    """chisel::Chisel surface over the shards of one node (only what the sharded path changes)."""

    def __init__(self, local_map, exchange, integrator):
        self.map = local_map
        self.x = exchange
        self.integrator = integrator

    def IntegrateBatch(self, local_depth, local_poses, local_cameras, local_color=None):
        """Every rank passes the frames it ingested (x.per of them); all ranks integrate the whole batch in frame order."""
        torch = self.x.torch
        self.Settle()
        # This is the simple, synchronous form (PipelinedExchange is the overlapped one): the map's streams may still be reading
        # the receive / colour buffer and the temporaries of the call before last, which the exchange below refills on the
        # communication library's stream -- wait for the map first.
        self.map.synchronize()
        meta = torch.from_numpy(np.stack([pack_meta(p, c) for p, c in zip(local_poses, local_cameras)])).to(local_depth.device)
        depth, meta_all, color = self.x.exchange(local_depth, meta, local_color)
        meta_host = meta_all.cpu().numpy()
        frames, colors = [], ([] if color is not None else None)
        for j in range(self.x.batch):
            pose, cam = unpack_meta(meta_host[j], self.x.W, self.x.H)
            frames.append((depth[j], pose, cam))
            if color is not None:
                colors.append((color[j], pose, cam))
        self.map.IntegrateBatch(self.integrator, frames, colors)
        return frames

    def IntegratePointCloud(self, points, colors, extrinsic, truncation, max_dist, src=0):
        """Chisel::IntegratePointCloud of the sharded map: rank `src` ingested the cloud (device tensors (n, 3); the other ranks
        pass None), it is broadcast over RCCL and every rank updates the chunks it owns -- no other exchange is needed, the
        update of a chunk depends on the cloud and that chunk only."""
        torch, dist = self.x.torch, self.x.dist
        self.Settle()
        rank = dist.get_rank()
        head = [None]
        if rank == src:
            head[0] = (int(points.shape[0]), colors is not None, np.asarray(extrinsic, np.float32)[:3, :4].tolist())
        dist.broadcast_object_list(head, src=src)
        n, has_color, pose = head[0]
        pose = np.asarray(pose, np.float32)
        dev = self.x.device
        pts = points.contiguous() if rank == src else torch.empty((n, 3), dtype=torch.float32, device=dev)
        dist.broadcast(pts, src=src)
        cols = None
        if has_color:
            cols = colors.contiguous() if rank == src else torch.empty((n, 3), dtype=torch.float32, device=dev)
            dist.broadcast(cols, src=src)
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()  # the map runs on its own stream
        self.map.IntegratePointCloud(self.integrator, (pts if dev.type == "cuda" else pts.numpy(),
                                                       None if cols is None else (cols if dev.type == "cuda" else cols.numpy())),
                                     pose, truncation, max_dist)

    def _all_to_all(self, out, inp, n_recv, n_send):
        """all_to_all_single; gloo has no device all-to-all: functional check only, bounced through the host"""
        dist = self.x.dist
        if dist.get_backend() == "gloo" and inp.is_cuda:
            o = self.x.torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), n_recv, n_send)  # (no sizes: equal splits)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp, n_recv, n_send)

    def UpdateMeshes(self, force=False, ids=None, wait_free=False):
        """Chisel::UpdateMeshes of the sharded map: every rank ends up with the meshes of the chunks it owns.

        1. every rank lists the chunks it has updated since the last recompute as a device int array (chisel_hip_dirty_ids_device) and
           ONE all_gather hands every rank all of them (a fixed-capacity int tensor: no pickled objects, no sizes to agree on first);
        2. the plan is made ON THE DEVICE from that tensor (chisel_hip_shell_plan_device: this rank's jobs, the shells it owes every
           peer, how much every owner owes it) -- the host reads 4 W + 4 numbers, its one wait of the recompute;
        3. the owners pack the SHELLS (the one or two voxel layers a neighbour's mesh reads, not whole chunks) into one byte segment
           per peer whose items say where their voxels are, ONE all_to_all moves the segments, the receivers install what arrived as
           ghost chunks; map stream and collective stream are ordered by events (record_event / wait_event);
        4. every rank recomputes its jobs (a list that never left the device) and drops the ghosts.
        ids: mesh these chunks (every rank passes ids of its own choice, the union is meshed) instead of meshesToUpdate.
        wait_free: the form in which the host reads NOTHING in between (_recompute_wait_free): everything is queued, the segments have a
        size the ranks agreed on at the previous recompute, and a recompute that did not fit them is called off on the device and made
        again -- in segments of the right size, or the way above if a list or table overflowed -- when the host next looks (Settle(): call
        it, or any method of this class, before touching self.map).
        -> bytes of ghost voxels this rank received (wait_free: of the rank that received most, at the previous recompute)"""
        self._mesh_calls = getattr(self, "_mesh_calls", 0) + 1
        if not force and (self._mesh_calls - 1) % 10:  # Chisel.cpp:53-58: every 10th call
            return 0
        self.Settle()
        if self.x.world == 1 and not getattr(self, "force_collectives", False):  # (force_collectives: tools/nccl_world1_check.py runs the N > 1 code on one rank)
            if ids is None:
                self.map.UpdateMeshes(force=True)
            else:
                self.map.UpdateMeshesOf(np.asarray(ids, np.int32).reshape(-1, 3))
            return 0
        if wait_free and ids is None and getattr(self, "_est", None) is not None and hasattr(self.map, "PlanShellsQueue"):
            return self._recompute_wait_free()
        return self._recompute_blocking(ids, post_sizes=wait_free)

    def _lap(self, name):
        """CHISEL_HIP_HOST_TIMING=1: host phases of a sharded recompute, summed in self.phase_us"""
        import os
        import time
        if getattr(self, "_timing", None) is None:
            self._timing = bool(os.environ.get("CHISEL_HIP_HOST_TIMING"))
        now = time.perf_counter()
        if self._timing and name is not None:
            self.phase_us = getattr(self, "phase_us", {})
            self.phase_us[name] = self.phase_us.get(name, 0.0) + (now - self._t_prev) * 1e6
        self._t_prev = now

    def _mesh_buffers(self, cap):
        """the two int tensors of step 1 (kept between recomputes) and the status vectors of the wait-free form"""
        torch, world, dev = self.x.torch, self.x.world, self.x.device
        key = (cap, world)
        if getattr(self, "_mesh_bufs_key", None) != key:
            self._mesh_bufs_key = key
            self._dirty_buf = torch.zeros((1 + 4 * cap,), dtype=torch.int32, device=dev)
            self._gathered = torch.zeros((world * (1 + 4 * cap),), dtype=torch.int32, device=dev)
            if dev.type == "cuda":
                self._order_map_after_collectives()  # (torch's allocator hands out memory its own stream may still be using)
        if getattr(self, "_status", None) is None:
            self._status = torch.zeros((8,), dtype=torch.int32, device=dev)
            self._status_host = torch.zeros((8,), dtype=torch.int32, pin_memory=dev.type == "cuda")
            self._status_event = torch.cuda.Event() if dev.type == "cuda" else None
        return self._dirty_buf, self._gathered

    def _list_dirty(self, buf, ids):
        on_gpu = self.x.device.type == "cuda"
        cap = (buf.numel() - 1) // 4
        if ids is None and on_gpu:
            self.map.DirtyIdsDevice(buf)  # (writes the count itself; the buffer's last reader, the previous recompute's all-gather, is long through)
            self._order_after_map()
        else:
            e = np.asarray(self.map.DirtyEntries(), np.int32).reshape(-1, 4) if ids is None else \
                np.concatenate([np.asarray(ids, np.int32).reshape(-1, 3), np.ones((len(np.asarray(ids).reshape(-1, 3)), 1), np.int32)], axis=1)
            head = np.zeros(1 + 4 * cap, np.int32)
            head[0] = len(e)
            head[1:1 + 4 * min(len(e), cap)] = e[:cap].reshape(-1)
            buf.copy_(self.x.torch.from_numpy(head))

    def _gather_dirty(self, gathered, buf):
        torch, dist = self.x.torch, self.x.dist
        if dist.get_backend() == "gloo" and buf.is_cuda:  # functional check only: gloo has no device collectives
            host = torch.empty(gathered.shape, dtype=torch.int32)
            dist.all_gather_into_tensor(host, buf.cpu())
            gathered.copy_(host)
        else:
            dist.all_gather_into_tensor(gathered, buf)

    def _reduce_status(self):
        """the ranks' status vectors -> their element-wise maximum on every rank, and on its way to the host (Settle reads it)"""
        torch, dist = self.x.torch, self.x.dist
        st = self._status
        if dist.get_backend() == "gloo" and st.is_cuda:
            host = st.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MAX)
            st.copy_(host)
        else:
            dist.all_reduce(st, op=dist.ReduceOp.MAX)
        self._status_host.copy_(st, non_blocking=True)
        if self._status_event is not None:
            self._status_event.record(torch.cuda.current_stream())

    def _recompute_blocking(self, ids, post_sizes=False):
        torch, dist, world, rank, dev = self.x.torch, self.x.dist, self.x.world, self.x.rank, self.x.device
        on_gpu = dev.type == "cuda"
        self._lap(None)
        # ---- 1. the ranks' dirty chunks, 2. the plan
        while True:
            cap = self._dirty_cap = getattr(self, "_dirty_cap", 1 << 12)  # (entries per rank in the gathered tensor; doubled below when a rank has more)
            buf, gathered = self._mesh_buffers(cap)
            self._list_dirty(buf, ids)
            self._lap("dirty ids (issue)")
            self._gather_dirty(gathered, buf)
            self._lap("all_gather (issue)")
            if on_gpu:
                self._order_map_after_collectives()
            plan = self.map.PlanShellsDevice(gathered, world, cap)  # the one host wait of a recompute
            self._lap("plan (device) + wait")
            if plan["max_count"] <= cap:
                break
            self._dirty_cap = 2 * plan["max_count"]  # (every rank sees the same counts and takes the same turn)
        # ---- 3. shells: export -> all_to_all -> import
        color = bool(getattr(self.map, "use_color", False))
        s_bytes = [segment_bytes(int(i), int(v), color) for i, v in plan["send"]]
        r_bytes = [segment_bytes(int(i), int(v), color) for i, v in plan["recv"]]
        # the two byte buffers are kept between recomputes and only ever grow (their last readers -- the previous recompute's
        # all-to-all and its drop kernel -- finished before the plan's wait above returned)
        need = (sum(s_bytes), sum(r_bytes))
        self._segment_buffers(need[0], need[1])
        send, recv = self._seg_send[:need[0]], self._seg_recv[:need[1]]
        self.map.ExportShellsPacked(send)
        if on_gpu:
            self._order_after_map()
        self._lap("export (issue)")
        self._all_to_all(recv, send, r_bytes, s_bytes)
        self._lap("all_to_all (issue)")
        if on_gpu:
            self._order_map_after_collectives()
        self.map.ImportShellsPacked(recv)
        self._lap("import (issue)")
        # ---- 4.
        self.map.UpdateMeshesPlanned()
        self.map.DropGhostChunks()
        self._lap("recompute + drop (issue)")
        # what the next recompute may size its fixed segments from (the wait-free form): the ranks' needs, reduced like its status
        if post_sizes and ids is None and hasattr(self.map, "PlanShellsQueue"):
            mine = np.array([0, plan["max_count"], max(s_bytes) if s_bytes else 0, plan["jobs"], int(plan["recv"][:, 0].sum()), plan["send_items"],
                             min(int(plan["recv"][:, 1].sum()), 2**31 - 1), 0], np.int32)
            self._status.copy_(torch.from_numpy(mine))
            self._reduce_status()
            self._pending = "sizes"
            self._lap("sizes for the next one (issue)")
        if getattr(self, "_timing", False):
            self.phase_us["recomputes"] = self.phase_us.get("recomputes", 0) + 1
        per_voxel = 12 if getattr(self.map, "use_color", False) else 8
        total = int(plan["recv"][:, 1].sum())
        edge = int(self.map.chunk_size[0])
        # for the record: what whole ghost chunks (the round-2 protocol) would have moved for the same ghosts.  The number of distinct
        # ghosts is counted on the device and reaches the host with the NEXT plan: both figures cover the recomputes before this one.
        self.whole_chunk_bytes = int(plan.get("ghosts_before", 0)) * edge ** 3 * per_voxel
        self.shell_bytes = getattr(self, "_shell_bytes_all", 0)
        self._shell_bytes_all = self.shell_bytes + int(total * per_voxel)
        self.last_plan = plan
        return int(total * per_voxel)

    def _segment_buffers(self, n_send, n_recv):
        torch, dev = self.x.torch, self.x.device
        if getattr(self, "_seg_cap", (0, 0))[0] < n_send or self._seg_cap[1] < n_recv:
            self._seg_cap = (max(2 * n_send, 1 << 20), max(2 * n_recv, 1 << 20))
            self._seg_send = torch.empty((self._seg_cap[0],), dtype=torch.uint8, device=dev)
            self._seg_recv = torch.empty((self._seg_cap[1],), dtype=torch.uint8, device=dev)
            if dev.type == "cuda":
                self._order_map_after_collectives()  # (torch's allocator hands out memory its own stream may still be using)

    def _recompute_wait_free(self):
        """The recompute as ONE burst of queued work -- the host reads nothing until Settle():
             dirty ids -> all_gather -> plan + status + export (map stream) -> all_reduce(status, MAX) + all_to_all of EQUAL splits (collective
             stream) -> import + mesh + drop (map stream).
        The segments' size comes from the previous recompute (the largest segment any rank sent, plus a half); a rank whose list, plan or
        segment does not fit says so in its status, the all-reduce tells everybody, and the kernels behind the exchange do nothing at all:
        Settle() then makes the recompute again the blocking way, from a map nobody has touched."""
        torch, dist, world, dev = self.x.torch, self.x.dist, self.x.world, self.x.device
        on_gpu = dev.type == "cuda"
        est = self._est
        self._lap(None)
        if 2 * est["max_count"] > getattr(self, "_dirty_cap", 1 << 12):
            self._dirty_cap = 4 * est["max_count"]  # (the same figures, the same turn on every rank)
        cap = self._dirty_cap = getattr(self, "_dirty_cap", 1 << 12)
        # Room beyond the previous recompute's largest segment: a half (CHISEL_HIP_SHELL_SLACK_PERCENT: A/B).  The need mostly shrinks or holds
        # from one recompute to the next and jumps when the camera turns towards new space -- without warning: sizing by the last step's
        # growth called off as many recomputes as a fixed quarter (EXPERIMENTS.md).  Every rank computes this from the same figures.
        slack = int(__import__("os").environ.get("CHISEL_HIP_SHELL_SLACK_PERCENT", "50"))
        if getattr(self, "_exact_retry", False):
            slack = 0  # (Settle's second go at a recompute whose slots were too small: est holds what THIS recompute needs)
        stride = (est["seg_bytes"] + est["seg_bytes"] * slack // 100 + 4096 + 15) // 16 * 16
        self._last_stride = stride
        buf, gathered = self._mesh_buffers(cap)
        self._segment_buffers(world * stride, world * stride)
        send, recv = self._seg_send[:world * stride], self._seg_recv[:world * stride]
        self._list_dirty(buf, None)
        self._lap("dirty ids (issue)")
        self._gather_dirty(gathered, buf)
        self._lap("all_gather (issue)")
        if on_gpu:
            self._order_map_after_collectives()
        self.map.PlanShellsQueue(gathered, world, cap, stride, self._status, send, est["send_items"])
        if on_gpu:
            self._order_after_map()
        self._lap("plan + export (issue)")
        self._reduce_status()
        self._all_to_all(recv, send, None, None)
        self._lap("all_reduce + all_to_all (issue)")
        if on_gpu:
            self._order_map_after_collectives()
        self.map.ImportShellsFixed(recv, stride, self._status, est["jobs"], est["recv_items"])
        self._lap("import (issue)")
        self.map.UpdateMeshesPlanned()
        self.map.DropGhostChunks()
        self._lap("recompute + drop (issue)")
        self._pending = "recompute"
        self.wire_bytes = getattr(self, "wire_bytes", 0) + (world - 1) * stride
        self.wait_free_recomputes = getattr(self, "wait_free_recomputes", 0) + 1
        if getattr(self, "_timing", False):
            self.phase_us["recomputes"] = self.phase_us.get("recomputes", 0) + 1
        per_voxel = 12 if getattr(self.map, "use_color", False) else 8
        return int(est["recv_voxels"]) * per_voxel

    def Settle(self):
        """The host's look at what UpdateMeshes left in flight -- before it (or anybody who goes to self.map directly) changes or reads the
        map again.  After a blocking recompute: the sizes the ranks exchanged for the next one.  After a wait-free one: its all-reduced
        status; the mesh step's totals are settled (chisel_hip_shell_commit), and a recompute that was called off is made again -- in slots of
        exactly the size its status reported when only they were too small, in the blocking form otherwise."""
        pending = getattr(self, "_pending", None)
        if not pending:
            return
        self._pending = None
        self._lap(None)
        if self._status_event is not None:
            self._status_event.synchronize()
        st = [int(v) for v in self._status_host.tolist()]
        self._lap("settle (wait)")
        self._est = {"max_count": st[1], "seg_bytes": st[2], "jobs": st[3], "recv_items": st[4], "send_items": st[5], "recv_voxels": st[6]}
        if __import__("os").environ.get("CHISEL_HIP_WF_DEBUG") and self.x.rank == 0:
            print("settle", pending, "stride", getattr(self, "_last_stride", None), "status", st, file=__import__("sys").stderr, flush=True)
        if pending == "recompute":
            self.map.ShellCommit(st[0] != 0)
            per_voxel = 12 if getattr(self.map, "use_color", False) else 8
            self.shell_bytes = getattr(self, "_shell_bytes_all", 0)
            self.whole_chunk_bytes = st[7] * int(self.map.chunk_size[0]) ** 3 * per_voxel  # (as in the blocking form: both cover the recomputes before this one)
            if st[0] == 0:
                self._shell_bytes_all = self.shell_bytes + st[6] * per_voxel  # (of the rank that received most)
            else:
                self.wait_free_aborts = getattr(self, "wait_free_aborts", 0) + 1
                self.last_abort = st[0]
                self.abort_bits = getattr(self, "abort_bits", {})
                self.abort_bits[st[0]] = self.abort_bits.get(st[0], 0) + 1
                self.last_abort_status = st
                if st[0] == 4 and not getattr(self, "_exact_retry", False):
                    # only the slots were too small, and the status says by how much: the same recompute again (the map is as it was, so is
                    # the plan) in slots that hold exactly this -- still without reading the plan; settled right away, it cannot fail the same way
                    self._exact_retry = True
                    try:
                        self._recompute_wait_free()
                        self.wait_free_retries = getattr(self, "wait_free_retries", 0) + 1
                        self.Settle()
                    finally:
                        self._exact_retry = False
                else:
                    self._recompute_blocking(None, post_sizes=True)
            self._lap("settle (commit)")

    def _order_after_map(self):
        """the collective queued next on torch's current stream starts after what the map has queued so far (an event, no host wait)"""
        torch = self.x.torch
        if hasattr(self.map, "order_stream_after_map"):  # one call into the library instead of three through torch (19 -> 4 us)
            self.map.order_stream_after_map(torch.cuda.current_stream().cuda_stream)
            return
        ev = self._ev = getattr(self, "_ev", None) or torch.cuda.Event()
        ev.record(torch.cuda.current_stream())  # (a first record creates the hipEvent_t behind the torch object)
        self.map.record_event(ev.cuda_event)
        torch.cuda.current_stream().wait_event(ev)

    def _order_map_after_collectives(self):
        """the map's next call starts after the collectives queued on torch's current stream"""
        torch = self.x.torch
        if hasattr(self.map, "order_map_after_stream"):
            self.map.order_map_after_stream(torch.cuda.current_stream().cuda_stream)
            return
        ev = self._ev2 = getattr(self, "_ev2", None) or torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.map.wait_event(ev.cuda_event)

    def Reset(self):
        """Chisel::Reset of this rank's shard (every rank calls it): what the last recompute left in flight is settled first, and the sizes the
        wait-free form would have gone by are forgotten with the map they described (its next recompute takes the blocking form)"""
        self.Settle()
        self.map.Reset()
        self._est = None

    def NumChunks(self):
        torch, dist = self.x.torch, self.x.dist
        self.Settle()
        n = torch.tensor([self.map.NumChunks()], dtype=torch.int64, device=self.x.recv[0].device)
        if self.x.world > 1:
            dist.all_reduce(n)
        return int(n.item())

    def GatherChunkIDs(self):
        self.Settle()
        ids = np.asarray(self.map.GetChunkIDs(), np.int32).reshape(-1, 3)
        if self.x.world == 1:
            return ids
        out = [None] * self.x.world
        self.x.dist.all_gather_object(out, ids)
        return np.concatenate(out, axis=0)
