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

    def UpdateMeshes(self, force=False):
        from .chisel import mesh_shell_plan
        self.calls += 1
        if not force and (self.calls - 1) % 10:  # Chisel.cpp:53-58: every 10th call
            return
        entries = np.concatenate([s_.DirtyEntries() for s_ in self.shards], axis=0)
        plans = [mesh_shell_plan(entries, self.world, r) for r in range(self.world)]
        self.ghost_bytes = 0
        for r, (jobs, items) in enumerate(plans):
            for o in sorted(set(items[:, 0].tolist())):
                it = items[items[:, 0] == o][:, 1:5]
                sdf, wgt, col, found = self.shards[o].ExportShells(it, device=self.device_payload)  # payload stays in HBM
                if self.device_payload:
                    self.shards[o].synchronize()
                self.shards[r].ImportGhostShells(it, sdf, wgt, col, found)
                self.ghost_bytes += int(sdf.nbytes if isinstance(sdf, np.ndarray) else sdf.numel() * 4) * (3 if col is not None else 2)
        for r, (jobs, _) in enumerate(plans):
            self.shards[r].UpdateMeshesOf(jobs)
            self.shards[r].DropGhostChunks()


class ShardedChisel:
    """chisel::Chisel surface over the shards of one node (only what the sharded path changes)."""

    def __init__(self, local_map, exchange, integrator):
        self.map = local_map
        self.x = exchange
        self.integrator = integrator

    def IntegrateBatch(self, local_depth, local_poses, local_cameras, local_color=None):
        """Every rank passes the frames it ingested (x.per of them); all ranks integrate the whole batch in frame order."""
        torch = self.x.torch
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
            dist.all_to_all_single(o, inp.cpu(), n_recv, n_send)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp, n_recv, n_send)

    def UpdateMeshes(self, force=False, ids=None):
        """Chisel::UpdateMeshes of the sharded map: every rank ends up with the meshes of the chunks it owns.

        1. every rank lists the chunks it has updated since the last recompute as a device int array (chisel_hip_dirty_ids_device) and
           ONE all_gather hands every rank all of them (a fixed-capacity int tensor: no pickled objects, no sizes to agree on first);
        2. the plan is host arithmetic every rank evaluates for every rank (chisel_hip_mesh_shell_plan): its own jobs, the ghosts it
           needs, and what the others will ask of it -- the request lists are never exchanged;
        3. the owners pack the requested SHELLS (the one or two voxel layers a neighbour's mesh reads, not whole chunks:
           chisel_hip_export_shells), one all_to_all per voxel array moves them, the receivers install them as ghost chunks
           (chisel_hip_import_ghost_shells); map stream and collective stream are ordered by events (record_event / wait_event),
           the host waits once, for the gathered id list of step 1;
        4. every rank recomputes its jobs and drops the ghosts.
        ids: mesh these chunks (every rank passes ids of its own choice, the union is meshed) instead of meshesToUpdate.
        -> bytes of ghost voxels this rank received"""
        from .chisel import mesh_shell_plan_all, shell_volumes
        self._mesh_calls = getattr(self, "_mesh_calls", 0) + 1
        if not force and (self._mesh_calls - 1) % 10:  # Chisel.cpp:53-58: every 10th call
            return 0
        torch, dist, world, rank, dev = self.x.torch, self.x.dist, self.x.world, self.x.rank, self.x.device
        if world == 1 and not getattr(self, "force_collectives", False):  # (force_collectives: tools/nccl_world1_check.py runs the N > 1 code on one rank)
            if ids is None:
                self.map.UpdateMeshes(force=True)
            else:
                self.map.UpdateMeshesOf(np.asarray(ids, np.int32).reshape(-1, 3))
            return 0
        on_gpu = dev.type == "cuda"
        edge = int(self.map.chunk_size[0])
        import time
        timing = True  # host phases of a sharded recompute, summed in self.phase_us (six perf_counter calls per recompute)
        t_prev = [time.perf_counter()]
        def lap(name):
            if timing:
                now = time.perf_counter()
                self.phase_us = getattr(self, "phase_us", {})
                self.phase_us[name] = self.phase_us.get(name, 0.0) + (now - t_prev[0]) * 1e6
                t_prev[0] = now
        # ---- 1. the ranks' dirty chunks
        while True:
            cap = self._dirty_cap = getattr(self, "_dirty_cap", 1 << 12)  # (entries per rank in the gathered tensor; doubled below when a rank has more)
            buf = torch.zeros((1 + 4 * cap,), dtype=torch.int32, device=dev)
            if ids is None and on_gpu:
                self._order_map_after_collectives()  # (the buffer was zeroed on torch's stream)
                self.map.DirtyIdsDevice(buf)
                self._order_after_map()
            else:
                e = np.asarray(self.map.DirtyEntries(), np.int32).reshape(-1, 4) if ids is None else \
                    np.concatenate([np.asarray(ids, np.int32).reshape(-1, 3), np.ones((len(np.asarray(ids).reshape(-1, 3)), 1), np.int32)], axis=1)
                head = np.zeros(1 + 4 * cap, np.int32)
                head[0] = len(e)
                head[1:1 + 4 * min(len(e), cap)] = e[:cap].reshape(-1)
                buf.copy_(torch.from_numpy(head))
            gathered = torch.empty((world * (1 + 4 * cap),), dtype=torch.int32, device=dev)
            if dist.get_backend() == "gloo" and on_gpu:  # functional check only: gloo has no device collectives
                host = torch.empty(gathered.shape, dtype=torch.int32)
                dist.all_gather_into_tensor(host, buf.cpu())
            else:
                dist.all_gather_into_tensor(gathered, buf)
                host = gathered.cpu()  # the one host wait of a recompute
            g = host.numpy().reshape(world, 1 + 4 * cap)
            if int(g[:, 0].max()) <= cap:
                break
            self._dirty_cap = 2 * int(g[:, 0].max())  # (every rank sees the same counts and takes the same turn)
        entries = np.concatenate([g[r, 1:1 + 4 * int(g[r, 0])].reshape(-1, 4) for r in range(world)], axis=0)
        lap("dirty ids + all_gather + host copy")
        # ---- 2. the plans
        # (all ranks' plans in ONE pass of the planner: chisel_hip_mesh_shell_plan_all -- evaluating chisel_hip_mesh_shell_plan once per
        # rank cost every rank 1-2 ms per rank and recompute)
        all_jobs, all_items = mesh_shell_plan_all(entries, world)
        none = np.zeros((0, 4), np.int32)
        jobs = all_jobs[rank]
        recv_parts = [all_items.get((rank, o), none) for o in range(world)]   # what this rank asks of owner o
        recv_items = np.concatenate(recv_parts, axis=0)
        n_recv = [len(p) for p in recv_parts]
        send_parts = [all_items.get((q, rank), none) for q in range(world)]   # what rank q asks of this rank
        send_items = np.concatenate(send_parts, axis=0)
        n_send = [len(p) for p in send_parts]
        vol = lambda it: int(shell_volumes(it[:, 3], edge).sum()) if len(it) else 0
        v_send = [vol(p) for p in send_parts]
        v_recv = [vol(recv_items[sum(n_recv[:o]):sum(n_recv[:o + 1])]) for o in range(world)]
        lap("plan")
        # ---- 3. shells: export -> all_to_all -> import
        if on_gpu:
            self._order_map_after_collectives()
            sdf_s, wgt_s, col_s, found_s = self.map.ExportShells(send_items, device=True)
            self._order_after_map()
        else:
            a, b, c, f = self.map.ExportShells(send_items)
            sdf_s, wgt_s = torch.from_numpy(np.ascontiguousarray(a, np.float32)), torch.from_numpy(np.ascontiguousarray(b, np.float32))
            col_s = torch.from_numpy(np.ascontiguousarray(c, np.uint8)) if c is not None else None
            found_s = torch.from_numpy(np.ascontiguousarray(f, np.int32))
        lap("export")
        total = sum(v_recv)
        sdf_r = torch.empty((total,), dtype=torch.float32, device=dev)
        wgt_r = torch.empty((total,), dtype=torch.float32, device=dev)
        found_r = torch.empty((sum(n_recv),), dtype=torch.int32, device=dev)
        self._all_to_all(sdf_r, sdf_s, v_recv, v_send)
        self._all_to_all(wgt_r, wgt_s, v_recv, v_send)
        self._all_to_all(found_r, found_s, n_recv, n_send)
        col_r = None
        if col_s is not None:
            col_r = torch.empty((total, 4), dtype=torch.uint8, device=dev)
            self._all_to_all(col_r, col_s, v_recv, v_send)
        lap("all_to_all x 3-4 (issue)")
        if len(recv_items):
            if on_gpu:
                self._order_map_after_collectives()
                self.map.ImportGhostShells(recv_items, sdf_r, wgt_r, col_r, found_r)
            else:
                self.map.ImportGhostShells(recv_items, sdf_r.numpy(), wgt_r.numpy(), None if col_r is None else col_r.numpy(), found_r.numpy())
        # ---- 4.
        lap("import")
        self.map.UpdateMeshesOf(jobs)
        self.map.DropGhostChunks()
        lap("recompute + drop (issue)")
        if timing:
            self.phase_us["recomputes"] = self.phase_us.get("recomputes", 0) + 1
        per_voxel = 12 if col_r is not None else 8
        # for the record: what whole ghost chunks (the round-2 protocol) would have moved for the same ghosts
        self.whole_chunk_bytes = getattr(self, "whole_chunk_bytes", 0) + (len(np.unique(recv_items[:, :3], axis=0)) if len(recv_items) else 0) * edge ** 3 * per_voxel
        self.shell_bytes = getattr(self, "shell_bytes", 0) + int(total * per_voxel)
        return int(total * per_voxel)

    def _order_after_map(self):
        """the collective queued next on torch's current stream starts after what the map has queued so far (an event, no host wait)"""
        torch = self.x.torch
        ev = self._ev = getattr(self, "_ev", None) or torch.cuda.Event()
        ev.record(torch.cuda.current_stream())  # (a first record creates the hipEvent_t behind the torch object)
        self.map.record_event(ev.cuda_event)
        torch.cuda.current_stream().wait_event(ev)

    def _order_map_after_collectives(self):
        """the map's next call starts after the collectives queued on torch's current stream"""
        torch = self.x.torch
        ev = self._ev2 = getattr(self, "_ev2", None) or torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.map.wait_event(ev.cuda_event)

    def NumChunks(self):
        torch, dist = self.x.torch, self.x.dist
        n = torch.tensor([self.map.NumChunks()], dtype=torch.int64, device=self.x.recv[0].device)
        if self.x.world > 1:
            dist.all_reduce(n)
        return int(n.item())

    def GatherChunkIDs(self):
        ids = np.asarray(self.map.GetChunkIDs(), np.int32).reshape(-1, 3)
        if self.x.world == 1:
            return ids
        out = [None] * self.x.world
        self.x.dist.all_gather_object(out, ids)
        return np.concatenate(out, axis=0)
