"""Multi-process (gloo, world_size 2, CPU) tests of the sharded path's host logic: frame exchange, ownership, gathers.

No GPU: the local shard map is a recording stand-in (tests only), the ownership function is the product's
(chisel_hip_chunk_owner is a pure host function of libchisel_hip.so)."""
import os
import socket

import numpy as np
import pytest

from cvids_amd import synth

W, H, K = 32, 24, 4


class RecordingMap:
    """Stands in for cvids_amd.chisel.Chisel: records what IntegrateBatch received, owns ids by the product's ownership function."""

    def __init__(self, rank, world):
        self.rank, self.world, self.calls, self.syncs = rank, world, [], 0

    def synchronize(self):  # the synchronous batch path waits for the map before refilling a receive buffer
        self.syncs += 1

    def IntegrateBatch(self, integrator, frames, colors=None):
        self.calls.append(([(np.asarray(d.cpu()).copy(), np.asarray(p).copy(), (c.fx, c.fy, c.cx, c.cy, c.near_plane, c.far_plane, c.width, c.height))
                            for d, p, c in frames],
                           None if colors is None else [np.asarray(c.cpu()).copy() for c, _, _ in colors]))

    def NumChunks(self):
        return len(self.GetChunkIDs())

    def GetChunkIDs(self):
        from cvids_amd.chisel import chunk_owner
        ids = [(x, y, z) for x in range(-3, 3) for y in range(-2, 2) for z in range(0, 4)]
        return np.array([i for i in ids if chunk_owner(i, self.world, 2) == self.rank], np.int32).reshape(-1, 3)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, with_color, out):
    import torch
    import torch.distributed as dist
    from cvids_amd.chisel import PinholeCamera
    from cvids_amd.sharded import FrameExchange, ShardedChisel, frames_of_rank
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        intr = synth.intrinsics(W, H)
        cam = PinholeCamera(*intr, W, H, 0.05, 5.0)
        frames = list(synth.stream("sphere_room", 2 * K, W, H))
        color = synth.render_color(W, H, 3)
        x = FrameExchange(W, H, K, torch.device("cpu"), dist, channels=3 if with_color else 0)
        local = RecordingMap(rank, world)
        sm = ShardedChisel(local, x, integrator=None)
        for b in range(2):
            mine = [b * K + j for j in frames_of_rank(K, world, rank)]
            depth = torch.from_numpy(np.stack([frames[i][0] for i in mine]))
            col = torch.from_numpy(np.stack([color + np.uint8(i) for i in mine])) if with_color else None
            sm.IntegrateBatch(depth, [frames[i][1] for i in mine], [cam] * len(mine), col)
        # every rank must have received all frames of both batches, in frame order, bit for bit
        for b in range(2):
            got, gotc = local.calls[b]
            assert len(got) == K
            for j in range(K):
                d, p, c = got[j]
                assert np.array_equal(d.view(np.uint32), frames[b * K + j][0].view(np.uint32))
                assert np.array_equal(p[:3], np.asarray(frames[b * K + j][1])[:3])
                f32 = lambda v: float(np.float32(v))  # the payload carries fp32, like the C ABI's chisel_hip_depth_frame
                assert c == (f32(cam.fx), f32(cam.fy), f32(cam.cx), f32(cam.cy), f32(cam.near_plane), f32(cam.far_plane), W, H)
                if with_color:
                    assert np.array_equal(gotc[j], color + np.uint8(b * K + j))
        n_total = sm.NumChunks()
        ids = sm.GatherChunkIDs()
        out.put((rank, n_total, sorted(map(tuple, ids.tolist())), len(local.GetChunkIDs())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("with_color", [False, True])
def test_frame_exchange_and_gathers_world2(hip_lib, with_color):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, with_color, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(2))
    all_ids = sorted((x, y, z) for x in range(-3, 3) for y in range(-2, 2) for z in range(0, 4))
    for rank, n_total, ids, n_local in res:
        assert n_total == len(all_ids) and ids == all_ids  # the shards partition the id set: complete and disjoint
    assert res[0][3] + res[1][3] == len(all_ids) and min(res[0][3], res[1][3]) > len(all_ids) // 4  # and roughly balanced


def test_ownership_is_a_partition(hip_lib):
    from cvids_amd.chisel import chunk_owner
    rng = np.random.default_rng(5)
    ids = rng.integers(-500, 500, (4000, 3))
    for world in (1, 2, 4, 8):
        owners = np.array([chunk_owner(i, world, 2) for i in ids])
        assert owners.min() >= 0 and owners.max() < world
        if world > 1:
            counts = np.bincount(owners, minlength=world)
            assert counts.min() > 0.6 * len(ids) / world  # no starved shard
    # super-blocks: the 2x2x2 chunks of one block share an owner
    for base in ((0, 0, 0), (-2, 4, -6), (10, -8, 2)):
        o = {chunk_owner((base[0] + a, base[1] + b, base[2] + c), 8, 2) for a in (0, 1) for b in (0, 1) for c in (0, 1)}
        assert len(o) == 1


def test_frames_of_rank():
    from cvids_amd.sharded import frames_of_rank
    for world in (1, 2, 4, 8):
        got = sum((frames_of_rank(8, world, r) for r in range(world)), [])
        assert got == list(range(8))
    with pytest.raises(ValueError):
        frames_of_rank(6, 4, 0)


# ---- meshing a sharded map: the exchange protocol (no voxels are computed here) ------------------------------------------
def test_mesh_plan_covers_every_foreign_neighbour(hip_lib):
    from cvids_amd.chisel import chunk_owner
    from cvids_amd.sharded import mesh_plan
    world = 4
    owner = lambda i: chunk_owner(i, world, 2)
    rng = np.random.default_rng(11)
    union = {tuple(int(v) for v in i) for i in rng.integers(-6, 6, (200, 3))}
    seen_jobs = set()
    for rank in range(world):
        jobs, requests = mesh_plan(union, rank, world, owner)
        assert all(owner(j) == rank for j in jobs) and jobs == sorted(jobs)
        seen_jobs.update(jobs)
        asked = {i for ids in requests.values() for i in ids}
        for o, ids in requests.items():
            assert o != rank and all(owner(i) == o for i in ids) and len(set(ids)) == len(ids)
        for x, y, z in jobs:  # every neighbour of a job is owned or asked for
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    for dz in (-1, 0, 1):
                        n = (x + dx, y + dy, z + dz)
                        assert owner(n) == rank or n in asked
    assert seen_jobs == union  # the shards' jobs partition the union


class MeshRecordingMap(RecordingMap):
    """adds the mesh-side calls: resident chunks = owned ids of a fixed box; a voxel of a chunk's shell encodes the chunk's id, its
    owner and the voxel's coordinates, so that a ghost's box can be checked voxel by voxel"""
    EDGE = 8

    def __init__(self, rank, world):
        super().__init__(rank, world)
        self.chunk_size = (self.EDGE,) * 3
        self.ghosts, self.meshed, self.dropped = {}, None, 0

    def _resident(self):
        return set(map(tuple, self.GetChunkIDs().tolist()))

    def DirtyEntries(self):
        dirty = [i for i in sorted(self._resident()) if (i[0] + i[1] + i[2]) % 3 == 0]
        return np.array([list(i) + [0] for i in dirty], np.int32).reshape(-1, 4)

    @staticmethod
    def _box(code, n):
        rng = lambda c: range(n) if c == 0 else (range(0, 2) if c == 1 else (range(n - 1, n) if c == 2 else (0, 1, n - 1)))
        return [(x, y, z) for z in rng((code >> 4) & 3) for y in rng((code >> 2) & 3) for x in rng(code & 3)]

    @classmethod
    def voxel_value(cls, cid, owner, v):
        return float(((cid[0] * 31 + cid[1]) * 31 + cid[2]) * 1000 + owner * 512 + (v[2] * cls.EDGE + v[1]) * cls.EDGE + v[0])

    # the device protocol of ShardedChisel.UpdateMeshes (chisel_hip_shell_plan_device ...) restated with numpy: cvids_amd/sharded.py
    use_color = False

    def PlanShellsDevice(self, gathered, world, cap):
        from cvids_amd.chisel import chunk_owner
        from cvids_amd.sharded import plan_shells_reference, shell_box_coords
        g = np.asarray(gathered).reshape(world, 1 + 4 * cap)
        mx = int(g[:, 0].max())
        if mx > cap:
            return {"jobs": 0, "max_count": mx, "send_items": 0, "send": np.zeros((world, 2), np.int64), "recv": np.zeros((world, 2), np.int64)}
        entries = np.concatenate([g[r, 1:1 + 4 * int(g[r, 0])].reshape(-1, 4) for r in range(world)], axis=0)
        jobs, send, recv = plan_shells_reference(entries, world, self.rank, lambda i: chunk_owner(i, world, 2))
        self._plan = (jobs, send, recv)
        count = lambda d, p: (len(d.get(p, [])), sum(len(shell_box_coords(it[3], self.EDGE)) for it in d.get(p, [])))
        return {"jobs": len(jobs), "max_count": mx, "send_items": sum(len(v) for v in send.values()),
                "send": np.array([count(send, p) for p in range(world)], np.int64), "recv": np.array([count(recv, p) for p in range(world)], np.int64)}

    def ShellSegmentBytes(self, items, voxels):
        from cvids_amd.sharded import segment_bytes
        return segment_bytes(int(items), int(voxels), False)

    def _segments(self):
        """one byte segment per peer, as the device's export kernel lays them out"""
        from cvids_amd.sharded import pack_segment, shell_box_coords
        res = self._resident()
        _, send, _ = self._plan
        segs = []
        rng = np.random.default_rng(self.rank)
        for p in range(self.world):
            items = list(send.get(p, []))
            rng.shuffle(items)  # (the device emits them in the order its atomics happen to run: the receiver must not care)
            sdf, found, first, at = [], [], [], 0
            for x, y, z, code in items:
                box = shell_box_coords(code, self.EDGE)
                found.append(1 if (x, y, z) in res else 0)
                first.append(at)
                sdf.extend(self.voxel_value((x, y, z), self.rank, v) for v in box)
                at += len(box)
            sdf = np.array(sdf, np.float32)
            seg = bytearray(pack_segment(np.array(items, np.int32).reshape(-1, 4), np.array(found, np.int32), sdf, sdf + np.float32(0.5), None))
            rec = np.frombuffer(seg, np.int32, 8 * len(items), 16).reshape(-1, 8)
            rec[:, 5] = first
            segs.append(bytes(seg))
        return segs

    def ExportShellsPacked(self, out):
        blob = b"".join(self._segments())
        assert len(blob) == out.numel()
        out.copy_(__import__("torch").from_numpy(np.frombuffer(blob, np.uint8).copy()))

    # ---- the wait-free form (chisel_hip_shell_plan_queue ...) restated: fixed strides, a status vector, and nothing at all behind the
    # exchange when the all-reduced status calls the recompute off
    force_overflow = False  # a test's way to make ONE rank report a segment that does not fit

    def PlanShellsQueue(self, gathered, world, cap, stride, status, out, send_items_hint=0):
        from cvids_amd.sharded import segment_bytes
        plan = self.PlanShellsDevice(gathered, world, cap)
        cut = plan["max_count"] > cap
        if cut:
            self._plan = ([], {}, {})
        largest = max(segment_bytes(int(i), int(v), False) for i, v in plan["send"])
        flags = (1 if cut else 0) | (4 if (largest > stride or self.force_overflow) else 0)
        vals = [flags, plan["max_count"], largest, plan["jobs"], int(plan["recv"][:, 0].sum()), plan["send_items"], int(plan["recv"][:, 1].sum()), 0]
        status.copy_(__import__("torch").tensor(vals, dtype=status.dtype))
        self.queued_plans = getattr(self, "queued_plans", 0) + 1
        blob = b""
        for seg in self._segments():
            if len(seg) > stride:
                seg = np.array([0, 0, 1, 0], np.int32).tobytes()
            blob += seg + bytes(stride - len(seg))
        assert len(blob) == out.numel() == self.world * stride
        out.copy_(__import__("torch").from_numpy(np.frombuffer(blob, np.uint8).copy()))

    def ImportShellsFixed(self, buf, stride, status, jobs_hint=0, items_hint=0):
        from cvids_amd.sharded import segment_bytes, shell_box_coords, unpack_segment
        self._called_off = int(status[0]) != 0
        if self._called_off:
            return
        self.ghosts_created = getattr(self, "ghosts_created", 0)
        raw = np.asarray(buf)
        _, _, recv = self._plan
        for o in range(self.world):
            seg = raw[o * stride:(o + 1) * stride]
            head = np.frombuffer(seg.tobytes()[:16], np.int32)
            assert head[2] == 0  # (the status would have called the recompute off)
            rec, sdf, wgt, _ = unpack_segment(seg[:segment_bytes(int(head[0]), int(head[1]), False)], False)
            assert sorted(map(tuple, rec[:, :4].tolist())) == sorted(recv.get(o, []))
            for x, y, z, code, found, first, _, _ in rec.tolist():
                if not found:
                    continue
                cells = self.ghosts.setdefault((x, y, z), {})
                for k, v in enumerate(shell_box_coords(code, self.EDGE)):
                    cells[v] = (float(sdf[first + k]), float(wgt[first + k]))

    def ShellCommit(self, aborted):
        self.commits = getattr(self, "commits", []) + [bool(aborted)]
        assert bool(aborted) == getattr(self, "_called_off", False)
        self._called_off = False

    def ImportShellsPacked(self, buf):
        from cvids_amd.sharded import segment_bytes, shell_box_coords, unpack_segment
        _, _, recv = self._plan
        raw = np.asarray(buf)
        at = 0
        for o in range(self.world):
            n = len(recv.get(o, []))
            vox = sum(len(shell_box_coords(it[3], self.EDGE)) for it in recv.get(o, []))
            size = segment_bytes(n, vox, False)
            rec, sdf, wgt, _ = unpack_segment(raw[at:at + size], False)
            at += size
            assert sorted(map(tuple, rec[:, :4].tolist())) == sorted(recv.get(o, []))  # what this rank counted on is what the owner sent
            for x, y, z, code, found, first, _, _ in rec.tolist():
                if not found:
                    continue
                cells = self.ghosts.setdefault((x, y, z), {})
                for k, v in enumerate(shell_box_coords(code, self.EDGE)):
                    cells[v] = (float(sdf[first + k]), float(wgt[first + k]))
        assert at == len(raw)

    def UpdateMeshesPlanned(self):
        if getattr(self, "_called_off", False):
            return
        self.mesh_steps = getattr(self, "mesh_steps", 0) + 1
        self.meshed = (list(self._plan[0]), dict(self.ghosts))

    def UpdateMeshesOf(self, ids):
        self.meshed = ([tuple(int(v) for v in i) for i in np.asarray(ids).reshape(-1, 3)], dict(self.ghosts))

    def DropGhostChunks(self):
        if getattr(self, "_called_off", False):
            return
        self.dropped += 1
        self.ghosts = {}


def _mesh_worker(rank, world, port, out, dirty_cap=None):
    import torch
    import torch.distributed as dist
    from cvids_amd.sharded import FrameExchange, ShardedChisel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = FrameExchange(W, H, K, torch.device("cpu"), dist, channels=0)
        local = MeshRecordingMap(rank, world)
        sm = ShardedChisel(local, x, integrator=None)
        if dirty_cap is not None:
            sm._dirty_cap = dirty_cap  # far too small: the gathered tensor overflows and every rank takes the same second turn
        nbytes = sm.UpdateMeshes(force=True)
        if dirty_cap is not None:
            assert sm._dirty_cap > dirty_cap
        jobs, ghosts = local.meshed
        out.put((rank, jobs, ghosts, local.dropped, len(local.ghosts), nbytes))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dirty_cap", [None, 4])
def test_sharded_update_meshes_protocol_world2(hip_lib, dirty_cap):
    """The shell protocol between two processes (gloo): every rank meshes the owned part of the union of the 27-neighbourhoods, every
    neighbour another rank holds arrives as a ghost with exactly the voxel box its position asks for -- coordinates {0, 1} where the
    ghost lies on the + side of a job, {N - 1} on the - side, all along an axis they share -- carrying the owner's values, and far
    fewer bytes travel than whole chunks would take."""
    import torch.multiprocessing as mp
    from cvids_amd.chisel import chunk_owner
    from cvids_amd.sharded import plan_shells_reference
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mesh_worker, args=(r, world, port, q, dirty_cap)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=170) for _ in range(world)), key=lambda r: r[0])  # before the joins: a child blocks in put() until its data is read
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    E = MeshRecordingMap.EDGE
    box = [(x, y, z) for x in range(-3, 3) for y in range(-2, 2) for z in range(0, 4)]
    resident = {r: {i for i in box if chunk_owner(i, world, 2) == r} for r in range(world)}
    union = set()
    for r in range(world):
        for x, y, z, _ in MeshRecordingMap(r, world).DirtyEntries().tolist():
            union.update((x + dx, y + dy, z + dz) for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1))
    all_jobs = set()
    for rank, jobs, ghosts, dropped, left, nbytes in res:
        assert dropped == 1 and left == 0
        assert set(jobs) == {i for i in union if chunk_owner(i, world, 2) == rank}
        all_jobs.update(jobs)
        need = {}  # ghost -> set of voxel coordinates some job of this rank reads
        for x, y, z in jobs:
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    for dz in (-1, 0, 1):
                        n = (x + dx, y + dy, z + dz)
                        o = chunk_owner(n, world, 2)
                        if o == rank or n not in resident[o]:
                            continue
                        rng = lambda d: range(E) if d == 0 else (range(0, 2) if d > 0 else range(E - 1, E))
                        need.setdefault(n, set()).update((vx, vy, vz) for vx in rng(dx) for vy in rng(dy) for vz in rng(dz))
        assert set(ghosts) == set(need)  # every neighbour another shard holds has arrived, nothing else
        total_vox = 0
        for g, cells in ghosts.items():
            o = chunk_owner(g, world, 2)
            assert need[g] <= set(cells)  # the boxes cover what the jobs read
            assert len(cells) <= 1.5 * len(need[g])  # and little more
            for v, (sd, wg) in cells.items():  # with the owner's values
                assert sd == np.float32(MeshRecordingMap.voxel_value(g, o, v)) and wg == np.float32(np.float32(MeshRecordingMap.voxel_value(g, o, v)) + np.float32(0.5))
            total_vox += len(cells)
        assert nbytes >= 8 * total_vox  # (boxes of chunks that turned out not to be resident travel too)
        n_ghost_ids = len({it[:3] for o in range(world) for it in plan_shells_reference(
            np.concatenate([MeshRecordingMap(r, world).DirtyEntries() for r in range(world)]), world, rank, lambda i: chunk_owner(i, world, 2))[2].get(o, [])})
        assert nbytes < 0.7 * 8 * E ** 3 * n_ghost_ids  # less than whole ghost chunks would take, even with two shards, 8^3 chunks and unmerged boxes
        assert all(chunk_owner(g, world, 2) != rank for g in ghosts)
    assert all_jobs == union


def _wait_free_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    from cvids_amd.sharded import FrameExchange, ShardedChisel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = FrameExchange(W, H, K, torch.device("cpu"), dist, channels=0)
        local = MeshRecordingMap(rank, world)
        sm = ShardedChisel(local, x, integrator=None)
        snap = lambda: (sorted(local.meshed[0]), {g: dict(c) for g, c in local.meshed[1].items()})
        log = {}
        # 1. nothing to size the segments from yet: the blocking form, which leaves the ranks' sizes behind
        sm.UpdateMeshes(force=True, wait_free=True)
        assert getattr(sm, "wait_free_recomputes", 0) == 0 and local.mesh_steps == 1
        first = snap()
        sm.Settle()
        assert sm._est["seg_bytes"] > 0 and sm._est["jobs"] > 0
        # 2. the wait-free form: the same meshes from the same ghosts, nothing read in between, one commit
        nbytes = sm.UpdateMeshes(force=True, wait_free=True)
        assert sm.wait_free_recomputes == 1 and local.queued_plans == 1 and local.mesh_steps == 2 and nbytes > 0
        sm.Settle()
        assert snap() == first and local.commits == [False] and local.dropped == 2 and not local.ghosts
        # 3. segments far too small on every rank: called off (no mesh step, no ghost, no drop); Settle makes it again, still wait-free, in
        # slots of exactly the size the status reported
        sm._est["seg_bytes"] = 64
        sm.UpdateMeshes(force=True, wait_free=True)
        assert sm.wait_free_recomputes == 2 and local.mesh_steps == 2 and local.dropped == 2 and not local.ghosts
        sm.NumChunks()  # (any method of the sharded map settles first)
        assert sm.wait_free_aborts == 1 and sm.last_abort == 4 and sm.wait_free_retries == 1 and sm.wait_free_recomputes == 3
        assert local.commits == [False, True, False] and local.mesh_steps == 3 and local.dropped == 3 and snap() == first
        # 4. ONE rank whose segment does not fit: the all-reduce tells the other, both call the recompute off and make it again
        assert sm._est["seg_bytes"] > 64
        local.force_overflow = rank == 1
        sm.UpdateMeshes(force=True, wait_free=True)
        assert sm.wait_free_recomputes == 4 and local.mesh_steps == 3
        local.force_overflow = False
        sm.Settle()
        assert sm.wait_free_aborts == 2 and sm.wait_free_retries == 2 and local.mesh_steps == 4 and snap() == first and local.commits == [False, True, False, True, False]
        # 5. a dirty list that outgrows the gathered tensor: called off as well, and made again in the blocking form, which grows the tensor
        sm._dirty_cap = 4
        sm._est["max_count"] = 1
        sm.UpdateMeshes(force=True, wait_free=True)
        sm.Settle()
        assert sm.wait_free_aborts == 3 and (sm.last_abort & 1) and sm.wait_free_retries == 2 and sm._dirty_cap > 4 and snap() == first
        # 6. and on it goes
        sm.Settle()
        sm.UpdateMeshes(force=True, wait_free=True)
        sm.Settle()
        assert sm.wait_free_recomputes == 7 and sm.wait_free_aborts == 3 and snap() == first
        # 7. Reset: the sizes go with the map they described -- the next recompute takes the blocking form again, the one after it the wait-free one
        local.Reset = lambda: None
        sm.Reset()
        assert sm._est is None
        plans = local.queued_plans
        sm.UpdateMeshes(force=True, wait_free=True)
        assert local.queued_plans == plans and sm.wait_free_recomputes == 7
        sm.UpdateMeshes(force=True, wait_free=True)
        sm.Settle()
        assert local.queued_plans == plans + 1 and sm.wait_free_recomputes == 8 and sm.wait_free_aborts == 3 and snap() == first
        out.put((rank, True))
    finally:
        dist.destroy_process_group()


def test_wait_free_sharded_recompute_world2(hip_lib):
    """ShardedChisel.UpdateMeshes(wait_free=True) between two processes (gloo), against the blocking form: same jobs, same ghosts, same
    values; a recompute whose segments (on both ranks, on one rank only) or whose dirty list do not fit is called off on every rank before
    anything has happened and made again by Settle()."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wait_free_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=170) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]


def test_mesh_shell_plan_is_consistent_across_ranks(hip_lib):
    """chisel_hip_mesh_shell_plan: jobs partition the union, one item per ghost, sorted by owner then id, and what rank q plans to ask of
    rank r is what r works out for q (the request lists are never exchanged)."""
    from cvids_amd.chisel import chunk_owner, mesh_shell_plan, shell_volume
    rng = np.random.default_rng(11)
    for world in (2, 4, 8):
        dirty = np.unique(rng.integers(-6, 6, (60, 3)), axis=0)
        entries = np.concatenate([dirty, np.zeros((len(dirty), 1), np.int64)], axis=1).astype(np.int32)
        entries = np.concatenate([entries, np.array([[20, 20, 20, 1]], np.int32)])  # an id taken as it is
        union = {(x + dx, y + dy, z + dz) for x, y, z in dirty.tolist() for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)} | {(20, 20, 20)}
        seen = set()
        for rank in range(world):
            jobs, items = mesh_shell_plan(entries, world, rank)
            jl = [tuple(j) for j in jobs.tolist()]
            assert jl == sorted(jl) and all(chunk_owner(j, world, 2) == rank for j in jl)
            seen.update(jl)
            keys = [tuple(i) for i in items.tolist()]
            assert keys == sorted(keys) and len(set(keys)) == len(keys)
            for o, x, y, z, box in items.tolist():
                assert o != rank and chunk_owner((x, y, z), world, 2) == o
                assert 0 <= box < 64 and 1 <= shell_volume(box, 16) <= 4096
            asked = {tuple(i[1:4]) for i in items.tolist()}
            for x, y, z in jl:
                for dx in (-1, 0, 1):
                    for dy in (-1, 0, 1):
                        for dz in (-1, 0, 1):
                            n = (x + dx, y + dy, z + dz)
                            assert chunk_owner(n, world, 2) == rank or n in asked
        assert seen == union
    assert [shell_volume(b, 16) for b in (0, 1, 2, 3, 1 | (1 << 2), 2 | (2 << 2) | (2 << 4), 1 | (2 << 2) | (0 << 4))] == [4096, 512, 256, 768, 64, 1, 32]


class CloudRecordingMap:
    def __init__(self):
        self.clouds = []

    def IntegratePointCloud(self, integrator, cloud, extrinsic, truncation, max_dist):
        pts, cols = cloud
        self.clouds.append((np.asarray(pts).copy(), None if cols is None else np.asarray(cols).copy(), np.asarray(extrinsic).copy(),
                            truncation, max_dist))


def _cloud_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    from cvids_amd.sharded import FrameExchange, ShardedChisel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = FrameExchange(W, H, K, torch.device("cpu"), dist, channels=0)
        local = CloudRecordingMap()
        sm = ShardedChisel(local, x, integrator=None)
        intr = synth.intrinsics(W, H)
        for k, src in enumerate([0, 1, 1]):
            pose = synth.trajectory_pose(k)
            if rank == src:
                depth = synth.render_depth("sphere_room", pose, intr, W, H)
                pts, cols = synth.depth_to_cloud(depth, intr, 0.6, colors=True)
                if k == 2:
                    cols = None
                sm.IntegratePointCloud(torch.from_numpy(pts), None if cols is None else torch.from_numpy(cols), pose, 0.1, 5.0, src=src)
            else:
                sm.IntegratePointCloud(None, None, None, 0.1, 5.0, src=src)
        import hashlib
        digest = lambda a: None if a is None else hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()  # small: the parent reads the queue after join()
        out.put((rank, [(digest(c[0]), digest(c[1]), digest(c[2]), c[3], c[4]) for c in local.clouds]))
    finally:
        dist.destroy_process_group()


def test_sharded_pointcloud_broadcast_world2(hip_lib):
    """ShardedChisel.IntegratePointCloud: the cloud ingested by one rank reaches every rank bit for bit (points, colours, pose)."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cloud_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(world))
    assert res[0][1] == res[1][1] and len(res[0][1]) == 3
    intr = synth.intrinsics(W, H)
    for k, rec in enumerate(res[0][1]):
        pose = synth.trajectory_pose(k)
        pts, cols = synth.depth_to_cloud(synth.render_depth("sphere_room", pose, intr, W, H), intr, 0.6, colors=True)
        import hashlib
        digest = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()
        assert rec[0] == digest(pts) and rec[2] == digest(np.asarray(pose, np.float32)[:3, :4])
        assert rec[1] == (None if k == 2 else digest(cols))


def test_cull_space_enumerates_exactly_the_owned_ids(hip_lib):
    """CullSpace (kernels_cull.h): the candidate slots cull_kernel gives a shard hold every id of the range that the shard owns,
    each once, and nothing else; the slot count is about 1 / n of the range (evaluated on the host, no GPU)."""
    import ctypes as C
    from cvids_amd.chisel import chunk_owner
    L = hip_lib
    i32p = C.POINTER(C.c_int)
    L.chisel_hip_debug_cull_space.argtypes = [i32p, i32p, C.c_int, C.c_int, C.c_int, i32p, C.c_int, i32p]
    rng = np.random.default_rng(4)
    for trial in range(12):
        lo = rng.integers(-40, 20, 3).astype(np.int32)
        dim = rng.integers(1, 23, 3).astype(np.int32)
        n = int(rng.choice([1, 2, 3, 4, 8]))
        box = [(x, y, z) for x in range(lo[0], lo[0] + dim[0]) for y in range(lo[1], lo[1] + dim[1]) for z in range(lo[2], lo[2] + dim[2])]
        seen = set()
        slots = 0
        for rank in range(n):
            cap = len(box) + 8
            ids = np.zeros((cap, 3), np.int32)
            cnt = C.c_int(0)
            total = L.chisel_hip_debug_cull_space(lo.ctypes.data_as(i32p), dim.ctypes.data_as(i32p), n, rank, 2, ids.ctypes.data_as(i32p), cap,
                                                  C.byref(cnt))
            got = [tuple(int(v) for v in r) for r in ids[:cnt.value]]
            assert len(set(got)) == len(got), "an id twice"
            assert set(got) == {i for i in box if chunk_owner(i, n, 2) == rank}, (trial, n, rank)
            assert not (seen & set(got))
            seen.update(got)
            slots += total
        assert seen == set(box)
        if n > 1:
            assert slots <= len(box) + 8 * n * (dim[1] // 2 + 2) * (dim[2] // 2 + 2) * 2   # partial super-blocks and the rounded-up rows only


def test_unmerged_device_plan_covers_what_the_merging_host_planner_asks_for(hip_lib):
    """The plan of rounds 5-6 (cvids_amd.sharded.plan_shells_reference = what chisel_hip_shell_plan_device computes: one item per (job,
    direction), less -- since round 6 -- the items whose box is part of another item's of the same (rank, ghost)) against
    chisel_hip_mesh_shell_plan (the host planner of rounds 3-4: a box another one contains is dropped, the two ends of one axis become one
    box): the same jobs, the same ghosts per owner, and per ghost the same set of voxels -- the items that are left may still overlap,
    their union is what the merged boxes hold, and none of them lies inside another.  Scattered and contiguous dirty sets, entries of both flags."""
    from cvids_amd.chisel import chunk_owner, mesh_shell_plan
    from cvids_amd.sharded import plan_shells_reference, shell_box_coords
    rng = np.random.default_rng(11)
    shell = rng.normal(size=(120, 3))
    shell = np.unique(np.floor(shell / np.linalg.norm(shell, axis=1)[:, None] * 6.3).astype(np.int32), axis=0)
    slab = np.stack(np.meshgrid(np.arange(-3, 4), np.arange(-2, 3), [5, 6], indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
    E = 8
    for ids in (shell, slab, np.array([[0, 0, 0]], np.int32)):
        ent = np.concatenate([ids, np.zeros((len(ids), 1), np.int32)], 1)
        ent[::5, 3] = 1
        for world in (2, 3, 8):
            owner = lambda i: chunk_owner(i, world, 2)
            for r in range(world):
                rj, ri = mesh_shell_plan(ent, world, r, 2)
                jobs, send, recv = plan_shells_reference(ent, world, r, owner)
                assert [tuple(j) for j in rj.tolist()] == jobs
                merged, plain = {}, {}
                for o, x, y, z, box in ri.tolist():
                    merged.setdefault((o, x, y, z), set()).update(shell_box_coords(box, E))
                for o, items in recv.items():
                    for x, y, z, box in items:
                        plain.setdefault((o, x, y, z), set()).update(shell_box_coords(box, E))
                assert merged == plain, (world, r)
                # no item of a ghost lies inside another item of the same ghost (what shell_item_covered leaves out)
                per_ghost = {}
                for o, items in recv.items():
                    for x, y, z, box in items:
                        per_ghost.setdefault((x, y, z), []).append(set(shell_box_coords(box, E)))
                for boxes in per_ghost.values():
                    for a in range(len(boxes)):
                        assert not any(a != b and boxes[a] <= boxes[b] for b in range(len(boxes)))
                # what r plans to receive from o is what o plans to send r
                for o in range(world):
                    if o != r:
                        assert sorted(recv.get(o, [])) == sorted(plan_shells_reference(ent, world, o, owner)[1].get(r, []))
