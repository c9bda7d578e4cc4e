"""GPU parity of the marching-cubes path: chisel_hip_update_meshes & friends against the CPU oracle.

The mesh kernel orders its output with a prefix sum over the cubes in the reference's own traversal order, so the
vertex / normal / colour / grid arrays of every chunk must equal the oracle's element for element (bit-exact)."""
import os

import numpy as np
import pytest

from cvids_amd import synth
from tests.common import make_frames, small_camera, triangle_multiset
from tests.test_gpu_parity import _mk

pytestmark = pytest.mark.gpu


def _integrate(om, gm, integ, frames, cam, color_img):
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    for depth, pose in frames:
        if color_img is None:
            om.integrate_depth(depth, pose, intr, cam.near_plane, cam.far_plane)
            gm.IntegrateDepthScan(integ, depth, pose, cam)
        else:
            om.integrate_depth_color(depth, pose, intr, color_img, near=cam.near_plane, far=cam.far_plane)
            gm.IntegrateDepthScanColor(integ, depth, pose, cam, color_img, pose, cam)


def _compare_meshes(om, gm, color):
    oids = sorted(map(tuple, om.mesh_ids().tolist()))
    gids = sorted(map(tuple, gm.GetMeshIDs().tolist()))
    assert oids == gids, "mesh id sets differ: %d vs %d" % (len(oids), len(gids))
    nv = 0
    for cid in oids:
        a, b = om.get_mesh(cid), gm.GetMesh(cid)
        for key in ("vertices", "normals", "grids") + (("colors",) if color else ()):
            x, y = np.asarray(a[key]), np.asarray(b[key])
            assert x.shape == y.shape, "chunk %s %s: %s vs %s" % (cid, key, x.shape, y.shape)
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), "chunk %s: %s differ (max %g)" % (
                cid, key, np.abs(x - y).max() if x.size else 0)
        assert triangle_multiset(a["vertices"]) == triangle_multiset(b["vertices"])
        nv += len(a["vertices"])
    return len(oids), nv


@pytest.mark.parametrize("scene,color", [("wall", False), ("sphere_room", True), ("box_room", True)])
def test_mesh_parity(oracle_mod, scene, color):
    om, gm, integ = _mk(oracle_mod, 8, 0.05, color, max_chunks=8192)
    cam = small_camera(64, 48)
    cimg = synth.render_color(64, 48, 3) if color else None
    _integrate(om, gm, integ, make_frames(scene, 4, 64, 48), cam, cimg)
    om.update_meshes(force=True)
    gm.UpdateMeshes(force=True)
    n, nv = _compare_meshes(om, gm, color)
    assert n > 20 and nv > 3000
    assert len(gm.GetMeshesToUpdate()) == 0 and len(om.meshes_to_update()) == 0
    # second round: more frames, meshes of touched chunks are regenerated in place
    _integrate(om, gm, integ, make_frames(scene, 3, 64, 48, start=4), cam, cimg)
    om.update_meshes(force=True)
    gm.UpdateMeshes(force=True)
    _compare_meshes(om, gm, color)


@pytest.mark.parametrize("N,res,W,H", [(16, 0.04, 96, 72), (32, 0.02, 64, 48)])
def test_mesh_chunk_sizes(oracle_mod, N, res, W, H):
    om, gm, integ = _mk(oracle_mod, N, res, True, max_chunks=2048)
    cam = small_camera(W, H)
    _integrate(om, gm, integ, make_frames("box_room", 3, W, H), cam, synth.render_color(W, H, 3))
    om.update_meshes(force=True)
    gm.UpdateMeshes(force=True)
    n, nv = _compare_meshes(om, gm, True)
    assert n > 0 and nv > 0


@pytest.mark.parametrize("wait_free", [False, True])
@pytest.mark.parametrize("n_shards", [2, 4, 8])
def test_sharded_map_meshes_equal_the_unsharded_map(oracle_mod, n_shards, wait_free):
    """SURVEY.md 8e "meshing across shards": every shard meshes the chunks it owns with its neighbours' chunks imported as
    ghosts (export_chunks / import_ghost_chunks / update_meshes_of / drop_ghost_chunks, cvids_amd/sharded.py).  The union
    of the shards' meshes equals the oracle's meshes of the whole map element for element, two recomputes in a row, and
    the ghosts leave no trace in the shards' chunk lists."""
    from cvids_amd import chisel as ch
    from cvids_amd.sharded import LocalShardGroup
    om, gm, integ = _mk(oracle_mod, 8, 0.05, True, max_chunks=8192)
    shards = [ch.Chisel((8, 8, 8), 0.05, True, max_chunks=8192, n_shards=n_shards, shard_rank=r) for r in range(n_shards)]
    group = LocalShardGroup(shards)
    cam = small_camera(64, 48)
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    color = synth.render_color(64, 48, 3)

    class Union:  # the shards seen as one map by _compare_meshes
        def GetMeshIDs(self):
            ids = [s_.GetMeshIDs() for s_ in shards]
            return np.concatenate([i for i in ids if len(i)], axis=0) if any(len(i) for i in ids) else np.zeros((0, 3), np.int32)

        def GetMesh(self, cid):
            return shards[ch.chunk_owner(cid, n_shards, 2)].GetMesh(cid)

    # wait_free: the first recompute has nothing to size its segments from and takes the blocking form; the second is the wait-free form
    # (chisel_hip_shell_plan_queue ...: fixed segments, nothing read in between); the third gets segments of 64 bytes: called off on the
    # device -- no ghost, no mesh, no dirty flag cleared, on any shard -- and made again the blocking way
    for turn, (start, count) in enumerate(((0, 4), (4, 3), (7, 2)) if wait_free else ((0, 4), (4, 3))):
        part = make_frames("sphere_room", count, 64, 48, start=start)
        for d, p in part:
            om.integrate_depth_color(d, p, intr, color, near=cam.near_plane, far=cam.far_plane)
        for s_ in shards:
            s_.IntegrateBatch(integ, [(d, p, cam) for d, p in part], [(color, p, cam) for _, p in part])
        before = [s_.NumChunks() for s_ in shards]
        om.update_meshes(force=True)
        st = group.UpdateMeshes(force=True, wait_free=wait_free, stride=64 if turn == 2 else None)
        if wait_free:
            assert (st is None) if turn == 0 else (st[0] == (4 if turn == 2 else 0) and st[3] > 0 and st[4] > 0 and st[2] > 64)
            assert getattr(group, "wait_free_aborts", 0) == (1 if turn == 2 else 0)
        n, nv = _compare_meshes(om, Union(), True)
        assert n > 20 and nv > 3000
        assert [s_.NumChunks() for s_ in shards] == before  # ghosts dropped
        assert sum(before) == om.num_chunks()
        for r, s_ in enumerate(shards):
            assert all(ch.chunk_owner(cid, n_shards, 2) == r for cid in map(tuple, s_.GetMeshIDs().tolist()))
            assert len(s_.GetMeshesToUpdate()) == 0


@pytest.mark.parametrize("n_shards", [2, 8])
def test_device_shell_plan_equals_the_reference(n_shards):
    """chisel_hip_shell_plan_device (the plan of a sharded recompute, made on the device from the all-gathered dirty list) against its
    numpy restatement (cvids_amd.sharded.plan_shells_reference): job counts, per-peer item and voxel counts in both directions, and --
    through the exported segments -- the items themselves (as a set: their order is whatever the device's atomics produced), their
    payload offsets (disjoint, covering the segment) and the exported voxels of resident chunks."""
    import torch
    from cvids_amd import chisel as ch
    from cvids_amd.sharded import plan_shells_reference, segment_bytes, shell_box_coords, unpack_segment
    N = 8
    shards = [ch.Chisel((N, N, N), 0.05, True, max_chunks=8192, n_shards=n_shards, shard_rank=r) for r in range(n_shards)]
    integ = ch.ProjectionIntegrator(ch.InverseTruncator(2.0), ch.ConstantWeighter(1.0), 0.05, True)
    cam = small_camera(64, 48)
    color = synth.render_color(64, 48, 3)
    part = make_frames("sphere_room", 3, 64, 48)
    for s_ in shards:
        s_.IntegrateBatch(integ, [(d, p, cam) for d, p in part], [(color, p, cam) for _, p in part])
    owner = lambda i: ch.chunk_owner(i, n_shards, 2)
    dev = torch.device("cuda", 0)
    cap = 1 << 10
    gathered = torch.zeros((n_shards, 1 + 4 * cap), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for r, s_ in enumerate(shards):
        s_.DirtyIdsDevice(gathered[r])
        s_.synchronize()
    g = gathered.cpu().numpy()
    entries = np.concatenate([g[r, 1:1 + 4 * int(g[r, 0])].reshape(-1, 4) for r in range(n_shards)], axis=0)
    entries = np.concatenate([entries, np.array([[40, 40, 40, 1]], np.int32)])  # an id taken as it is (flag 1), owned by somebody
    extra = torch.from_numpy(np.array([40, 40, 40, 1], np.int32)).to(dev)
    gathered[0, 1 + 4 * int(g[0, 0]):5 + 4 * int(g[0, 0])] = extra
    gathered[0, 0] += 1
    torch.cuda.synchronize()
    assert len(entries) > 20
    for r, s_ in enumerate(shards):
        plan = s_.PlanShellsDevice(gathered.view(-1), n_shards, cap)
        jobs, send, recv = plan_shells_reference(entries, n_shards, r, owner)
        assert plan["max_count"] >= int(g[:, 0].max())
        assert plan["jobs"] == len(jobs)
        vol = lambda its: sum(len(shell_box_coords(it[3], N)) for it in its)
        for p in range(n_shards):
            assert tuple(plan["send"][p]) == (len(send.get(p, [])), vol(send.get(p, []))), (r, p)
            assert tuple(plan["recv"][p]) == (len(recv.get(p, [])), vol(recv.get(p, []))), (r, p)
        sizes = [s_.ShellSegmentBytes(*plan["send"][p]) for p in range(n_shards)]
        assert sizes == [segment_bytes(len(send.get(p, [])), vol(send.get(p, [])), True) for p in range(n_shards)]
        buf = torch.empty((sum(sizes),), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        s_.ExportShellsPacked(buf)
        s_.synchronize()
        raw = buf.cpu().numpy()
        resident = set(map(tuple, s_.GetChunkIDs().tolist()))
        at = 0
        for p in range(n_shards):
            rec, sdf, wgt, col = unpack_segment(raw[at:at + sizes[p]], True)
            at += sizes[p]
            assert sorted(map(tuple, rec[:, :4].tolist())) == sorted(send.get(p, []))
            spans = sorted((int(f), int(f) + len(shell_box_coords(int(b), N))) for f, b in zip(rec[:, 5], rec[:, 3]))
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:])) and (not spans or (spans[0][0] == 0 and spans[-1][1] == len(sdf)))
            for x, y, z, box, found, first, _, _ in rec.tolist()[:40]:
                assert found == (1 if (x, y, z) in resident else 0)
                if found:
                    cs, cw, cc = s_.GetChunk((x, y, z))
                    vox = shell_box_coords(box, N)
                    idx = np.array([(vz * N + vy) * N + vx for vx, vy, vz in vox])
                    assert np.array_equal(sdf[first:first + len(vox)], cs.reshape(-1)[idx]) and np.array_equal(wgt[first:first + len(vox)], cw.reshape(-1)[idx])
                    assert np.array_equal(col[first:first + len(vox)], cc.reshape(-1, 4)[idx].view(np.uint32).reshape(-1))
    for s_ in shards:
        s_.close()


def test_stream_with_keyframe_meshing_no_waits(oracle_mod):
    """The bench's call pattern on a small map: batches of 7 frames queued back to back, UpdateMeshes() after every frame
    count that crosses a multiple of 10 (the recompute is queued whole, its totals are looked at by the next batch), no
    wait anywhere until the end.  Voxels and every chunk's mesh arrays must equal the oracle's."""
    om, gm, integ = _mk(oracle_mod, 16, 0.04, True, max_chunks=4096)
    W, H = 160, 120
    cam = small_camera(W, H)
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    color = synth.render_color(W, H, 3)
    frames = make_frames("sphere_room", 21, W, H, agents=2, nan_fraction=0.01)  # 42 frames
    done = 0
    for lo in range(0, len(frames), 7):
        part = frames[lo:lo + 7]
        for d, p in part:
            om.integrate_depth_color(d, p, intr, color, near=cam.near_plane, far=cam.far_plane)
        gm.IntegrateBatch(integ, [(d, p, cam) for d, p in part], [(color, p, cam) for _, p in part])
        if (done + len(part)) // 10 > done // 10:
            om.update_meshes(force=True)
            gm.UpdateMeshes(force=True)
        done += len(part)
    from tests.common import compare_fields
    compare_fields(om.fields(), gm.fields(), om.V, True)
    n, nv = _compare_meshes(om, gm, True)
    assert n > 10 and nv > 1000


def test_recompute_that_outgrows_its_buffers(oracle_mod, monkeypatch):
    """CHISEL_HIP_MESH_TINY: the triangle list (256 entries) and the arena (4096 floats) are far too small, so every recompute first
    runs into the overflow flag (the speculative triangle kernel must emit nothing), grows the list, counts again and emits again
    into an arena of the right size -- with batches queued in between.  Voxels and meshes must still equal the oracle's."""
    monkeypatch.setenv("CHISEL_HIP_MESH_TINY", "1")
    om, gm, integ = _mk(oracle_mod, 16, 0.04, True, max_chunks=4096)
    W, H = 160, 120
    cam = small_camera(W, H)
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    color = synth.render_color(W, H, 3)
    frames = make_frames("sphere_room", 24, W, H)
    for lo in range(0, len(frames), 6):
        part = frames[lo:lo + 6]
        for d, p in part:
            om.integrate_depth_color(d, p, intr, color, near=cam.near_plane, far=cam.far_plane)
        gm.IntegrateBatch(integ, [(d, p, cam) for d, p in part], [(color, p, cam) for _, p in part])
        om.update_meshes(force=True)
        gm.UpdateMeshes(force=True)
    from tests.common import compare_fields
    compare_fields(om.fields(), gm.fields(), om.V, True)
    n, nv = _compare_meshes(om, gm, True)
    assert n > 10 and nv > 1000


def test_integration_queued_behind_an_unseen_recompute_is_replayed(oracle_mod, monkeypatch):
    """An integration call right after UpdateMeshes() is queued behind the recompute before the host has seen whether that fitted its
    buffers; one that did not fit sets a word on the device, the integration kernel leaves the map alone, and the host -- when it next
    looks -- emits the recompute again from the untouched map and replays the launch.  Forced here for every recompute
    (CHISEL_HIP_MESH_TINY: nothing ever fits; CHISEL_HIP_DEFER_TOTALS=2: queued unseen even when the totals are there): meshes after
    every step and voxels at the end equal the oracle's, and the statistics say the replays happened."""
    monkeypatch.setenv("CHISEL_HIP_MESH_TINY", "1")
    monkeypatch.setenv("CHISEL_HIP_DEFER_TOTALS", "2")
    om, gm, integ = _mk(oracle_mod, 16, 0.04, True, max_chunks=4096)
    W, H = 160, 120
    cam = small_camera(W, H)
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    color = synth.render_color(W, H, 3)
    frames = make_frames("sphere_room", 20, W, H)
    for lo in range(0, len(frames), 5):
        part = frames[lo:lo + 5]
        for d, p in part:
            om.integrate_depth_color(d, p, intr, color, near=cam.near_plane, far=cam.far_plane)
        gm.IntegrateBatch(integ, [(d, p, cam) for d, p in part], [(color, p, cam) for _, p in part])
        om.update_meshes(force=True)
        gm.UpdateMeshes(force=True)
        if lo == 10:
            _compare_meshes(om, gm, True)  # (a reader in the middle: settles what is queued, then reads)
    st = gm.launch_stats()
    assert st["behind_unseen_recompute"] >= 2 and st["replayed"] >= 2, st  # (steps 2 and 3; step 4 follows a reader, which settled everything)
    from tests.common import compare_fields
    compare_fields(om.fields(), gm.fields(), om.V, True)
    n, nv = _compare_meshes(om, gm, True)
    assert n > 10 and nv > 1000
    # the same stream with the totals looked at before every launch gives the same map (and no replays)
    monkeypatch.setenv("CHISEL_HIP_DEFER_TOTALS", "0")
    _, gm0, integ0 = _mk(oracle_mod, 16, 0.04, True, max_chunks=4096)
    for lo in range(0, len(frames), 5):
        part = frames[lo:lo + 5]
        gm0.IntegrateBatch(integ0, [(d, p, cam) for d, p in part], [(color, p, cam) for _, p in part])
        gm0.UpdateMeshes(force=True)
    assert gm0.launch_stats()["behind_unseen_recompute"] == 0
    compare_fields(gm0.fields(), gm.fields(), om.V, True)


def test_mesh_color_lookup_near_origin(oracle_mod):
    """10 cm voxels: InterpolateColor's integer-index lookups (ChunkManager.cpp:506-520) land inside the map"""
    om, gm, integ = _mk(oracle_mod, 8, 0.10, True, trunc=("constant", 0.3), max_chunks=8192)
    cam = small_camera(64, 48)
    cimg = synth.render_color(64, 48, 3)
    frames = [(np.full((48, 64), 0.9, np.float32), synth.pose_yaw(a)) for a in (0.0, 90.0, 180.0, 270.0)]
    _integrate(om, gm, integ, frames, cam, cimg)
    om.update_meshes(force=True)
    gm.UpdateMeshes(force=True)
    _compare_meshes(om, gm, True)


def test_update_meshes_cadence_and_carved_mesh(oracle_mod):
    """Chisel.cpp:53-58: only every 10th un-forced call recomputes; a mesh whose surface is carved away stays in the map, empty"""
    om, gm, integ = _mk(oracle_mod, 8, 0.05, False, carving=True, carving_dist=0.0)
    cam = small_camera(64, 48)
    pose = synth.pose_yaw(0.0)
    _integrate(om, gm, integ, [(np.full((48, 64), 1.2, np.float32), pose)] * 3, cam, None)
    for call in range(12):
        om.update_meshes(force=False)
        gm.UpdateMeshes(force=False)
        assert len(om.mesh_ids()) == len(gm.GetMeshIDs()), "call %d" % call
        assert len(om.meshes_to_update()) == len(gm.GetMeshesToUpdate()), "call %d" % call
        if call == 0:
            assert len(gm.GetMeshIDs()) > 0
            _compare_meshes(om, gm, False)
            _integrate(om, gm, integ, [(np.full((48, 64), 2.4, np.float32), pose)] * 4, cam, None)
        if call == 5:
            assert len(gm.GetMeshesToUpdate()) > 0  # calls 1..9 do not recompute
    _compare_meshes(om, gm, False)


def test_sdf_queries(oracle_mod):
    om, gm, integ = _mk(oracle_mod, 8, 0.05, False)
    cam = small_camera(64, 48)
    _integrate(om, gm, integ, make_frames("sphere_room", 3, 64, 48), cam, None)
    rng = np.random.default_rng(11)
    pts = rng.uniform(-3.0, 3.0, (300, 3)).astype(np.float32)
    # plus points that certainly hit observed voxels: vertices of the mesh
    om.update_meshes(force=True)
    gm.UpdateMeshes(force=True)
    verts = np.concatenate([om.get_mesh(c)["vertices"] for c in om.mesh_ids()[:20]])[:300]
    hits = 0
    for p in np.concatenate([pts, verts]):
        ok_o, d_o = om.get_sdf(p)
        ok_g, d_g = gm.GetSDF(p)
        assert ok_o == ok_g
        if ok_o:
            assert d_o == d_g
        ok_o, d_o, g_o = om.get_sdf_and_gradient(p)
        ok_g, d_g, g_g = gm.GetSDFAndGradient(p)
        assert ok_o == ok_g
        if ok_o:
            hits += 1
            assert d_o == d_g and np.array_equal(g_o.view(np.uint32), g_g.view(np.uint32))
    assert hits > 50


def _parse_ply(path):
    lines = open(path).read().split("\n")
    assert lines[0] == "ply" and lines[1] == "format ascii 1.0"
    nv = int(lines[2].split()[-1])
    end = lines.index("end_header")
    header = lines[:end + 1]
    body = lines[end + 1:end + 1 + nv]
    faces = lines[end + 1 + nv:]
    return header, body, [f for f in faces if f]


def test_save_ply(oracle_mod, tmp_path):
    om, gm, integ = _mk(oracle_mod, 8, 0.05, True)
    cam = small_camera(64, 48)
    _integrate(om, gm, integ, make_frames("box_room", 3, 64, 48), cam, synth.render_color(64, 48, 3))
    om.update_meshes(force=True)
    gm.UpdateMeshes(force=True)
    po, pg = str(tmp_path / "oracle.ply"), str(tmp_path / "hip.ply")
    assert om.save_ply(po) and gm.SaveAllMeshesToPLY(pg)
    ho, bo, fo = _parse_ply(po)
    hg, bg, fg = _parse_ply(pg)
    assert ho == hg                      # identical header (vertex / face counts, colour properties)
    assert fo == fg                      # identical face list text
    # vertex lines: same text triples, chunk order aside (the reference's order is its unordered_map's)
    tri = lambda b: sorted(tuple(b[i:i + 3]) for i in range(0, len(b), 3))
    assert tri(bo) == tri(bg)
    assert not gm.SaveAllMeshesToPLY(str(tmp_path / "no_such_dir" / "x.ply"))
