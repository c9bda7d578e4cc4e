"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bar (BASELINE.json north_star): SDF and weight within 1e-4 absolute of the CPU reference.  The kernels
reproduce the reference's fp32 operation order, so these tests demand BIT-EXACT sdf / weight / colour and
identical voxel counters; ATOL below documents the contractual tolerance and is only used at full size.
"""
import json
import os
import struct

import numpy as np
import pytest

from cvids_amd import synth
from tests.common import compare_fields, make_frames, small_camera

pytestmark = pytest.mark.gpu
ATOL = 1e-4
HERE = os.path.dirname(os.path.abspath(__file__))


def f32(hexstr):
    return struct.unpack("<f", struct.pack("<I", int(hexstr, 16)))[0]


def _mk(oracle_mod, N, res, color, trunc=("inverse", 2.0), weight=1.0, carving=True, carving_dist=0.05, max_chunks=4096):
    from cvids_amd import chisel as ch
    kinds = {"constant": (0, ch.ConstantTruncator), "inverse": (1, ch.InverseTruncator), "quadratic": (2, ch.QuadraticTruncator)}
    k, cls = kinds[trunc[0]]
    om = oracle_mod.OracleMap(N, res, color)
    om.set_integrator(k, trunc[1], weight, carving, carving_dist)
    gm = ch.Chisel((N, N, N), res, color, max_chunks=max_chunks)
    integ = ch.ProjectionIntegrator(cls(trunc[1]), ch.ConstantWeighter(weight), carving_dist, carving)
    return om, gm, integ


def _run(om, gm, integ, frames, cam, color_img=None, check_each=True, atol=0.0):
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    for i, (depth, pose) in enumerate(frames):
        if color_img is None:
            om.integrate_depth(depth, pose, intr, cam.near_plane, cam.far_plane)
            gm.IntegrateDepthScan(integ, depth, pose, cam)
        else:
            om.integrate_depth_color(depth, pose, intr, color_img, near=cam.near_plane, far=cam.far_plane)
            gm.IntegrateDepthScanColor(integ, depth, pose, cam, color_img, pose, cam)
        oc = om.counters()
        gc = gm.counters(reset=True)
        for k in ("sdf", "col", "col_sat", "probe", "carved", "updated_chunks"):
            assert oc[k] == gc[k], "frame %d counter %s: oracle %d gpu %d" % (i, k, oc[k], gc[k])
        if check_each or i == len(frames) - 1:
            assert om.num_chunks() == gm.NumChunks(), "frame %d: chunk count" % i
            compare_fields(om.fields(), gm.fields(), om.V, om.use_color, atol=atol, what="frame %d" % i)
            assert sorted(map(tuple, om.meshes_to_update().tolist())) == sorted(map(tuple, gm.GetMeshesToUpdate().tolist()))


# ---- device arithmetic against the reference-built golden vectors ---------------------------------------
def test_device_kat_truncators(hip_lib):
    import ctypes as C
    kat = json.load(open(os.path.join(HERE, "golden", "ref_kat.json")))
    depths = np.array([f32(h) for h in kat["depths"]], np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    kinds = {"constant": 0, "inverse": 1, "quadratic": 2}
    for key, vals in kat["truncation"].items():
        name, phex = key.rsplit("_", 1)
        exp = np.array([int(v, 16) for v in vals], np.uint32)
        t = np.zeros_like(depths)
        w = np.zeros_like(depths)
        if name in kinds:
            assert hip_lib.chisel_hip_kat_truncation(kinds[name], f32(phex), fp(depths), len(depths), fp(t), fp(w)) == 0
            got = t.view(np.uint32)
        elif name == "weight1_inverse":
            assert hip_lib.chisel_hip_kat_truncation(1, f32(phex), fp(depths), len(depths), fp(t), fp(w)) == 0
            got = w.view(np.uint32)
        else:
            continue
        nan = np.isnan(exp.view(np.float32))
        assert np.array_equal(np.isnan(got.view(np.float32)), nan), key
        assert np.array_equal(got[~nan], exp[~nan]), key


def test_device_kat_voxels(hip_lib):
    import ctypes as C
    kat = json.load(open(os.path.join(HERE, "golden", "ref_kat.json")))
    for seq in kat["dist_sequences"]:
        ops = np.array([[0.0 if s[0] == "c" else 1.0, f32(s[1]) if s[0] == "i" else 0.0, f32(s[2]) if s[0] == "i" else 0.0]
                        for s in seq["steps"]], np.float32)
        exp = np.array([[int(s[3], 16), int(s[4], 16)] for s in seq["steps"]], np.uint32)
        out = np.zeros((len(ops), 2), np.float32)
        assert hip_lib.chisel_hip_kat_dist(ops.ctypes.data_as(C.POINTER(C.c_float)), len(ops), out.ctypes.data_as(C.POINTER(C.c_float))) == 0
        got = out.view(np.uint32)
        nan = np.isnan(exp.view(np.float32))
        assert np.array_equal(np.isnan(out), nan)
        assert np.array_equal(got[~nan], exp[~nan])
    for seq in kat["color_sequences"]:
        a = np.array(seq, np.uint8)
        ops = np.ascontiguousarray(a[:, :4])
        out = np.zeros_like(ops)
        assert hip_lib.chisel_hip_kat_color(ops.ctypes.data_as(C.POINTER(C.c_uint8)), len(ops), out.ctypes.data_as(C.POINTER(C.c_uint8))) == 0
        assert np.array_equal(out, a[:, 4:])


def test_device_color_fast_path_exhaustive(hip_lib):
    """the division-free colour update of the integration kernel == ColorVoxel::Integrate arithmetic for every
    (weight < 8, old, new) -- 524288 words checked on the device against the integer-quotient form that
    tests/test_oracle_kat.py pins to the reference's float expression"""
    import ctypes as C
    bad = C.c_uint(12345)
    assert hip_lib.chisel_hip_kat_color_fresh(C.byref(bad)) == 0
    assert bad.value == 0


def test_device_reciprocal_exhaustive(hip_lib):
    """the short reciprocal used for the projection of chunks in front of the camera == IEEE 1.0f / z for every float of
    its admitted range [2^-40, 2^40] (6.8e8 values, all checked on the device)"""
    import ctypes as C
    bad, ex = C.c_ulonglong(1), C.c_uint(0)
    assert hip_lib.chisel_hip_kat_reciprocal(C.byref(bad), C.byref(ex)) == 0
    assert bad.value == 0, "first mismatch at bits 0x%08x" % ex.value


def test_device_floor_exhaustive(hip_lib):
    """the one-instruction floor-to-int of the projection ((int)u, (int)v of ProjectionIntegrator.h:72 / :131 after the image
    test) == v_floor_f32 + v_cvt_i32_f32 for every float that is not a NaN, and a NaN never comes out as a possible pixel
    coordinate (all 2^32 bit patterns checked on the device)"""
    import ctypes as C
    bad, ex = C.c_ulonglong(1), C.c_uint(0)
    assert hip_lib.chisel_hip_kat_floor(C.byref(bad), C.byref(ex)) == 0
    assert bad.value == 0, "first mismatch at bits 0x%08x" % ex.value


# ---- frame-level parity ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("scene", ["wall", "sphere_room", "box_room"])
def test_depth_only_stream(oracle_mod, scene):
    om, gm, integ = _mk(oracle_mod, 8, 0.05, False)
    cam = small_camera(64, 48)
    _run(om, gm, integ, make_frames(scene, 4, 64, 48), cam)
    assert gm.NumChunks() > 50


@pytest.mark.parametrize("channels", [1, 3, 4])
def test_color_stream(oracle_mod, channels):
    om, gm, integ = _mk(oracle_mod, 8, 0.05, True)
    cam = small_camera(64, 48)
    color = synth.render_color(64, 48, channels)
    _run(om, gm, integ, make_frames("sphere_room", 12, 64, 48), cam, color_img=color)  # > 8 frames: colour weight saturates


@pytest.mark.parametrize("N,res,W,H", [(16, 0.04, 96, 72), (32, 0.02, 64, 48), (8, 0.10, 160, 120)])
def test_chunk_sizes(oracle_mod, N, res, W, H):
    om, gm, integ = _mk(oracle_mod, N, res, True, max_chunks=2048)
    cam = small_camera(W, H)
    _run(om, gm, integ, make_frames("box_room", 3, W, H), cam, color_img=synth.render_color(W, H, 3))


@pytest.mark.parametrize("trunc", [("constant", 0.12), ("quadratic", 1.5), ("inverse", 0.7)])
@pytest.mark.parametrize("color", [False, True])
def test_truncators(oracle_mod, trunc, color):
    om, gm, integ = _mk(oracle_mod, 8, 0.05, color, trunc=trunc, weight=2.0)
    cam = small_camera(64, 48)
    _run(om, gm, integ, make_frames("sphere_room", 3, 64, 48), cam, color_img=synth.render_color(64, 48, 3) if color else None)


@pytest.mark.parametrize("color", [False, True])
def test_carving_moves_surface(oracle_mod, color):
    """Integrate a near wall, then a far one along the same rays: the old surface is carved / decayed."""
    om, gm, integ = _mk(oracle_mod, 8, 0.05, color, carving=True, carving_dist=0.0)
    cam = small_camera(64, 48)
    pose = synth.pose_yaw(0.0)
    near_wall = np.full((48, 64), 1.2, np.float32)
    far_wall = np.full((48, 64), 2.4, np.float32)
    frames = [(near_wall, pose)] * 7 + [(far_wall, pose)] * 4
    _run(om, gm, integ, frames, cam, color_img=synth.render_color(64, 48, 3) if color else None)
    # carving really happened in this scenario
    om2, gm2, integ2 = _mk(oracle_mod, 8, 0.05, color, carving=True, carving_dist=0.0)
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    carved = 0
    for d, p in frames:
        if color:
            om2.integrate_depth_color(d, p, intr, synth.render_color(64, 48, 3))
        else:
            om2.integrate_depth(d, p, intr)
        carved += om2.counters()["carved"]
    assert carved > 1000


def test_carving_disabled(oracle_mod):
    om, gm, integ = _mk(oracle_mod, 8, 0.05, False, carving=False)
    cam = small_camera(64, 48)
    pose = synth.pose_yaw(0.0)
    frames = [(np.full((48, 64), 1.2, np.float32), pose)] * 2 + [(np.full((48, 64), 2.4, np.float32), pose)] * 2
    _run(om, gm, integ, frames, cam)


@pytest.mark.parametrize("color", [False, True])
def test_invalid_pixels(oracle_mod, color):
    """NaN, zero, negative, > 50 m / > 100 m, +-inf depth: every special value the reference tolerates."""
    om, gm, integ = _mk(oracle_mod, 8, 0.05, color)
    cam = small_camera(64, 48)
    rng = np.random.default_rng(7)
    frames = []
    for depth, pose in make_frames("sphere_room", 3, 64, 48, nan_fraction=0.05):
        d = depth.copy()
        sel = rng.random(d.shape)
        d[sel < 0.03] = 0.0
        d[(sel >= 0.03) & (sel < 0.05)] = -0.7
        d[(sel >= 0.05) & (sel < 0.07)] = 75.0
        d[(sel >= 0.07) & (sel < 0.09)] = 150.0
        d[(sel >= 0.09) & (sel < 0.10)] = np.inf
        d[(sel >= 0.10) & (sel < 0.11)] = -np.inf
        frames.append((d, pose))
    _run(om, gm, integ, frames, cam, color_img=synth.render_color(64, 48, 3) if color else None)


def test_all_invalid_and_empty_map(oracle_mod):
    om, gm, integ = _mk(oracle_mod, 8, 0.05, False)
    cam = small_camera(64, 48)
    _run(om, gm, integ, [(np.full((48, 64), np.nan, np.float32), synth.pose_yaw(10.0))], cam)
    assert gm.NumChunks() == 0 and len(gm.GetChunkIDs()) == 0
    with pytest.raises(KeyError):
        gm.GetChunk((0, 0, 0))
    assert not gm.HasChunk((0, 0, 0))


def test_far_plane_limits_candidates(oracle_mod):
    """Depth beyond the far plane: the reference never enumerates those chunks; neither may the GPU."""
    om, gm, integ = _mk(oracle_mod, 8, 0.05, False)
    cam = small_camera(64, 48, far=1.5)
    _run(om, gm, integ, make_frames("sphere_room", 2, 64, 48), cam)


def test_moving_camera_noise_multi_agent(oracle_mod):
    om, gm, integ = _mk(oracle_mod, 8, 0.05, True, max_chunks=8192)
    cam = small_camera(64, 48)
    frames = make_frames("sphere_room", 3, 64, 48, agents=4, noise=True, nan_fraction=0.02)
    _run(om, gm, integ, frames, cam, color_img=synth.render_color(64, 48, 3), check_each=False)


def test_ragged_image_sizes(oracle_mod):
    """Width not a multiple of 4 / 64: pyramid edge tiles and the unaligned depth path."""
    om, gm, integ = _mk(oracle_mod, 8, 0.05, False)
    cam = small_camera(67, 45)
    _run(om, gm, integ, make_frames("box_room", 2, 67, 45), cam)


def test_reset_and_garbage_collect(oracle_mod):
    om, gm, integ = _mk(oracle_mod, 8, 0.05, False)
    cam = small_camera(64, 48)
    frames = make_frames("wall", 2, 64, 48)
    _run(om, gm, integ, frames, cam)
    ids = gm.GetChunkIDs()
    victims = ids[:: 3]
    gm.GarbageCollect(victims)
    for v in victims:
        om.remove_chunk(v)
    gm.GarbageCollect(victims[:2])  # removing twice is a no-op (ChunkManager.h:99-108 returns false)
    assert gm.NumChunks() == om.num_chunks()
    compare_fields(om.fields(), gm.fields(), om.V, False)
    _run(om, gm, integ, make_frames("wall", 2, 64, 48, start=2), cam)  # freed slots and tombstones are reused
    gm.Reset()
    om.reset()
    assert gm.NumChunks() == 0
    _run(om, gm, integ, frames, cam)


def test_meshes_to_update_kept_incrementally(oracle_mod):
    """chisel_hip_meshes_to_update_since (what the C++ facade's GetMeshesToUpdate keeps its set with, read after every frame:
    ChiselServer.cpp:346) against the full listing and the oracle's meshesToUpdate, across frames, a garbage collection of dirty
    chunks, a recompute (Chisel.cpp:57 clears the set), a tiny staging capacity (the grow-and-retry path) and a reset."""
    import ctypes as C
    om, gm, integ = _mk(oracle_mod, 8, 0.05, True, carving=True, carving_dist=0.02)
    cam = small_camera(64, 48)
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    color = synth.render_color(64, 48, 3)
    frames = make_frames("sphere_room", 14, 64, 48)
    cursor = (C.c_uint64 * 2)(0, 0)
    mine = set()

    def step(capacity=8192):
        ids, cleared = gm.GetMeshesToUpdateSince(cursor, capacity)
        if cleared:
            mine.clear()
        mine.update(map(tuple, ids.tolist()))
        full = set(map(tuple, gm.GetMeshesToUpdate().tolist()))
        assert mine == full, (len(mine), len(full))
        assert full == set(map(tuple, om.meshes_to_update().tolist())), (len(full), len(om.meshes_to_update()))

    for k, (d, p) in enumerate(frames):
        om.integrate_depth_color(d, p, intr, color, near=cam.near_plane, far=cam.far_plane)
        gm.IntegrateDepthScanColor(integ, d, p, cam, color, p, cam)
        if k % 2:  # the facade's form: the listing queued behind the integration, one wait for both
            gm.PrefetchMeshesToUpdate(cursor)
            gm.synchronize()
        step(8 if k == 2 else 8192)
        if k == 4:  # dirty chunks disappear: their neighbourhoods stay in the set (Chisel.h:228 lives on the host)
            victims = gm.GetChunkIDs()[::3]
            gm.GarbageCollect(victims)
            for v in victims:
                om.remove_chunk(v)
            step()
        if k == 6:
            # a listing prefetched behind the integration, then something ELSE dirties chunks before the caller reads the set (the facade's
            # order IntegrateDepthScan, IntegratePointCloud, GetMeshesToUpdate): the prefetched listing is stale and must not be served
            gm.PrefetchMeshesToUpdate(cursor)
            gm.synchronize()
            pose = synth.trajectory_pose(40)
            pts = synth.depth_to_cloud(synth.render_depth("sphere_room", pose, synth.intrinsics(32, 24), 32, 24), synth.intrinsics(32, 24), 0.6)
            om.integrate_pointcloud(pts, pose, None, 0.1, 5.0)
            gm.IntegratePointCloud(integ, (pts, None), pose, 0.1, 5.0)
            step()
        if k == 8:
            gm.UpdateMeshes(force=True)
            om.update_meshes()
            step()
            assert not mine
        if k == 11:
            gm.Reset()
            om.reset()
            step()
            assert not mine
    assert len(mine) > 20


def test_reset_and_garbage_collect_between_batches_in_flight(oracle_mod, monkeypatch):
    """Reset and GarbageCollect issued while batches are queued on all three streams (no synchronisation by the caller): the front
    halves of the batches that follow must see the map as those calls left it (pending sets of batches before the reset are
    stale, removed chunks must not be found in the hash, freed slots are handed out again)"""
    monkeypatch.setenv("CHISEL_HIP_FORCE_PIPELINE", "1")
    om, gm, integ = _mk(oracle_mod, 8, 0.05, True, carving=True, carving_dist=0.02, max_chunks=8192)
    monkeypatch.delenv("CHISEL_HIP_FORCE_PIPELINE", raising=False)
    cam = small_camera(64, 48)
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    color = synth.render_color(64, 48, 3)
    frames = make_frames("sphere_room", 24, 64, 48, agents=2, nan_fraction=0.02)

    def both(part):
        for d, p in part:
            om.integrate_depth_color(d, p, intr, color, near=cam.near_plane, far=cam.far_plane)
        for lo in range(0, len(part), 2):
            chunk = part[lo:lo + 2]
            gm.IntegrateBatch(integ, [(d, p, cam) for d, p in chunk], [(color, p, cam) for _, p in chunk])  # asynchronous

    both(frames[:8])
    gm.Reset()
    om.reset()
    both(frames[4:14])
    ids = gm.GetChunkIDs()  # (waits for the batches queued so far)
    victims = ids[::4]
    gm.GarbageCollect(victims)
    for v in victims:
        om.remove_chunk(v)
    both(frames[10:24])  # re-creates some of the removed chunks while the hash still holds their tombstones
    assert om.num_chunks() == gm.NumChunks()
    compare_fields(om.fields(), gm.fields(), om.V, True)


def test_pool_exhaustion_is_reported(oracle_mod):
    from cvids_amd import capi
    om, gm, integ = _mk(oracle_mod, 8, 0.05, False, max_chunks=16)
    cam = small_camera(64, 48)
    depth, pose = make_frames("sphere_room", 1, 64, 48)[0]
    gm.IntegrateDepthScan(integ, depth, pose, cam)
    with pytest.raises(capi.ChiselHipError) as e:
        gm.synchronize()
    assert e.value.code == 3


def test_growing_pool_integrates_past_its_first_size(oracle_mod):
    """max_chunks < 0: a pool that starts small and grows like the reference's map of heap chunks (ChunkManager.h:40-55): a stream that needs
    many times the first commitment integrates bit for bit like the oracle, without CHISEL_HIP_ERR_POOL_FULL, in launch sets of several
    frames queued back to back (the growth is decided from lagging reports) and with a reset in between (the grown pool stays)."""
    if os.environ.get("CHISEL_HIP_GROW") == "0":
        pytest.skip("CHISEL_HIP_GROW=0: every pool keeps its first size (tools/stress_hooks.sh runs the suite under it)")
    om, gm, integ = _mk(oracle_mod, 16, 0.02, True, max_chunks=-16)  # (commits whole 2 MiB pages: 128 chunks of 16^3 to begin with)
    info0 = gm.pool_info()
    assert info0["growable"] and info0["limit"] > info0["committed"] >= 16
    cam = small_camera(96, 72)
    color = synth.render_color(96, 72, 3)
    frames = make_frames("sphere_room", 12, 96, 72)
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    for rnd in range(2):
        for lo in range(0, 12, 3):
            part = frames[lo:lo + 3]
            for d, p in part:
                om.integrate_depth_color(d, p, intr, color, near=cam.near_plane, far=cam.far_plane)
            gm.IntegrateBatch(integ, [(d, p, cam) for d, p in part], [(color, p, cam) for _, p in part])
        gm.synchronize()
        assert om.num_chunks() == gm.NumChunks()
        compare_fields(om.fields(), gm.fields(), om.V, True)
        info = gm.pool_info()
        assert info["committed"] >= gm.NumChunks() and (info["grown"] >= 1 or info0["committed"] >= om.num_chunks())
        if rnd == 0:
            gm.Reset()
            om = _mk(oracle_mod, 16, 0.02, True)[0]
    assert gm.pool_info()["grown"] >= 1 and om.num_chunks() > info0["committed"]  # (the scene needs more than the first commitment: the test is one)
    # a saved map loaded into a pool that has to grow for it
    _, g2, _ = _mk(oracle_mod, 16, 0.02, True, max_chunks=-16)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "m.bin")
        gm.SaveMap(path)
        g2.LoadMap(path)
    assert g2.NumChunks() == gm.NumChunks() and g2.pool_info()["committed"] >= gm.NumChunks()
    compare_fields(gm.fields(), g2.fields(), om.V, True)


def test_device_resident_frames_and_batch(oracle_mod):
    """Frames already in HBM (torch tensors) through integrate_batch == frame-by-frame host frames."""
    import torch
    om, gm, integ = _mk(oracle_mod, 8, 0.05, True)
    cam = small_camera(64, 48)
    frames = make_frames("sphere_room", 5, 64, 48)
    color = synth.render_color(64, 48, 3)
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    for d, p in frames:
        om.integrate_depth_color(d, p, intr, color)
    dev = torch.device("cuda:0")
    d_dev = [torch.from_numpy(d).to(dev) for d, _ in frames]
    c_dev = torch.from_numpy(color).to(dev)
    torch.cuda.synchronize()
    gm.IntegrateBatch(integ, [(d_dev[i], frames[i][1], cam) for i in range(5)], [(c_dev, frames[i][1], cam) for i in range(5)])
    compare_fields(om.fields(), gm.fields(), om.V, True)


def test_page_locked_host_frames(oracle_mod):
    """Depth frames in page-locked host memory are read by the pyramid kernel straight over the bus (no staging copy): same result
    as pageable host frames, batches queued back to back."""
    import torch
    om, gm, integ = _mk(oracle_mod, 16, 0.04, True)
    W, H = 160, 120
    cam = small_camera(W, H)
    frames = make_frames("sphere_room", 12, W, H, nan_fraction=0.01)
    color = synth.render_color(W, H, 3)
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    for d, p in frames:
        om.integrate_depth_color(d, p, intr, color, near=cam.near_plane, far=cam.far_plane)
    pins = [torch.from_numpy(d).pin_memory() for d, _ in frames]
    assert all(t.is_pinned() for t in pins)
    views = [t.numpy() for t in pins]   # host pointers into page-locked memory
    for lo in range(0, 12, 4):
        gm.IntegrateBatch(integ, [(views[i], frames[i][1], cam) for i in range(lo, lo + 4)], [(color, frames[i][1], cam) for i in range(lo, lo + 4)])
    gm.synchronize()
    compare_fields(om.fields(), gm.fields(), om.V, True)


def _run_batched(om, gm, integ, frames, cam, color_img, batch):
    """oracle frame by frame, HIP path `batch` frames per chisel_hip_integrate_batch call; compare after every call"""
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    for lo in range(0, len(frames), batch):
        part = frames[lo:lo + batch]
        tot = dict(sdf=0, col=0, col_sat=0, probe=0, carved=0, updated_chunks=0)
        for depth, pose in part:
            if color_img is None:
                om.integrate_depth(depth, pose, intr, cam.near_plane, cam.far_plane)
            else:
                om.integrate_depth_color(depth, pose, intr, color_img, near=cam.near_plane, far=cam.far_plane)
            oc = om.counters()
            for k in tot:
                tot[k] += oc[k]
        gm.IntegrateBatch(integ, [(d, p, cam) for d, p in part], None if color_img is None else [(color_img, p, cam) for _, p in part])
        gc = gm.counters(reset=True)
        for k in tot:
            assert tot[k] == gc[k], "frames %d..: counter %s: oracle %d gpu %d" % (lo, k, tot[k], gc[k])
        assert om.num_chunks() == gm.NumChunks()
        compare_fields(om.fields(), gm.fields(), om.V, om.use_color, what="frames %d.." % lo)
        assert sorted(map(tuple, om.meshes_to_update().tolist())) == sorted(map(tuple, gm.GetMeshesToUpdate().tolist()))


@pytest.mark.parametrize("batch", [2, 3, 8, 10, 11, 16, 17])
@pytest.mark.parametrize("color", [False, True])
def test_batched_launch_equals_frame_by_frame(oracle_mod, batch, color):
    """K frames in one launch set (voxel state kept in registers across frames) == the reference's frame-by-frame result,
    including chunks created by one frame of the batch and carved / probed by a later one."""
    om, gm, integ = _mk(oracle_mod, 8, 0.05, color, carving=True, carving_dist=0.0, max_chunks=8192)
    cam = small_camera(64, 48)
    pose = synth.pose_yaw(0.0)
    near_wall = np.full((48, 64), 1.2, np.float32)
    far_wall = np.full((48, 64), 2.4, np.float32)
    frames = [(near_wall, pose)] * 6 + [(far_wall, pose)] * 5 + make_frames("sphere_room", 6, 64, 48, agents=2, nan_fraction=0.02)
    _run_batched(om, gm, integ, frames, cam, synth.render_color(64, 48, 3) if color else None, batch)


@pytest.mark.parametrize("batch", [1, 2, 8])
@pytest.mark.parametrize("force_lookup", [None, False, True])
def test_back_to_back_batches_pipeline(oracle_mod, batch, force_lookup, monkeypatch):
    """Batches issued without any synchronisation in between: the work-list of batch b+1 is built on the auxiliary stream
    while batch b is still being integrated, so chunks batch b creates reach batch b+1 as SLOT_LOOKUP items (looked up by
    the integration kernel).  Walls that appear, move away (carving what the previous batch created) and come back.
    force_lookup None: the library picks per batch (two streams while a batch is in flight, the short single-stream form
    when the map is idle); False: always two streams; True: also the conservative mode used when a pending set overflows
    -- every candidate without a slot is looked up by the integration kernel."""
    if force_lookup is not None:
        monkeypatch.setenv("CHISEL_HIP_FORCE_UNCERTAIN" if force_lookup else "CHISEL_HIP_FORCE_PIPELINE", "1")
    om, gm, integ = _mk(oracle_mod, 8, 0.05, True, carving=True, carving_dist=0.0, max_chunks=8192)
    monkeypatch.delenv("CHISEL_HIP_FORCE_UNCERTAIN", raising=False)
    monkeypatch.delenv("CHISEL_HIP_FORCE_PIPELINE", raising=False)
    cam = small_camera(64, 48)
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    color = synth.render_color(64, 48, 3)
    pose = synth.pose_yaw(0.0)
    walls = [np.full((48, 64), d, np.float32) for d in (1.2, 2.4, 1.7, 1.2)]
    frames = []
    for w in walls:
        frames += [(w, pose)] * 3
    frames += make_frames("sphere_room", 10, 64, 48, agents=2, nan_fraction=0.02)
    tot = dict(sdf=0, col=0, col_sat=0, probe=0, carved=0, updated_chunks=0)
    for d, p in frames:
        om.integrate_depth_color(d, p, intr, color, near=cam.near_plane, far=cam.far_plane)
        oc = om.counters()
        for k in tot:
            tot[k] += oc[k]
    for lo in range(0, len(frames), batch):
        part = frames[lo:lo + batch]
        gm.IntegrateBatch(integ, [(d, p, cam) for d, p in part], [(color, p, cam) for _, p in part])  # asynchronous
    gc = gm.counters(reset=True)
    for k in tot:
        assert tot[k] == gc[k], "counter %s: oracle %d gpu %d" % (k, tot[k], gc[k])
    assert om.num_chunks() == gm.NumChunks()
    compare_fields(om.fields(), gm.fields(), om.V, True)


def test_launch_statistics_account_for_every_launch_set(oracle_mod, monkeypatch):
    """chisel_hip_get_launch_stats: every launch set is counted once under a granularity and once under a cull shape; a caller that
    waits after every call gets the single-stream form every time; the forced granularities show up where they were forced; a reset
    starts the count again."""
    cam = small_camera(96, 72)
    color = synth.render_color(96, 72, 3)
    frames = make_frames("sphere_room", 24, 96, 72)

    def run(wait):
        _, gm, integ = _mk(oracle_mod, 16, 0.04, True, max_chunks=4096)
        for lo in range(0, 24, 4):
            part = frames[lo:lo + 4]
            gm.IntegrateBatch(integ, [(d, p, cam) for d, p in part], [(color, p, cam) for _, p in part])
            if wait:
                gm.synchronize()
        gm.synchronize()
        return gm, gm.launch_stats()

    forced = bool(os.environ.get("CHISEL_HIP_FORCE_PIPELINE"))  # (tools/stress_hooks.sh runs the suite under it: no single-stream form then)
    gm, st = run(wait=True)
    assert st["launch_sets"] == 6 and st["single_stream_sets"] == (0 if forced else 6)
    assert st["integrate_2_per_lane"] + st["integrate_4_per_lane"] + st["integrate_4_with_2_tail"] == 6
    assert st["cull_4_waves"] + st["cull_wave_per_frame"] == 6
    assert gm.launch_stats(reset=True)["launch_sets"] == 6 and gm.launch_stats()["launch_sets"] == 0
    _, st = run(wait=False)
    assert st["launch_sets"] == 6 and (st["single_stream_sets"] == 0 if forced else 1 <= st["single_stream_sets"] <= 6)  # (the first set finds the map idle; the rest depends on timing)
    monkeypatch.setenv("CHISEL_HIP_VPL", "2")
    _, st = run(wait=True)
    assert st["integrate_2_per_lane"] == 6
    monkeypatch.setenv("CHISEL_HIP_VPL", "4")
    _, st = run(wait=True)
    assert st["integrate_2_per_lane"] == 0


@pytest.mark.parametrize("chunk", [8, 16, 32])
@pytest.mark.parametrize("mode", ["vpl2", "vpl4", "persistent", "growing", "cull1", "cull4", "cull16"])
def test_integration_schedules(oracle_mod, chunk, mode, monkeypatch):
    """The integration kernel's two granularities (2 / 4 voxels per lane, normally picked per launch from the item count a
    recent launch reported) and its two ways of handing out units (one unit per wave with a grid sized from that count;
    persistent waves pulling from the queue heads) give the same map: each forced over a stream with batches of 1..8 frames,
    colour, carving.  "growing": the reported count is far too small for the next launch (a small wall, then the whole room),
    so most units come from the queue heads of a small grid.  "cull1" / "cull4" / "cull16": the cull kernel's three workgroup shapes (one wave that takes
    every frame, or four waves of several frames each, picked by the host for launches whose frames look at different parts of the space; one
    wave per frame), each forced over the two-agent stream."""
    if mode in ("cull1", "cull4", "cull16"):
        monkeypatch.setenv("CHISEL_HIP_CULL_WAVES", mode[4:])
    if mode == "vpl2":
        monkeypatch.setenv("CHISEL_HIP_VPL", "2")
    elif mode == "vpl4":
        monkeypatch.setenv("CHISEL_HIP_VPL", "4")
    elif mode == "persistent":
        monkeypatch.setenv("CHISEL_HIP_PERSISTENT", "1")
    res = {8: 0.05, 16: 0.03, 32: 0.02}[chunk]
    om, gm, integ = _mk(oracle_mod, chunk, res, True, carving=True, carving_dist=0.02, max_chunks=8192)
    cam = small_camera(96, 72)
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    color = synth.render_color(96, 72, 3)
    frames = make_frames("sphere_room", 14, 96, 72, agents=2, nan_fraction=0.02)
    if mode == "growing":
        tiny = np.full((72, 96), np.nan, np.float32)
        tiny[30:40, 40:56] = 1.0
        frames = [(tiny, synth.pose_yaw(0.0))] * 3 + frames
    for d, p in frames:
        om.integrate_depth_color(d, p, intr, color, near=cam.near_plane, far=cam.far_plane)
    sizes = [1, 1, 1, 4, 8, 2, 5] if mode == "growing" else [1, 4, 8, 2, 5]
    lo = 0
    for n in sizes * 4:
        part = frames[lo:lo + n]
        if not part:
            break
        gm.IntegrateBatch(integ, [(d, p, cam) for d, p in part], [(color, p, cam) for _, p in part])
        lo += n
    assert lo >= len(frames)
    assert om.num_chunks() == gm.NumChunks()
    compare_fields(om.fields(), gm.fields(), om.V, True)


@pytest.mark.parametrize("agents", [3, 4])
def test_chunks_seen_by_every_third_launch(oracle_mod, agents, monkeypatch):
    """Agents that look in different directions, ONE frame per call, nothing waited for: a chunk only agent 0 sees is a candidate of
    launch sets k, k + agents, ... and of none in between -- the front half of set k + 3 may look it up while the integration of set k
    is still creating it.  The map must equal the oracle's all the same (no chunk created twice, none missed)."""
    monkeypatch.setenv("CHISEL_HIP_FORCE_PIPELINE", "1")
    om, gm, integ = _mk(oracle_mod, 16, 0.02, False, max_chunks=16384)
    W, H = 320, 240
    cam = small_camera(W, H)
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    frames = make_frames("sphere_room", 8, W, H, agents=agents)
    for d, p in frames:
        om.integrate_depth(d, p, intr, cam.near_plane, cam.far_plane)
    for d, p in frames:
        gm.IntegrateBatch(integ, [(d, p, cam)])
    assert om.num_chunks() == gm.NumChunks()
    compare_fields(om.fields(), gm.fields(), om.V, False)


def test_chunk_listing_is_in_ascending_order(oracle_mod):
    """chisel_hip_list_chunks: ascending (x, y, z) whatever order the device compacted the slots in -- a caller that acts on every
    n-th listed chunk (tools/soak.py's garbage collection) then removes the same chunks every time"""
    _, gm, integ = _mk(oracle_mod, 16, 0.04, False, max_chunks=4096)
    cam = small_camera(96, 72)
    gm.IntegrateBatch(integ, [(d, p, cam) for d, p in make_frames("sphere_room", 6, 96, 72, agents=2)])
    ids = np.asarray(gm.GetChunkIDs()).reshape(-1, 3)
    assert len(ids) > 50
    assert [tuple(r) for r in ids.tolist()] == sorted(tuple(r) for r in ids.tolist())


def test_checkpoint_and_resume(oracle_mod, tmp_path):
    """chisel_hip_save_map / load_map: a map dumped in the middle of a stream and restored into a fresh map continues
    bit for bit like the uninterrupted run (and like the oracle); the dump itself reads back identically"""
    om, gm, integ = _mk(oracle_mod, 8, 0.05, True, max_chunks=8192)
    cam = small_camera(64, 48)
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    color = synth.render_color(64, 48, 3)
    frames = make_frames("sphere_room", 10, 64, 48, nan_fraction=0.02)
    for d, p in frames[:6]:
        om.integrate_depth_color(d, p, intr, color, near=cam.near_plane, far=cam.far_plane)
    gm.IntegrateBatch(integ, [(d, p, cam) for d, p in frames[:6]], [(color, p, cam) for _, p in frames[:6]])
    path = str(tmp_path / "map.chsl")
    gm.SaveMap(path)
    compare_fields(om.fields(), gm.fields(), om.V, True)
    from cvids_amd import chisel as ch
    g2 = ch.Chisel((8, 8, 8), 0.05, True, max_chunks=8192)
    g2.LoadMap(path)
    compare_fields(om.fields(), g2.fields(), om.V, True, what="restored")
    for d, p in frames[6:]:
        om.integrate_depth_color(d, p, intr, color, near=cam.near_plane, far=cam.far_plane)
    g2.IntegrateBatch(integ, [(d, p, cam) for d, p in frames[6:]], [(color, p, cam) for _, p in frames[6:]])
    assert om.num_chunks() == g2.NumChunks()
    compare_fields(om.fields(), g2.fields(), om.V, True, what="resumed")
    from cvids_amd import capi
    with pytest.raises(capi.ChiselHipError):
        ch.Chisel((16, 16, 16), 0.05, True).LoadMap(path)  # another chunk size


def test_caller_owned_stream(oracle_mod):
    """chisel_hip_set_stream: the map's kernels run on the caller's stream, behind the kernels that produce the frames there
    (no event, no host wait between producer and integration); switching back to the map's own stream afterwards"""
    import torch
    om, gm, integ = _mk(oracle_mod, 8, 0.05, True, max_chunks=8192)
    cam = small_camera(64, 48)
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    color = synth.render_color(64, 48, 3)
    frames = make_frames("sphere_room", 6, 64, 48, agents=2, nan_fraction=0.02)  # 12 frames
    for d, p in frames:
        om.integrate_depth_color(d, p, intr, color, near=cam.near_plane, far=cam.far_plane)
    dev = torch.device("cuda:0")
    src = [torch.from_numpy(np.stack([frames[lo + j][0] for j in range(4)])).to(dev) for lo in range(0, 12, 4)]
    c_dev = torch.from_numpy(color).to(dev)
    buf = torch.zeros((4, 48, 64), dtype=torch.float32, device=dev)
    user = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    gm.set_stream(user.cuda_stream)
    for b in range(2):
        with torch.cuda.stream(user):
            torch.cuda._sleep(10_000_000)          # the producer is slow; stream order alone must protect the buffer
            buf.copy_(src[b], non_blocking=True)
        part = frames[4 * b:4 * b + 4]
        gm.IntegrateBatch(integ, [(buf[j], p, cam) for j, (_, p) in enumerate(part)], [(c_dev, p, cam) for _, p in part])
    gm.set_stream(0)                               # back to the map's own stream (waits for the queued work)
    part = frames[8:12]
    gm.IntegrateBatch(integ, [(src[2][j], p, cam) for j, (_, p) in enumerate(part)], [(c_dev, p, cam) for _, p in part])
    assert om.num_chunks() == gm.NumChunks()
    compare_fields(om.fields(), gm.fields(), om.V, True)


def test_event_ordered_device_frames(oracle_mod):
    """chisel_hip_wait_event / chisel_hip_record_event: device frames produced late on another stream (as an RCCL
    all-gather would) and one frame buffer reused for every batch, ordered with events only -- no host wait anywhere."""
    import torch
    om, gm, integ = _mk(oracle_mod, 8, 0.05, True, max_chunks=8192)
    cam = small_camera(64, 48)
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    color = synth.render_color(64, 48, 3)
    frames = make_frames("sphere_room", 6, 64, 48, agents=2, nan_fraction=0.02)  # 12 frames
    assert len(frames) == 12
    for d, p in frames:
        om.integrate_depth_color(d, p, intr, color, near=cam.near_plane, far=cam.far_plane)
    dev = torch.device("cuda:0")
    src = [torch.from_numpy(np.stack([frames[lo + j][0] for j in range(4)])).to(dev) for lo in range(0, 12, 4)]
    c_dev = torch.from_numpy(color).to(dev)
    buf = torch.zeros((4, 48, 64), dtype=torch.float32, device=dev)  # ONE buffer for all batches
    producer = torch.cuda.Stream(device=dev)
    ready, free = torch.cuda.Event(), torch.cuda.Event()
    ready.record(producer)
    free.record(producer)
    torch.cuda.synchronize()
    for b in range(3):
        with torch.cuda.stream(producer):
            producer.wait_event(free)               # the previous batch has read the buffer
            torch.cuda._sleep(20_000_000)           # the data arrives late (about 10 ms)
            buf.copy_(src[b], non_blocking=True)
            ready.record(producer)
        gm.wait_event(ready.cuda_event)
        part = frames[4 * b:4 * b + 4]
        gm.IntegrateBatch(integ, [(buf[j], p, cam) for j, (_, p) in enumerate(part)], [(c_dev, p, cam) for _, p in part])
        gm.record_event(free.cuda_event)
    assert om.num_chunks() == gm.NumChunks()
    compare_fields(om.fields(), gm.fields(), om.V, True)


def test_event_ordered_device_frames_of_a_call_cut_into_several_launch_sets(oracle_mod, monkeypatch):
    """One chisel_hip_integrate_batch call of 40 device frames = three launch sets whose front halves alternate between the two
    auxiliary streams: the event of chisel_hip_wait_event must hold back every one of them, not only the first (the frames arrive
    about 10 ms after the call has been issued)."""
    import torch
    monkeypatch.setenv("CHISEL_HIP_FORCE_PIPELINE", "1")
    om, gm, integ = _mk(oracle_mod, 8, 0.05, True, max_chunks=8192)
    cam = small_camera(64, 48)
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    color = synth.render_color(64, 48, 3)
    frames = make_frames("sphere_room", 20, 64, 48, agents=2, nan_fraction=0.02)
    assert len(frames) == 40
    for d, p in frames:
        om.integrate_depth_color(d, p, intr, color, near=cam.near_plane, far=cam.far_plane)
    dev = torch.device("cuda:0")
    src = torch.from_numpy(np.stack([d for d, _ in frames])).to(dev)
    c_dev = torch.from_numpy(color).to(dev)
    buf = torch.full((40, 48, 64), float("nan"), dtype=torch.float32, device=dev)  # until the producer has run: no valid pixel
    producer = torch.cuda.Stream(device=dev)
    ready = torch.cuda.Event()
    torch.cuda.synchronize()
    with torch.cuda.stream(producer):
        torch.cuda._sleep(20_000_000)
        buf.copy_(src, non_blocking=True)
        ready.record(producer)
    gm.wait_event(ready.cuda_event)
    gm.IntegrateBatch(integ, [(buf[j], p, cam) for j, (_, p) in enumerate(frames)], [(c_dev, p, cam) for _, p in frames])
    assert om.num_chunks() == gm.NumChunks()
    compare_fields(om.fields(), gm.fields(), om.V, True)


@pytest.mark.parametrize("N,res,W,H", [(16, 0.04, 96, 72), (32, 0.02, 64, 48)])
def test_batched_launch_chunk_sizes(oracle_mod, N, res, W, H):
    om, gm, integ = _mk(oracle_mod, N, res, True, max_chunks=2048)
    cam = small_camera(W, H)
    _run_batched(om, gm, integ, make_frames("box_room", 8, W, H), cam, synth.render_color(W, H, 3), 4)


def test_batch_mixed_image_sizes(oracle_mod):
    """a batch whose frames differ in size is split into launch sets of equal size, order preserved"""
    om, gm, integ = _mk(oracle_mod, 8, 0.05, False)
    intr_a, intr_b = synth.intrinsics(64, 48), synth.intrinsics(80, 60)
    from cvids_amd.chisel import PinholeCamera
    cam_a, cam_b = PinholeCamera(*intr_a, 64, 48), PinholeCamera(*intr_b, 80, 60)
    fa = make_frames("sphere_room", 2, 64, 48)
    fb = make_frames("sphere_room", 2, 80, 60, start=2)
    seq = [(fa[0], cam_a, intr_a), (fb[0], cam_b, intr_b), (fb[1], cam_b, intr_b), (fa[1], cam_a, intr_a)]
    for (d, p), cam, intr in seq:
        om.integrate_depth(d, p, intr, cam.near_plane, cam.far_plane)
    gm.IntegrateBatch(integ, [(d, p, cam) for (d, p), cam, _ in seq])
    compare_fields(om.fields(), gm.fields(), om.V, False)


def test_image_size_changes_between_pipelined_batches(oracle_mod):
    """Twelve batches issued back to back whose image size changes three times (growing, then shrinking): the pixel-record and pyramid
    buffers of ALL buffer sets are reallocated while earlier batches are still in flight (the library waits for them first), the host
    staging rings grow, and the rings of buffer sets / pending sets go round more than once; host frames with colour."""
    from cvids_amd.chisel import PinholeCamera
    om, gm, integ = _mk(oracle_mod, 8, 0.05, True, max_chunks=8192)
    sizes = [(64, 48)] * 3 + [(96, 72)] * 3 + [(128, 96)] * 3 + [(80, 60)] * 3
    k = 0
    for W, H in sizes:
        intr = synth.intrinsics(W, H)
        cam = PinholeCamera(*intr, W, H)
        color = synth.render_color(W, H, 3)
        part = make_frames("sphere_room", 3, W, H, start=k, nan_fraction=0.01)
        k += 3
        for d, p in part:
            om.integrate_depth_color(d, p, intr, color, near=cam.near_plane, far=cam.far_plane)
        gm.IntegrateBatch(integ, [(d, p, cam) for d, p in part], [(color, p, cam) for _, p in part])  # asynchronous: nothing waited for
    assert om.num_chunks() == gm.NumChunks()
    compare_fields(om.fields(), gm.fields(), om.V, True)
    assert gm.counters()["frames"] == 36


@pytest.mark.parametrize("n_shards", [2, 4, 8])
def test_spatial_shards_reproduce_the_unsharded_map(oracle_mod, n_shards):
    """every voxel has one owner -> the union of the shards' chunks is bit-identical to one map (and to the
    oracle); the shards are disjoint and each holds only chunks it owns.  All shards live on this one GPU here."""
    from cvids_amd import chisel as ch
    om, gm, integ = _mk(oracle_mod, 8, 0.05, True, max_chunks=8192)
    shards = [ch.Chisel((8, 8, 8), 0.05, True, max_chunks=8192, n_shards=n_shards, shard_rank=r) for r in range(n_shards)]
    cam = small_camera(64, 48)
    color = synth.render_color(64, 48, 3)
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    frames = make_frames("sphere_room", 6, 64, 48, agents=2, nan_fraction=0.02)
    for lo in range(0, len(frames), 4):
        part = frames[lo:lo + 4]
        for d, p in part:
            om.integrate_depth_color(d, p, intr, color)
        for s_ in shards:
            s_.IntegrateBatch(integ, [(d, p, cam) for d, p in part], [(color, p, cam) for _, p in part])
    union = {}
    tot = dict(sdf=0, col=0, probe=0)
    for r, s_ in enumerate(shards):
        f = s_.fields()
        assert not (set(f) & set(union)), "shards overlap"
        for cid in f:
            assert ch.chunk_owner(cid, n_shards, 2) == r
        union.update(f)
        c = s_.counters()
        for k in tot:
            tot[k] += c[k]
    compare_fields(om.fields(), union, om.V, True, what="%d shards" % n_shards)
    assert len(union) == om.num_chunks()
    assert min(len(s_.fields()) for s_ in shards) > 0


def test_upload_download_roundtrip(oracle_mod):
    om, gm, integ = _mk(oracle_mod, 8, 0.05, True)
    rng = np.random.default_rng(3)
    sdf = rng.normal(size=512).astype(np.float32)
    w = rng.random(512).astype(np.float32)
    rgbw = rng.integers(0, 256, (512, 4)).astype(np.uint8)
    gm.AddChunk((3, -2, 7), sdf, w, rgbw)
    s2, w2, c2 = gm.GetChunk((3, -2, 7))
    assert np.array_equal(sdf, s2) and np.array_equal(w, w2) and np.array_equal(rgbw, c2)
    assert gm.HasChunk((3, -2, 7)) and gm.NumChunks() == 1


@pytest.mark.parametrize("color", [False, True])
def test_full_size_frame_2cm(oracle_mod, color):
    """BASELINE configs[0]/[1]: one 640x480 frame at 2 cm, chunk 16 (the oracle needs ~1 GB and a few seconds)."""
    om, gm, integ = _mk(oracle_mod, 16, 0.02, color, max_chunks=16384)
    cam = small_camera(640, 480)
    frames = make_frames("sphere_room", 2, 640, 480, nan_fraction=0.02)
    _run(om, gm, integ, frames, cam, color_img=synth.render_color(640, 480, 3) if color else None, check_each=False, atol=0.0)


def test_full_size_stream_is_reproducible_across_modes(oracle_mod, monkeypatch):
    """BASELINE-sized frames (640x480 @ 1 cm, near-camera chunks whose pixel boxes do not fit the LDS tile, ~1200 work items
    per launch, several items per workgroup): the two-stream pipeline, the single-stream form and the conservative
    look-everything-up mode must produce the same map, bit for bit, and the same counters, run after run; the first
    launch set is also checked against the oracle.  (Two races that only showed at this size were found this way: a
    free-running wave overwriting the tile buffer a slower wave still read, and a half-inserted hash entry read by the
    work-list builder in the conservative mode.)"""
    import torch
    from cvids_amd import chisel as ch
    W, H, N, res = 640, 480, 16, 0.01
    intr = synth.intrinsics(W, H)
    cam = ch.PinholeCamera(*intr, W, H, 0.05, 5.0)
    integ = ch.ProjectionIntegrator(ch.InverseTruncator(1.0), ch.ConstantWeighter(1.0), 0.05, True)
    frames = list(synth.stream("sphere_room", 27, W, H))
    color = synth.render_color(W, H, 3)
    dev = torch.device("cuda:0")
    d_dev = [torch.from_numpy(d).to(dev) for d, _ in frames]
    c_dev = torch.from_numpy(color).to(dev)

    def run(env, n_frames=27, batch=9):
        for k in ("CHISEL_HIP_FORCE_UNCERTAIN", "CHISEL_HIP_FORCE_PIPELINE", "CHISEL_HIP_SERIAL"):
            monkeypatch.delenv(k, raising=False)
        if env:
            monkeypatch.setenv(env, "1")
        m = ch.Chisel((N,) * 3, res, True)
        if env:
            monkeypatch.delenv(env, raising=False)
        for lo in range(0, n_frames, batch):
            idx = range(lo, min(lo + batch, n_frames))
            m.IntegrateBatch(integ, [(d_dev[i], frames[i][1], cam) for i in idx], [(c_dev, frames[i][1], cam) for i in idx])
        f, c = m.fields(), m.counters()
        m.close()
        return f, c

    ref_f, ref_c = run(None)
    for env in (None, "CHISEL_HIP_FORCE_PIPELINE", "CHISEL_HIP_FORCE_UNCERTAIN", "CHISEL_HIP_SERIAL"):
        f, c = run(env)
        for k in ("sdf", "col", "col_sat", "probe", "carved", "new_chunks", "updated_chunks"):
            assert c[k] == ref_c[k], (env, k, c[k], ref_c[k])
        assert set(f) == set(ref_f)
        for cid in f:
            for a, b in zip(f[cid], ref_f[cid]):
                assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), (env, cid)
    # one launch set against the oracle (about 1 s of CPU per frame)
    om = oracle_mod.OracleMap(N, res, True, threads=16)
    om.set_integrator(oracle_mod.TRUNC_INVERSE, 1.0, 1.0, True, 0.05)
    for d, p in frames[:5]:
        om.integrate_depth_color(d, p, intr, color, near=0.05, far=5.0)
    f5, _ = run(None, n_frames=5, batch=5)
    compare_fields(om.fields(), f5, om.V, True)


@pytest.mark.parametrize("shape", [(480, 640), (240, 320), (376, 1241), (720, 1280), (97, 131), (960, 1280), (1, 1)])
def test_depth_conditioning(shape):
    """chisel_hip_condition_depth == the restated PublishDenseInfo conditioning (oracle/publish_dense.py), bit for bit: same
    size (copy), up- and down-scaling, odd sizes, readings out of range, NaN and infinity in the input"""
    from cvids_amd.chisel import condition_depth
    from oracle import publish_dense as pd
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    src = rng.uniform(0.0, 25.0, shape)
    src[rng.random(shape) < 0.02] = np.nan
    src[rng.random(shape) < 0.01] = np.inf
    src[rng.random(shape) < 0.01] = -1.0
    src[0, 0] = 0.1  # float(0.1) is kept: the test is "< 0.1f"
    want = pd.condition_depth(src, 640, 480)
    K0 = (718.856, 718.856, 607.1928, 185.2157)
    got, K = condition_depth(src, 640, 480, K0)
    assert got.shape == (480, 640) and got.dtype == np.float32
    assert np.array_equal(np.isnan(want), np.isnan(got))
    assert np.array_equal(want[~np.isnan(want)].view(np.uint32), got[~np.isnan(got)].view(np.uint32))
    assert K == pd.rescale_intrinsics(*K0, shape[1], shape[0], 640, 480)
    assert np.isnan(got).mean() > 0.05  # the range mask does something on this input


@pytest.mark.parametrize("shape", [(480, 640), (240, 320), (376, 1241), (720, 1280), (97, 131), (960, 1280), (1, 1)])
@pytest.mark.parametrize("cn", [0, 1, 3, 4])
def test_color_conditioning(shape, cn):
    """chisel_hip_condition_color == the restated cv::resize of the 8-bit colour image (oracle/publish_dense.py), byte for byte:
    copy, up- and down-scaling, exact halving (INTER_AREA), odd sizes; MONO8 / BGR8 / BGRA8"""
    from cvids_amd.chisel import condition_color
    from oracle import publish_dense as pd
    rng = np.random.default_rng(shape[0] * 11 + shape[1] + cn)
    src = rng.integers(0, 256, shape if cn == 0 else shape + (cn,), dtype=np.uint8)
    want = pd.condition_color(src, 640, 480)
    got = condition_color(src, 640, 480)
    assert got.shape == want.shape and got.dtype == np.uint8
    assert np.array_equal(want, got)


@pytest.mark.parametrize("cn", [0, 3])
def test_publish_cloud(cn):
    """chisel_hip_publish_cloud == the restated SendPointCloud (oracle/publish_dense.py), word for word: range mask, pixel
    coordinates as floats, the grey byte taken at byte offset `column` of the row for 1- and 3-channel images"""
    from cvids_amd.chisel import publish_cloud
    from oracle import publish_dense as pd
    rng = np.random.default_rng(5 + cn)
    h, w = 480, 640
    depth = rng.uniform(0.0, 12.0, (h, w))
    depth[rng.random((h, w)) < 0.02] = np.nan
    depth[0, 0], depth[0, 1], depth[0, 2] = 0.1, 10.0, 0.5  # float(0.1) > 0.1f is false, 10.0 < 10.0f is false
    color = rng.integers(0, 256, (h, w) if cn == 0 else (h, w, cn), dtype=np.uint8)
    want = pd.publish_cloud(depth, color)
    got = publish_cloud(depth, color)
    assert np.array_equal(want, got)
    assert (got[0, 0] == 0x7fc00000).all() and (got[0, 1] == 0x7fc00000).all() and got[0, 2, 2] == np.float32(0.5).view(np.uint32)


def test_rccl_exchange_world_size_one():
    """the N > 1 bench's frame exchange over backend "nccl" (= RCCL) on its own stream with event ordering, run with one
    rank on this GPU in a child process (tools/nccl_world1_check.py): same map as direct integration"""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "nccl_world1_check.py")], cwd=root, env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "nccl world-1 check ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
