"""GPU parity of the point-cloud fusion mode (Chisel::IntegratePointCloud, Chisel.cpp:107-157) against the CPU oracle.

Every voxel must receive the updates of the rays that meet it in cloud order (running averages in fp32), so the fields are
compared bit for bit, like the depth-image path.
"""
import ctypes as C

import numpy as np
import pytest

from cvids_amd import synth
from tests.common import compare_fields

pytestmark = pytest.mark.gpu


def _mk(oracle_mod, N, res, color, trunc=("inverse", 2.0), weight=1.0, carving=True, carving_dist=0.05, max_chunks=4096):
    from cvids_amd import chisel as ch
    kinds = {"constant": (0, ch.ConstantTruncator), "inverse": (1, ch.InverseTruncator), "quadratic": (2, ch.QuadraticTruncator)}
    k, cls = kinds[trunc[0]]
    om = oracle_mod.OracleMap(N, res, color)
    om.set_integrator(k, trunc[1], weight, carving, carving_dist)
    gm = ch.Chisel((N, N, N), res, color, max_chunks=max_chunks)
    integ = ch.ProjectionIntegrator(cls(trunc[1]), ch.ConstantWeighter(weight), carving_dist, carving)
    return om, gm, integ


def _cloud(scene, k, W, H, scale, colors, t=(0.0, 0.0, 0.0), nan_fraction=0.0):
    intr = synth.intrinsics(W, H)
    pose = synth.pose_yaw(3.0 * k, (0.02 * k + t[0], t[1], t[2]))
    depth = synth.render_depth(scene, synth.pose_yaw(3.0 * k, (0.02 * k, 0, 0)), intr, W, H, nan_fraction=nan_fraction, frame_index=k)
    out = synth.depth_to_cloud(depth, intr, scale, colors=colors)
    pts, col = out if colors else (out, None)
    return pts, col, pose


def _step(om, gm, integ, pts, col, pose, truncation=0.1, max_dist=5.0, what=""):
    om.integrate_pointcloud(pts, pose, col, truncation, max_dist)
    gm.IntegratePointCloud(integ, (pts, col), pose, truncation, max_dist)
    oc, gc = om.counters(), gm.counters(reset=True)
    pairs = [("sdf", "sdf"), ("carved", "carved"), ("visited", "probe"), ("candidates", "work_chunks"), ("updated_chunks", "updated_chunks")]
    for a, b in pairs:
        assert oc[a] == gc[b], "%s counter %s: oracle %d gpu %d" % (what, a, oc[a], gc[b])
    assert oc["created"] - oc["collected"] == gc["new_chunks"], what
    assert om.num_chunks() == gm.NumChunks(), what
    compare_fields(om.fields(), gm.fields(), om.V, om.use_color, atol=0.0, what=what)
    assert sorted(map(tuple, om.meshes_to_update().tolist())) == sorted(map(tuple, gm.GetMeshesToUpdate().tolist())), what
    return oc


def test_device_raycast_matches_the_oracle(hip_lib, oracle_mod):
    """RayWalk (kernels_cloud.h) against oracle.raycast (Raycast.cpp:35-128): random, axis-aligned, integer-valued, negative rays."""
    rng = np.random.default_rng(7)
    n = 4000
    a = rng.uniform(-6, 22, (n, 3)).astype(np.float32)
    b = (a + rng.uniform(-24, 24, (n, 3))).astype(np.float32)
    b[:400, 0] = a[:400, 0]                       # idle axis
    b[400:600, :2] = a[400:600, :2]               # two idle axes
    a[600:900] = np.round(a[600:900])             # starts on cell corners
    b[900:1200] = np.round(b[900:1200])
    a[1200:1300] = np.round(a[1200:1300]) - np.float32(1e-7)
    b[1300:1400] = a[1300:1400]                   # start == end
    a[1400:1500] *= np.float32(1e-3)              # inside one cell
    b[1400:1500] = a[1400:1500] + np.float32(1e-4)
    a[1500, 0] = np.nan
    b[1501, 2] = np.inf
    rays = np.ascontiguousarray(np.concatenate([a, b], axis=1))
    lo = np.array([0, 0, 0], np.int32)
    hi = np.array([16, 16, 16], np.int32)
    cap = 64
    cells = np.zeros((n, cap, 3), np.int32)
    count = np.zeros(n, np.int32)
    i32p = C.POINTER(C.c_int)
    rc = hip_lib.chisel_hip_kat_raycast(rays.ctypes.data_as(C.POINTER(C.c_float)), n, lo.ctypes.data_as(i32p), hi.ctypes.data_as(i32p),
                                        cells.ctypes.data_as(i32p), cap, count.ctypes.data_as(i32p))
    assert rc == 0
    nonempty = 0
    for i in range(n):
        ref = oracle_mod.raycast(a[i], b[i], lo, hi)
        assert count[i] == len(ref), "ray %d: %d cells, oracle %d" % (i, count[i], len(ref))
        assert np.array_equal(cells[i, :count[i]], ref), "ray %d" % i
        nonempty += len(ref) > 0
    assert nonempty > 500
    # chunk-level walk: unbounded box
    lo2 = np.array([-(2 ** 31 - 1)] * 3, np.int32)
    hi2 = np.array([2 ** 31 - 1] * 3, np.int32)
    cap2 = 256
    cells2 = np.zeros((n, cap2, 3), np.int32)
    rc = hip_lib.chisel_hip_kat_raycast(rays.ctypes.data_as(C.POINTER(C.c_float)), n, lo2.ctypes.data_as(i32p), hi2.ctypes.data_as(i32p),
                                        cells2.ctypes.data_as(i32p), cap2, count.ctypes.data_as(i32p))
    assert rc == 0
    for i in range(0, n, 7):
        ref = oracle_mod.raycast(a[i], b[i], lo2, hi2)
        assert count[i] == len(ref) and np.array_equal(cells2[i, :count[i]], ref), "ray %d (unbounded)" % i


def test_device_colour_update_any_weight(hip_lib):
    """color_integrate_any (division-free, used by cloud_integrate_kernel) == ColorVoxel::Integrate(r, g, b, 1) for every
    (weight, old, new); the general path itself is pinned to the reference-built vectors in test_gpu_parity.py."""
    bad = C.c_uint(12345)
    assert hip_lib.chisel_hip_kat_color_any(C.byref(bad)) == 0
    assert bad.value == 0


@pytest.mark.parametrize("N,res,color,trunc", [
    (16, 0.02, False, ("inverse", 2.0)),
    (16, 0.02, True, ("inverse", 2.0)),
    (8, 0.03, True, ("constant", 0.08)),
    (32, 0.01, True, ("quadratic", 4.0)),
    (16, 0.01, True, ("inverse", 1.0)),
])
def test_cloud_sequence_matches_the_oracle(hip_lib, oracle_mod, N, res, color, trunc):
    om, gm, integ = _mk(oracle_mod, N, res, color, trunc)
    W, H = 80, 60
    total = 0
    for k in range(4):
        pts, col, pose = _cloud("sphere_room", k, W, H, 0.6, color)
        oc = _step(om, gm, integ, pts, col, pose, what="cloud %d" % k)
        total += oc["sdf"]
    assert total > 10000 and om.num_chunks() > 5


def test_cloud_carving_and_offset_sensor(hip_lib, oracle_mod):
    """u = depth - ((inversePose * centroid).z - cameraPose.translation().z) (ProjectionIntegrator.cpp:89): a sensor translated
    along world z shifts every u, which sends voxels into the carve branch (weight decay by Integrate(1e-5, 5), :100)."""
    om, gm, integ = _mk(oracle_mod, 16, 0.02, True, ("constant", 0.06), carving_dist=0.01)
    W, H = 80, 60
    carved = 0
    for k, tz in enumerate([0.0, 0.1, 0.1, -0.05, 0.12]):
        pts, col, pose = _cloud("wall", k, W, H, 0.7, True, t=(0.0, 0.0, tz))
        oc = _step(om, gm, integ, pts, col, pose, what="cloud %d" % k)
        carved += oc["carved"]
    assert carved > 100


def test_cloud_depth_limit_shifts_the_colour_index(hip_lib, oracle_mod):
    """Points deeper than the limit are skipped without advancing the colour index (ProjectionIntegrator.cpp:130-132)."""
    om, gm, integ = _mk(oracle_mod, 16, 0.03, True, ("inverse", 2.0))
    W, H = 64, 48
    intr = synth.intrinsics(W, H)
    pose = synth.pose_yaw(10.0, (0.1, 0.0, 0.05))
    depth = synth.render_depth("box_room", pose, intr, W, H)   # depths 4.5 m to 5.5 m after scaling
    pts, col = synth.depth_to_cloud(depth * 2.0, intr, 1.0, colors=True)
    assert (pts[:, 2] > 5.0).sum() > 100 and (pts[:, 2] <= 5.0).sum() > 100
    _step(om, gm, integ, pts, col, pose, max_dist=20.0, what="coloured")
    # without colours the limit is 2 m (:68-70)
    om2, gm2, integ2 = _mk(oracle_mod, 16, 0.03, True, ("inverse", 2.0))
    pts2 = (pts * np.float32(0.4)).astype(np.float32)
    assert (pts2[:, 2] > 2.0).sum() > 100 and (pts2[:, 2] <= 2.0).sum() > 100
    _step(om2, gm2, integ2, pts2, None, pose, max_dist=20.0, what="plain")
    # colours given but a map without colour voxels: plain path (ProjectionIntegrator.cpp:42)
    om3, gm3, integ3 = _mk(oracle_mod, 16, 0.03, False, ("inverse", 2.0))
    _step(om3, gm3, integ3, pts2, col, pose, max_dist=20.0, what="no colour voxels")


def test_cloud_edge_cases(hip_lib, oracle_mod):
    om, gm, integ = _mk(oracle_mod, 16, 0.02, True)
    pose = synth.pose_yaw(5.0, (0.0, 0.1, 0.0))
    # empty cloud: nothing happens (Chisel.cpp:112-113)
    _step(om, gm, integ, np.zeros((0, 3), np.float32), None, pose, what="empty")
    assert gm.NumChunks() == 0
    # NaN / inf points, points behind the sensor, a point at the sensor, max_dist cutting the list
    pts, col, _ = _cloud("sphere_room", 0, 64, 48, 0.6, True)
    pts = pts.copy()
    pts[5] = np.nan
    pts[17, 0] = np.inf
    pts[40, 2] = np.nan
    pts[100] = 0.0
    pts[200:220, 2] *= -1.0
    _step(om, gm, integ, pts, col, pose, max_dist=1.45, what="odd points")
    # a second cloud over resident chunks, truncation larger than the chunk size
    pts2, col2, pose2 = _cloud("sphere_room", 2, 64, 48, 0.6, True)
    _step(om, gm, integ, pts2, col2, pose2, truncation=0.5, what="wide list")
    # zero truncation parameter: no chunk is listed (start and end of every chunk walk share a cell)
    n_before = gm.NumChunks()
    _step(om, gm, integ, pts2, col2, pose2, truncation=0.0, what="nothing listed")
    assert gm.NumChunks() == n_before


def test_cloud_between_depth_frames(hip_lib, oracle_mod):
    """Clouds and depth frames interleaved on one map: the cloud kernels are ordered against the two-stream depth pipeline."""
    from tests.common import make_frames, small_camera
    om, gm, integ = _mk(oracle_mod, 16, 0.03, True, ("inverse", 2.0))
    W, H = 64, 48
    cam = small_camera(W, H)
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    color_img = synth.render_color(W, H, 3)
    frames = make_frames("sphere_room", 6, W, H)
    for i, (depth, pose) in enumerate(frames):
        d = (depth * np.float32(0.6)).astype(np.float32)
        om.integrate_depth_color(d, pose, intr, color_img, near=cam.near_plane, far=cam.far_plane)
        gm.IntegrateDepthScanColor(integ, d, pose, cam, color_img, pose, cam)
        if i % 2 == 1:
            pts, col, cpose = _cloud("sphere_room", i, W, H, 0.6, True)
            om.integrate_pointcloud(pts, cpose, col, 0.1, 5.0)
            gm.IntegratePointCloud(integ, (pts, col), cpose, 0.1, 5.0)
    gm.counters(reset=True)
    assert om.num_chunks() == gm.NumChunks()
    compare_fields(om.fields(), gm.fields(), om.V, True, atol=0.0, what="interleaved")
    assert sorted(map(tuple, om.meshes_to_update().tolist())) == sorted(map(tuple, gm.GetMeshesToUpdate().tolist()))
    # and the meshes built from the result agree
    om.update_meshes(True)
    gm.UpdateMeshes(True)
    assert sorted(map(tuple, om.mesh_ids().tolist())) == sorted(map(tuple, gm.GetMeshIDs().tolist()))


def test_cloud_device_resident_and_full_size(hip_lib, oracle_mod):
    """640x480 cloud (the shape ChiselServer hands over), device-resident: two maps fed the same clouds agree bit for bit, a host-fed map
    agrees with them, and a 160x120 sub-sampling of the same scene agrees with the oracle."""
    import torch
    from cvids_amd import chisel as ch
    W, H = 640, 480
    maps = [ch.Chisel((16, 16, 16), 0.01, True, max_chunks=20000) for _ in range(3)]
    integ = ch.ProjectionIntegrator(ch.InverseTruncator(1.0), ch.ConstantWeighter(1.0), 0.05, True)
    for k in range(2):
        pts, col, pose = _cloud("sphere_room", k, W, H, 0.6, True, t=(0.0, 0.0, 0.03 * k))
        tp, tc = torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda()
        maps[0].IntegratePointCloud(integ, ch.PointCloud(tp, tc), pose, 0.1, 5.0)
        maps[1].IntegratePointCloud(integ, (tp, tc), pose, 0.1, 5.0)
        maps[2].IntegratePointCloud(integ, (pts, col), pose, 0.1, 5.0)
    c = [m.counters(reset=True) for m in maps]
    assert c[0] == c[1] == c[2]
    assert c[0]["sdf"] > 5_000_000 and c[0]["work_chunks"] > 500
    f0 = maps[0].fields()
    compare_fields(f0, maps[1].fields(), 4096, True, atol=0.0, what="device twice")
    compare_fields(f0, maps[2].fields(), 4096, True, atol=0.0, what="device vs host")
    # oracle at a size it finishes in seconds
    om, gm, integ2 = _mk(oracle_mod, 16, 0.01, True, ("inverse", 1.0), max_chunks=8192)
    for k in range(2):
        pts, col, pose = _cloud("sphere_room", k, 160, 120, 0.6, True, t=(0.0, 0.0, 0.03 * k))
        _step(om, gm, integ2, pts, col, pose, what="160x120 cloud %d" % k)


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_clouds_equal_the_unsharded_map(hip_lib, world):
    """Each shard lists and updates only the chunks it owns: the union of the shards is the unsharded map, chunk for chunk."""
    from cvids_amd import chisel as ch
    from cvids_amd.sharded import LocalShardGroup
    one = ch.Chisel((16, 16, 16), 0.02, True, max_chunks=4096)
    shards = [ch.Chisel((16, 16, 16), 0.02, True, max_chunks=4096, n_shards=world, shard_rank=r, shard_block=2) for r in range(world)]
    group = LocalShardGroup(shards)
    integ = ch.ProjectionIntegrator(ch.InverseTruncator(2.0), ch.ConstantWeighter(1.0), 0.05, True)
    for k in range(3):
        pts, col, pose = _cloud("sphere_room", k, 96, 72, 0.6, True, t=(0.0, 0.0, 0.02 * k))
        one.IntegratePointCloud(integ, (pts, col), pose, 0.1, 5.0)
        group.IntegratePointCloud(integ, (pts, col), pose, 0.1, 5.0)
    want = one.fields()
    got = {}
    for r, s_ in enumerate(shards):
        f = s_.fields()
        assert all(ch.chunk_owner(cid, world, 2) == r for cid in f)
        assert not (set(f) & set(got))
        got.update(f)
    assert len(want) > 20 and sum(1 for s_ in shards if s_.NumChunks() > 0) == world
    compare_fields(want, got, 4096, True, atol=0.0, what="sharded clouds")
    c1 = one.counters()
    cs = [s_.counters() for s_ in shards]
    for k in ("sdf", "col", "probe", "carved", "new_chunks", "updated_chunks", "work_chunks"):
        assert c1[k] == sum(c[k] for c in cs), k


def _random_pose(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    T = np.eye(4)
    T[:3, :3] = R
    # the reference subtracts the sensor's world z from a sensor-frame z (ProjectionIntegrator.cpp:89): only small z offsets fuse anything
    T[:3, 3] = [rng.uniform(-1.0, 1.0), rng.uniform(-1.0, 1.0), rng.uniform(-0.03, 0.03)]
    return T.astype(np.float32)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_unorganised_clouds_and_arbitrary_poses(hip_lib, oracle_mod, seed):
    """Points in no particular order (nothing like image rows), sensors rotated about every axis, negative chunk ids: the
    result depends on the cloud order only through the per-voxel sequence of updates, which must still be the reference's."""
    rng = np.random.default_rng(seed)
    N = [16, 8, 16, 32][seed - 1]
    om, gm, integ = _mk(oracle_mod, N, 0.02, True, [("inverse", 2.0), ("constant", 0.07), ("quadratic", 3.0), ("inverse", 1.5)][seed - 1],
                        carving_dist=0.02)
    for k in range(3):
        n = 6000
        # a bumpy sheet in front of the sensor, visited in random order, with repeated points
        uv = rng.uniform(-0.5, 0.5, (n, 2))
        z = 1.1 + 0.15 * np.sin(5 * uv[:, 0]) * np.cos(4 * uv[:, 1]) + rng.normal(0, 0.004, n)
        pts = np.stack([uv[:, 0] * z, uv[:, 1] * z, z], axis=1).astype(np.float32)
        pts[100:200] = pts[0:100]
        col = rng.uniform(0, 1, (n, 3)).astype(np.float32)
        pose = _random_pose(rng)
        _step(om, gm, integ, pts, col if k != 1 else None, pose, truncation=0.1 + 0.1 * k, max_dist=3.0, what="seed %d cloud %d" % (seed, k))
    assert om.num_chunks() > 3


def test_cloud_outside_the_supported_envelope_is_reported(hip_lib, oracle_mod):
    """Chunk ids beyond +-2^20 cannot be packed: the call that waits reports CHISEL_HIP_ERR_UNSUPPORTED once, the map stays usable
    and a following cloud still matches the oracle."""
    from cvids_amd.capi import ChiselHipError
    om, gm, integ = _mk(oracle_mod, 16, 0.02, True)
    pose = synth.pose_yaw(0.0, (0.0, 0.0, 0.0))
    far = np.array([[4.0e5, 0.0, 1.0], [0.0, 0.0, 1.0]], np.float32)   # the first point lies 400 km to the side
    gm.IntegratePointCloud(integ, (far, None), pose, 0.1, 1e9)
    with pytest.raises(ChiselHipError) as e:
        gm.synchronize()
    assert e.value.code == 5 and "point cloud" in str(e.value)
    gm.synchronize()                       # reported once
    gm.Reset()
    gm.counters(reset=True)
    pts, col, pose = _cloud("sphere_room", 1, 64, 48, 0.6, True)
    _step(om, gm, integ, pts, col, pose, what="after the error")


def test_cloud_produced_on_another_stream_is_ordered_by_event(hip_lib):
    """chisel_hip.h: device clouds may be ordered by an event.  The cloud is written on a second stream behind a long kernel; the
    map is told to wait for the event recorded there (chisel_hip_wait_event) and must read the finished cloud -- the same map as
    from a host cloud.  Afterwards the map's depth path still works (the event was consumed by the cloud call, not left armed)."""
    import torch
    from cvids_amd import chisel as ch
    integ = ch.ProjectionIntegrator(ch.InverseTruncator(1.0), ch.ConstantWeighter(1.0), 0.05, True)
    pts, col, pose = _cloud("sphere_room", 0, 320, 240, 0.6, True)
    ref = ch.Chisel((16, 16, 16), 0.01, True, max_chunks=20000)
    ref.IntegratePointCloud(integ, (pts, col), pose, 0.1, 5.0)
    want = ref.fields()
    gm = ch.Chisel((16, 16, 16), 0.01, True, max_chunks=20000)
    side = torch.cuda.Stream()
    tp = torch.full((len(pts), 3), float("nan"), device="cuda")  # garbage until the producer has run
    tc = torch.zeros((len(pts), 3), device="cuda")
    hp, hc = torch.from_numpy(pts).pin_memory(), torch.from_numpy(col).pin_memory()
    torch.cuda.synchronize()
    ev = torch.cuda.Event()
    with torch.cuda.stream(side):
        busy = torch.randn(4096, 4096, device="cuda")
        for _ in range(20):  # ~ tens of milliseconds in front of the copies
            busy = busy @ busy * 1e-3
        tp.copy_(hp, non_blocking=True)
        tc.copy_(hc, non_blocking=True)
        ev.record(side)
    gm.wait_event(ev.cuda_event)
    gm.IntegratePointCloud(integ, (tp, tc), pose, 0.1, 5.0)
    compare_fields(want, gm.fields(), 4096, True, atol=0.0, what="cloud behind an event")
    # the event is gone: a depth frame right after must not wait on (or trip over) it
    del ev
    cam = ch.PinholeCamera(*synth.intrinsics(64, 48), 64, 48, 0.05, 5.0)
    depth, p2 = list(synth.stream("sphere_room", 1, 64, 48))[0]
    gm.IntegrateDepthScanColor(integ, depth, p2, cam, synth.render_color(64, 48, 3), p2, cam)
    gm.synchronize()
    ref.close()
    gm.close()


def test_cloud_report_does_not_hide_pool_exhaustion(hip_lib):
    """The map keeps two error words: a point cloud's one-off report must not overwrite "chunk pool exhausted", and a mesh
    recompute after a cloud report must not be taken for pool exhaustion."""
    from cvids_amd import chisel as ch
    from cvids_amd.capi import ChiselHipError
    integ = ch.ProjectionIntegrator(ch.InverseTruncator(2.0), ch.ConstantWeighter(1.0), 0.05, True)
    cam = ch.PinholeCamera(*synth.intrinsics(64, 48), 64, 48, 0.05, 5.0)
    depth, pose = list(synth.stream("sphere_room", 1, 64, 48))[0]
    far = np.array([[2.0e6, 0.0, 1.0], [0.0, 0.0, 1.0]], np.float32)  # chunk id 5e6: beyond +-2^20
    # (1) cloud report, then a recompute: the recompute succeeds, the report comes once as UNSUPPORTED
    gm = ch.Chisel((8, 8, 8), 0.05, False, max_chunks=4096)
    gm.IntegrateDepthScan(integ, depth, pose, cam)
    gm.IntegratePointCloud(integ, (far, None), synth.pose_yaw(0.0), 0.1, 1e9)
    gm.UpdateMeshes(force=True)
    with pytest.raises(ChiselHipError) as e:
        gm.synchronize()
    assert e.value.code == 5
    assert len(gm.GetMeshIDs()) > 10      # the recompute was not abandoned
    gm.synchronize()
    gm.close()
    # (2) pool exhaustion, then a cloud report on top: the wait still says POOL_FULL, and keeps saying so until Reset
    gm = ch.Chisel((8, 8, 8), 0.05, False, max_chunks=16)
    gm.IntegrateDepthScan(integ, depth, pose, cam)
    try:
        gm.IntegratePointCloud(integ, (far, None), synth.pose_yaw(0.0), 0.1, 1e9)
    except ChiselHipError as e0:  # a call that happens to wait may already see the exhausted pool
        assert e0.code == 3
    for _ in range(2):
        with pytest.raises(ChiselHipError) as e:
            gm.synchronize()
        assert e.value.code == 3
    gm.Reset()
    gm.synchronize()
    gm.close()


@pytest.mark.parametrize("N,res", [(8, 0.05), (16, 0.02), (32, 0.02)])
def test_cloud_candidates_match_the_oracle(hip_lib, oracle_mod, N, res):
    """ChunkManager::GetChunkIDsIntersecting(cloud, cameraTransform, truncation, maxDist, chunkList) (ChunkManager.cpp:214-257) on its own:
    chisel_hip_cloud_candidates (the listing kernel of the point-cloud mode) against the oracle's restatement -- same set of ids for
    organised clouds under yawed poses, random clouds under arbitrary poses, a far limit that drops points, and an empty cloud; the map
    itself is left untouched."""
    om, gm, integ = _mk(oracle_mod, N, res, False)
    rng = np.random.default_rng(11)
    cases = []
    for k in range(3):
        pts, _, pose = _cloud("sphere_room", k, 96, 72, 1.5, False)
        cases.append((pts, pose, 0.1, 5.0))
    cases.append((cases[0][0], cases[0][1], 0.25, 1.9))                      # longer segments, points beyond the far limit skipped
    cases.append((rng.uniform(-1.5, 1.5, (3000, 3)).astype(np.float32), _random_pose(rng), 0.1, 2.0))
    cases.append((np.zeros((0, 3), np.float32), synth.pose_yaw(0.0), 0.1, 5.0))
    counts = []
    for i, (pts, pose, trunc, far) in enumerate(cases):
        want = om.cloud_chunk_ids(pts, pose, trunc, far)
        got = gm.CloudCandidates((pts, None), pose, trunc, far)
        assert got.shape == want.shape and np.array_equal(got, want), "case %d: %d vs %d ids" % (i, len(got), len(want))
        counts.append(len(want))
    assert max(counts) > 10 and sum(c > 0 for c in counts) >= 3 and counts[-1] == 0, counts
    assert gm.NumChunks() == 0


@pytest.mark.parametrize("N", [8, 16])
def test_mesh_of_one_cube_matches_the_oracle(hip_lib, oracle_mod, N):
    """ChunkManager::ExtractInsideVoxelMesh / ExtractBorderVoxelMesh (ChunkManager.cpp:259-379) through chisel_hip_mesh_cube: for cubes inside
    a chunk, on its +x / +y / +z faces (corners in the neighbour chunks) and with index -1 (corners in the -x neighbour), the vertices
    and face normals are those of the oracle's MeshCube over the same eight corner distances; a cube with an unobserved corner or an
    absent neighbour yields nothing."""
    from tests.common import make_frames, small_camera
    from cvids_amd import chisel as ch
    res = 0.04
    om, gm, integ = _mk(oracle_mod, N, res, False)
    cam = small_camera(96, 72)
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    for d, p in make_frames("sphere_room", 3, 96, 72):
        om.integrate_depth(d, p, intr, cam.near_plane, cam.far_plane)
        gm.IntegrateDepthScan(integ, d, p, cam)
    ids = [tuple(i) for i in om.chunk_ids().tolist()]
    chunks = {i: om.get_chunk(i) for i in ids}
    off = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]  # cubeIndexOffsets, ChunkManager.cpp:67-69

    def corners(cid, v):
        s = []
        for o in off:
            c = [v[a] + o[a] for a in range(3)]
            nid = list(cid)
            for a in range(3):
                if c[a] < 0:
                    nid[a] -= 1
                    c[a] = N - 1
                elif c[a] >= N:
                    nid[a] += 1
                    c[a] = 0
            ck = chunks.get(tuple(nid))
            if ck is None:
                return None
            lin = (c[2] * N + c[1]) * N + c[0]
            if not ck[1].reshape(-1)[lin] > 0.5:
                return None
            s.append(ck[0].reshape(-1)[lin])
        return np.asarray(s, np.float32)

    L, checked, empty = gm.L, 0, 0
    step = 1 if N == 8 else 3
    with_voxels = [c for c in ids if (chunks[c][1] > 0.5).sum() > 50][:5]
    for cid in with_voxels:
        picks = [(x, y, z) for z in range(-1, N, step) for y in range(-1, N, step) for x in range(-1, N, step)]
        for v in picks:
            coords = (np.asarray(v, np.float32) * np.float32(res) + np.float32(res * 0.5)) + np.asarray(cid, np.float32) * np.float32(N * res)
            ve, no = np.zeros((15, 3), np.float32), np.zeros((15, 3), np.float32)
            nv, occ = C.c_int(0), C.c_int(0)
            idv, vv = (C.c_int * 3)(*cid), (C.c_int * 3)(*[int(x) for x in v])
            fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
            rc = L.chisel_hip_mesh_cube(gm.h, idv, vv, fp(coords), fp(ve), fp(no), C.byref(nv), C.byref(occ))
            assert rc == 0
            s = corners(cid, v)
            if s is None:
                assert nv.value == 0 and occ.value == 0
                empty += 1
                continue
            wv, wn = oracle_mod.mesh_cube(s, coords, res)
            assert nv.value == len(wv) and occ.value == (1 if len(wv) else 0)
            assert np.array_equal(ve[:nv.value].view(np.uint32), wv.view(np.uint32)) and np.array_equal(no[:nv.value].view(np.uint32), wn.view(np.uint32))
            checked += len(wv) > 0
    assert checked > 20 and empty > 5
