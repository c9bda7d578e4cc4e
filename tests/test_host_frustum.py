"""Product host arithmetic (csrc/host_frustum.h) against the oracle's restatement of Frustum / GetChunkIDsIntersecting.
CPU only: the function under test runs on the host."""
import ctypes as C

import numpy as np
import pytest

from cvids_amd import synth


def product_range(lib, pose, near, far, fy, cy, W, H, N, res):
    p = np.ascontiguousarray(np.asarray(pose, np.float32)[:3, :4])
    rmin, rdim = (C.c_int * 3)(), (C.c_int * 3)()
    planes, corners = np.zeros(24, np.float32), np.zeros(24, np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    lib.chisel_hip_debug_frustum_range(fp(p), near, far, fy, cy, W, H, N, res, rmin, rdim, fp(planes), fp(corners))
    return list(rmin), list(rdim), planes.reshape(6, 4), corners.reshape(8, 3)


@pytest.mark.parametrize("W,H,N,res,near,far", [(640, 480, 16, 0.02, 0.05, 5.0), (640, 480, 16, 0.01, 0.05, 5.0),
                                                (1280, 720, 16, 0.005, 0.05, 5.0), (64, 48, 8, 0.05, 0.3, 5.0),
                                                (640, 480, 32, 0.01, 0.05, 5.0), (640, 480, 8, 0.10, 0.3, 5.0)])
def test_range_and_planes_match_oracle(hip_lib, oracle_mod, W, H, N, res, near, far):
    intr = synth.intrinsics(W, H)
    poses = [synth.trajectory_pose(k, a) for k in (0, 7, 33, 199) for a in (0, 1, 3)]
    poses.append(synth.pose_yaw(37.0, (50.0, -3.0, 200.0)))
    poses.append(synth.pose_yaw(-120.0, (-30.0, 0.5, -0.25)))
    om = oracle_mod.OracleMap(N, res, False)
    for pose in poses:
        rmin, rdim, planes, corners = product_range(hip_lib, pose, near, far, intr[1], intr[3], W, H, N, res)
        oc, op, oa = oracle_mod.frustum(pose, near, far, intr[1], intr[3], W, H)
        assert np.array_equal(corners, oc)
        assert np.array_equal(planes, op)
        if rdim[0] * rdim[1] * rdim[2] > 400000:
            continue
        cand = om.candidates(pose, intr, W, H, near, far)
        # the reference enumerates the full box [rmin, rmin+rdim) whenever the plane test passes;
        # every enumerated id must lie inside the product's range and the range must be tight
        assert len(cand) <= rdim[0] * rdim[1] * rdim[2]
        assert (cand.min(0) >= np.array(rmin)).all() and (cand.max(0) < np.array(rmin) + np.array(rdim)).all()
        if len(cand) == rdim[0] * rdim[1] * rdim[2]:
            assert list(cand.min(0)) == rmin and list(cand.max(0) - cand.min(0) + 1) == rdim


def test_frustum_export_matches_oracle(hip_lib, oracle_mod):
    """chisel_hip_frustum (the boundary's PinholeCamera::SetupFrustum: what chisel_ros draws, ChiselServer.cpp:97-134): corners and
    planes equal the oracle's restatement of Frustum::SetFromVectors bit for bit, the 24 line end points are the corners in the
    order of Frustum.cpp:190-217, NULL outputs are accepted."""
    W, H = 640, 480
    intr = synth.intrinsics(W, H)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    order = [0, 1, 3, 2, 1, 3, 2, 0, 4, 7, 6, 5, 5, 7, 6, 4, 0, 5, 1, 6, 2, 7, 3, 4]
    for pose in [synth.trajectory_pose(k, a) for k in (0, 50, 199) for a in (0, 2)] + [synth.pose_yaw(-120.0, (-30.0, 0.5, -0.25))]:
        p = np.ascontiguousarray(np.asarray(pose, np.float32)[:3, :4])
        corners, lines, planes = np.zeros((8, 3), np.float32), np.zeros((24, 3), np.float32), np.zeros((6, 4), np.float32)
        assert hip_lib.chisel_hip_frustum(fp(p), intr[1], intr[3], W, H, 0.05, 5.0, fp(corners), fp(lines), fp(planes)) == 0
        oc, op, _ = oracle_mod.frustum(pose, 0.05, 5.0, intr[1], intr[3], W, H)
        assert np.array_equal(corners, oc) and np.array_equal(planes, op)
        assert np.array_equal(lines, corners[order])
        assert hip_lib.chisel_hip_frustum(fp(p), intr[1], intr[3], W, H, 0.05, 5.0, None, fp(lines), None) == 0
    assert hip_lib.chisel_hip_frustum(None, 1.0, 1.0, W, H, 0.05, 5.0, None, None, None) == 1


def test_candidates_export_equals_the_oracles_enumeration(hip_lib, oracle_mod):
    """chisel_hip_candidates (ChunkManager::GetChunkIDsIntersecting(frustum), ChunkManager.cpp:182-212, host arithmetic as in the
    reference) against the oracle's oc_candidates: the same ids in the same order (x outer, z inner)."""
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    for W, H, N, res, near, far in [(640, 480, 16, 0.04, 0.05, 5.0), (64, 48, 8, 0.05, 0.3, 5.0), (320, 240, 32, 0.02, 0.05, 3.0)]:
        intr = synth.intrinsics(W, H)
        om = oracle_mod.OracleMap(N, res, False)
        for pose in [synth.trajectory_pose(k, a) for k in (0, 33) for a in (0, 3)] + [synth.pose_yaw(-120.0, (-30.0, 0.5, -0.25))]:
            p = np.ascontiguousarray(np.asarray(pose, np.float32)[:3, :4])
            corners, planes = np.zeros((8, 3), np.float32), np.zeros((6, 4), np.float32)
            assert hip_lib.chisel_hip_frustum(fp(p), intr[1], intr[3], W, H, near, far, fp(corners), None, fp(planes)) == 0
            cs = np.array([N, N, N], np.int32)
            n = C.c_int64(0)
            assert hip_lib.chisel_hip_candidates(fp(corners), fp(planes), ip(cs), res, None, 0, C.byref(n)) == 0
            ids = np.zeros((n.value, 3), np.int32)
            assert hip_lib.chisel_hip_candidates(fp(corners), fp(planes), ip(cs), res, ip(ids), n.value, C.byref(n)) == 0
            want = om.candidates(pose, intr, W, H, near, far)
            assert np.array_equal(ids, want), (W, H, N, len(ids), len(want))


def test_frustum_from_vectors_equals_set_from_params(hip_lib):
    """Frustum::SetFromParams is SetFromVectors on the vectors it derives from the pose (Frustum.cpp:143-153): the export of the latter
    (chisel_hip_frustum_from_vectors, what the facade's Frustum::SetFromVectors / SetFromOpenGLViewProjection call) reproduces the former
    bit for bit."""
    import ctypes as C
    import math
    f32 = C.c_float
    hip_lib.chisel_hip_frustum_from_vectors.argtypes = [C.POINTER(f32)] * 4 + [f32] * 4 + [C.POINTER(f32)] * 3
    hip_lib.chisel_hip_frustum.argtypes = [C.POINTER(f32), f32, f32, C.c_int, C.c_int, f32, f32] + [C.POINTER(f32)] * 3
    rng = np.random.default_rng(5)
    for _ in range(20):
        a = rng.normal(size=(3, 3))
        q, _r = np.linalg.qr(a)
        pose = np.zeros((3, 4), np.float32)
        pose[:, :3] = q.astype(np.float32)
        pose[:, 3] = rng.uniform(-3, 3, 3).astype(np.float32)
        W, H, fy, cy, near, far = 640, 480, np.float32(525.0), np.float32(239.5), np.float32(0.05), np.float32(5.0)
        c0, l0, p0 = (f32 * 24)(), (f32 * 72)(), (f32 * 24)()
        flat = (f32 * 12)(*pose.reshape(-1).tolist())
        assert hip_lib.chisel_hip_frustum(flat, fy, cy, W, H, near, far, c0, l0, p0) == 0
        vec = lambda v: (f32 * 3)(*[float(x) for x in v])
        aspect = np.float32(np.float32(fy * np.float32(W)) / np.float32(fy * np.float32(H)))
        fov = np.float32(math.atan2(float(cy), float(fy)) + math.atan2(float(np.float32(H) - cy), float(fy)))
        c1, l1, p1 = (f32 * 24)(), (f32 * 72)(), (f32 * 24)()
        assert hip_lib.chisel_hip_frustum_from_vectors(vec(pose[:, 2]), vec(pose[:, 3]), vec(pose[:, 0]), vec(-pose[:, 1]), near, far, fov, aspect, c1, l1, p1) == 0
        assert bytes(c0) == bytes(c1) and bytes(l0) == bytes(l1) and bytes(p0) == bytes(p1)
