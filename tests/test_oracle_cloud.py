"""CPU checks of the oracle's point-cloud fusion mode (oracle/chisel_oracle.cpp: Raycast, GetChunkIDsIntersectingCloud,
IntegrateCloudChunk, IntegratePointCloudScan; reference Chisel.cpp:107-157, ProjectionIntegrator.cpp:52-173, Raycast.cpp:4-128).

The reference has no test or fixture for this path and Raycast.cpp cannot be compiled here (it includes Eigen through
Geometry.h), so the restatement is "parity unpinned"; these tests pin its arithmetic to hand-derived cases of the reference's
lines and its structure to properties the algorithm has.
"""
import numpy as np

from cvids_amd import synth

BIG = 2 ** 31 - 1


def test_raycast_hand_derived_case(oracle_mod):
    """Raycast.cpp:9-33: mod() calls the double fmod (the float overloads live in std::), so a start coordinate just below a
    cell boundary gives s = (float)(1 - 1e-9) = 1 and tMax = 0: the first step is along that axis.  With fmodf the first step
    would be along y."""
    cells = oracle_mod.raycast([-1e-9, 0.5, 0.5], [1.5, 2.5, 0.5], [-BIG] * 3, [BIG] * 3)
    assert cells.tolist() == [[-1, 0, 0], [0, 0, 0], [0, 1, 0], [1, 1, 0], [1, 2, 0]]
    # ties go to z, then y (Raycast.cpp:96-125: strict '<' comparisons)
    cells = oracle_mod.raycast([0.5, 0.5, 0.5], [1.5, 1.5, 1.5], [-BIG] * 3, [BIG] * 3)
    assert cells.tolist() == [[0, 0, 0], [0, 0, 1], [0, 1, 1], [1, 1, 1]]
    # start and end in the same cell: nothing, not even that cell (:79-80)
    assert len(oracle_mod.raycast([0.2, 0.2, 0.2], [0.8, 0.3, 0.9], [-BIG] * 3, [BIG] * 3)) == 0
    # negative direction: intbound(-s, -ds)
    cells = oracle_mod.raycast([2.25, 0.5, 0.5], [-0.5, 0.75, 0.5], [-BIG] * 3, [BIG] * 3)
    assert cells.tolist() == [[2, 0, 0], [1, 0, 0], [0, 0, 0], [-1, 0, 0]]


def test_raycast_structure(oracle_mod):
    rng = np.random.default_rng(11)
    for _ in range(400):
        a = rng.uniform(-30, 30, 3).astype(np.float32)
        b = (a + rng.uniform(-25, 25, 3)).astype(np.float32)
        cells = oracle_mod.raycast(a, b, [-BIG] * 3, [BIG] * 3)
        fa, fb = np.floor(a).astype(int), np.floor(b).astype(int)
        if (fa == fb).all():
            assert len(cells) == 0
            continue
        # a 6-connected path from the start cell to the end cell, monotone on every axis, one step per cell boundary
        assert (cells[0] == fa).all() and (cells[-1] == fb).all()
        d = np.diff(cells, axis=0)
        assert (np.abs(d).sum(axis=1) == 1).all()
        assert len(cells) == np.abs(fb - fa).sum() + 1
        for k in range(3):
            assert (d[:, k] * np.sign(fb[k] - fa[k]) >= 0).all()
        # the clipped walk is the unclipped one filtered by the box (Raycast.cpp:83-86)
        lo, hi = np.array([-4, -4, -4]), np.array([12, 12, 12])
        inside = cells[((cells >= lo) & (cells < hi)).all(axis=1)]
        assert np.array_equal(oracle_mod.raycast(a, b, lo, hi), inside)
    # coordinates that are not finite meet no cell
    assert len(oracle_mod.raycast([np.nan, 0, 0], [5, 5, 5], [-BIG] * 3, [BIG] * 3)) == 0
    assert len(oracle_mod.raycast([0, 0, 0], [5, np.inf, 5], [-BIG] * 3, [BIG] * 3)) == 0


def test_affine_inverse(oracle_mod):
    rng = np.random.default_rng(3)
    for k in range(20):
        pose = synth.pose_yaw(17.0 * k, rng.uniform(-2, 2, 3)).astype(np.float32)
        inv = oracle_mod.invert_pose(pose)
        full = np.eye(4)
        full[:3] = inv
        assert np.allclose(full @ pose.astype(np.float64), np.eye(4), atol=2e-6)
    # a pure translation inverts exactly
    pose = np.eye(4, dtype=np.float32)
    pose[:3, 3] = [0.25, -1.5, 3.0]
    assert np.array_equal(oracle_mod.invert_pose(pose), np.array([[1, 0, 0, -0.25], [0, 1, 0, 1.5], [0, 0, 1, -3.0]], np.float32))


def test_single_point_cloud(oracle_mod):
    """One point straight ahead at identity pose: the voxels met are the cells of the segment point -+ truncation, the first
    update writes sdf = u = depth - voxel centre z and weight = w / (5 truncation) (ProjectionIntegrator.cpp:85-95)."""
    res, N = np.float32(0.05), 8
    om = oracle_mod.OracleMap(N, float(res), False)
    om.set_integrator(oracle_mod.TRUNC_CONSTANT, 0.12, 2.0, True, 0.05)
    pts = np.array([[0.01, 0.02, 1.01]], np.float32)
    # chunk listing (ChunkManager.cpp:214-257): a segment point -+ truncation that stays inside one chunk lists NOTHING
    # (Raycast returns no cell when start and end share one, Raycast.cpp:79-80) -- 0.1 m in a 0.4 m chunk here
    om.integrate_pointcloud(pts, np.eye(4, dtype=np.float32), None, 0.1, 5.0)
    assert om.counters()["candidates"] == 0 and om.num_chunks() == 0
    om.integrate_pointcloud(pts, np.eye(4, dtype=np.float32), None, 0.3, 5.0)
    c = om.counters()
    assert c["candidates"] >= 1 and c["sdf"] > 0 and c["created"] - c["collected"] == om.num_chunks()
    fields = om.fields()
    touched = 0
    tau = np.float32(0.12)
    for cid, (sdf, w, _) in fields.items():
        idx = np.nonzero(w > 0)[0]
        for i in idx:
            x, y, z = i % N, (i // N) % N, i // (N * N)
            assert (x, y) == (0, 0) and cid[0] == 0 and cid[1] == 0      # the ray stays in the first voxel column
            centre_z = (np.float32(z) * res + res * np.float32(0.5)) + np.float32(N * cid[2]) * res
            u = np.float32(1.01) - centre_z
            assert abs(u) < tau
            assert sdf[i] == u and w[i] == np.float32(2.0) / (np.float32(5) * tau)
            touched += 1
    # the segment 1.01 -+ 0.12 covers the cells 17 .. 22 along z; only those with |u| < truncation are written
    assert touched == c["sdf"] and 4 <= touched <= 6


def test_colour_index_skips_rejected_points(oracle_mod):
    """ProjectionIntegrator.cpp:124-132: the colour index advances only on points that pass the 5 m limit, so a rejected point
    shifts the colours of all later points by one."""
    res, N = 0.05, 8
    om = oracle_mod.OracleMap(N, res, True)
    om.set_integrator(oracle_mod.TRUNC_CONSTANT, 0.1, 1.0, False, 0.05)
    pts = np.array([[0.0, 0.0, 6.0], [0.02, 0.02, 1.0], [0.52, 0.02, 1.0]], np.float32)
    cols = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]], np.float32)
    om.integrate_pointcloud(pts, np.eye(4, dtype=np.float32), cols, 0.5, 20.0)
    seen = set()
    for cid, (sdf, w, rgbw) in om.fields().items():
        for i in np.nonzero(rgbw[:, 3] > 0)[0]:
            x = i % N + N * cid[0]
            seen.add((x >= 5, tuple(int(v) for v in rgbw[i, :3])))
    # the point at x = 0.02 takes colours[0] (red), the point at x = 0.52 takes colours[1] (green); blue is never used
    assert seen == {(False, (255, 0, 0)), (True, (0, 255, 0))}


def test_cloud_sequence_is_plausible(oracle_mod):
    """A wall seen through three clouds: weights grow with every cloud, the zero crossing of the SDF sits on the wall."""
    om = oracle_mod.OracleMap(16, 0.02, False)
    om.set_integrator(oracle_mod.TRUNC_CONSTANT, 0.08, 1.0, True, 0.05)
    W, H = 48, 36
    intr = synth.intrinsics(W, H)
    wsum = []
    for k in range(3):
        pose = synth.pose_yaw(0.0, (0.01 * k, 0.0, 0.0))
        depth = synth.render_depth("wall", pose, intr, W, H)
        pts = synth.depth_to_cloud(depth, intr, 0.6)   # wall at z = 1.2
        om.integrate_pointcloud(pts, pose, None, 0.1, 5.0)
        assert om.counters()["sdf"] > 1000
        wsum.append(sum(float(w.sum()) for _, w, _ in om.fields().values()))
    assert wsum[0] < wsum[1] < wsum[2]
    ok, d_front = om.get_sdf((0.01, 0.01, 1.17))
    ok2, d_back = om.get_sdf((0.01, 0.01, 1.23))
    assert ok and ok2 and d_front > 0 > d_back
    assert len(om.meshes_to_update()) > 0


def test_cloud_chunk_listing_hand_derived(oracle_mod):
    """ChunkManager.cpp:214-257 on its own (oc_cloud_chunk_ids): 8-voxel chunks of 5 cm = 0.4 m.  A point at z = 0.79 straight ahead with
    truncation 0.1: the segment z in [0.69, 0.89] crosses the chunk boundary at 0.8 -> chunks (0,0,1) and (0,0,2); a point whose segment
    stays inside one chunk lists nothing (Raycast.cpp:79-80); a point beyond maxDist is skipped (:231-234)."""
    om = oracle_mod.OracleMap(8, 0.05, False)
    pose = synth.pose_yaw(0.0)
    ids = om.cloud_chunk_ids(np.array([[0.01, 0.01, 0.79]], np.float32), pose, 0.1, 5.0)
    assert ids.tolist() == [[0, 0, 1], [0, 0, 2]]
    assert len(om.cloud_chunk_ids(np.array([[0.01, 0.01, 0.60]], np.float32), pose, 0.1, 5.0)) == 0
    assert len(om.cloud_chunk_ids(np.array([[0.01, 0.01, 0.79]], np.float32), pose, 0.1, 0.5)) == 0
    # the listing is a set: two points through the same boundary list it once
    ids = om.cloud_chunk_ids(np.array([[0.01, 0.01, 0.79], [0.02, 0.01, 0.81]], np.float32), pose, 0.1, 5.0)
    assert ids.tolist() == [[0, 0, 1], [0, 0, 2]]
    assert om.num_chunks() == 0
