"""The C++ facade (cvids_amd/open_chisel/include/open_chisel/*.h: the reference's class surface over the C ABI) driven the
way chisel_ros::ChiselServer drives OpenChisel, against the Python host path and the oracle on the same frames."""
import os
import subprocess

import numpy as np
import pytest

from cvids_amd import synth
from tests.common import compare_fields

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_facade_matches_oracle(oracle_mod, tmp_path):
    tdir = os.path.join(ROOT, "cvids_amd", "open_chisel", "tests")
    subprocess.check_call(["make", "-C", tdir, "build"])
    dump = str(tmp_path / "fields.bin")
    out = subprocess.run([os.path.join(tdir, "facade_smoke"), dump], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "chunks" in out.stdout and "found 1" in out.stdout
    # the same three frames through the oracle
    W, H, N, res = 64, 48, 8, 0.05
    om = oracle_mod.OracleMap(N, res, True)
    om.set_integrator(oracle_mod.TRUNC_INVERSE, 2.0, 1.0, True, 0.05)
    color = synth.render_color(W, H, 3)
    intr = (52.5, 52.5, 31.5, 23.5)
    pose = np.eye(4, dtype=np.float32)
    for k in range(3):
        om.integrate_depth_color(np.full((H, W), np.float32(1.5) + np.float32(0.1) * np.float32(k), np.float32), pose, intr, color)
    # and the coloured cloud the program integrates afterwards (same fp32 expressions)
    u = np.arange(W, dtype=np.float32)[None, :].repeat(H, 0)
    v = np.arange(H, dtype=np.float32)[:, None].repeat(W, 1)
    z = np.float32(1.2)
    pts = np.stack([((u - np.float32(31.5)) / np.float32(52.5)) * z, ((v - np.float32(23.5)) / np.float32(52.5)) * z,
                    np.full((H, W), z, np.float32)], axis=-1).reshape(-1, 3).astype(np.float32)
    ui, vi = u.astype(np.int64), v.astype(np.int64)
    cols = np.stack([(ui % 256).astype(np.float32) / np.float32(255.0), (vi % 256).astype(np.float32) / np.float32(255.0),
                     ((ui + vi) % 256).astype(np.float32) / np.float32(255.0)], axis=-1).reshape(-1, 3).astype(np.float32)
    cpose = np.eye(4, dtype=np.float32)
    cpose[:3, 3] = [0.05, 0.0, 0.02]
    om.integrate_pointcloud(pts, cpose, cols, 0.1, 5.0)
    V = N ** 3
    raw = np.fromfile(dump, np.uint8)
    rec = 12 + V * 8 + V * 4
    assert len(raw) % rec == 0 and len(raw) > 0
    got = {}
    for i in range(len(raw) // rec):
        b = raw[i * rec:(i + 1) * rec]
        cid = tuple(int(v) for v in b[:12].view(np.int32))
        sw = b[12:12 + V * 8].view(np.float32).reshape(V, 2)
        got[cid] = (sw[:, 0].copy(), sw[:, 1].copy(), b[12 + V * 8:].reshape(V, 4).copy())
    compare_fields(om.fields(), got, V, True, what="facade")
    assert om.num_chunks() == len(got)
    # the sdf query printed by the facade equals the oracle's
    ok, dist, grad = om.get_sdf_and_gradient((0.01, 0.01, 1.62))
    assert ok
    line = [l for l in out.stdout.splitlines() if l.startswith("sdf at")][0]
    vals = [float(x) for x in line.split("dist")[1].replace("grad", "").split()]
    assert np.float32(vals[0]) == np.float32(dist) and np.allclose(vals[1:], grad, rtol=0, atol=1e-7)


def _write_recording(path, frames, intr, color, encoding):
    """CVIDSRC1 recording (cvids_amd/open_chisel/tests/replay.cpp): what the topics of chisel_ros would carry"""
    import struct
    H, W = frames[0][0].shape
    ch = 0 if color is None else (1 if color.ndim == 2 else color.shape[2])
    with open(path, "wb") as f:
        f.write(b"CVIDSRC1" + struct.pack("<8i", len(frames), W, H, 1 if encoding == "16UC1" else 0, ch, 0, 0, 0))
        for depth, pose in frames:
            T = np.asarray(pose, np.float64)
            R, t = T[:3, :3], T[:3, 3]
            Ri, ti = R.T, -R.T @ t                       # the camera <- base transform tf would return
            # quaternion of Ri (x, y, z, w)
            w = np.sqrt(max(0.0, 1.0 + Ri[0, 0] + Ri[1, 1] + Ri[2, 2])) / 2.0
            q = np.array([(Ri[2, 1] - Ri[1, 2]) / (4 * w), (Ri[0, 2] - Ri[2, 0]) / (4 * w), (Ri[1, 0] - Ri[0, 1]) / (4 * w), w])
            f.write(np.asarray(intr, np.float64).tobytes() + ti.tobytes() + q.tobytes())
            if encoding == "16UC1":
                mm = np.where(np.isfinite(depth), np.round(depth.astype(np.float64) * 1000.0), 0).astype(np.uint16)
                f.write(mm.tobytes())
            else:
                f.write(np.ascontiguousarray(depth, np.float32).tobytes())
            if color is not None:
                f.write(np.ascontiguousarray(color).tobytes())


@pytest.mark.parametrize("encoding,use_color", [("32FC1", True), ("16UC1", True), ("32FC1", False)])
def test_replay_of_a_recorded_stream_matches_the_oracle(oracle_mod, tmp_path, encoding, use_color):
    """replay.cpp walks a 20-frame recording through ChiselServer's callback order on the C++ facade (camera info -> colour ->
    depth -> IntegrateDepthScan[Color] -> chunk boxes -> frustum -> UpdateMeshes -> meshes / pose, then the GetAllChunks /
    SaveMesh / Reset services).  The map it ends with -- read through the ChunkPtr mirrors GetChunks() hands out -- equals the
    oracle's on the same images (16UC1 millimetres converted as Conversions.h:140-150 does) and the poses the program derived."""
    tdir = os.path.join(ROOT, "cvids_amd", "open_chisel", "tests")
    subprocess.check_call(["make", "-C", tdir, "build"])
    W, H, N, res = 160, 120, 16, 0.04
    intr = synth.intrinsics(W, H)
    frames = list(synth.stream("sphere_room", 20, W, H, nan_fraction=0.01))
    color = synth.render_color(W, H, 3)
    rec = str(tmp_path / "stream.rec")
    _write_recording(rec, frames, intr, color, encoding)
    prefix = str(tmp_path / "out")
    out = subprocess.run([os.path.join(tdir, "replay"), rec, prefix, str(N), repr(res), "1" if use_color else "0", "0.05", "5.0", "2.0", "0.05"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.splitlines()
    per_frame = [l for l in lines if l.startswith("frame ")]
    assert len(per_frame) == 20 and all("frustum points 24" in l for l in per_frame)
    # meshes are recomputed on the 1st, 11th call (Chisel.cpp:53-58) and published while nothing is pending
    assert "published 0 times" not in per_frame[-1]
    poses = np.fromfile(prefix + ".poses", np.float32).reshape(20, 3, 4)
    om = oracle_mod.OracleMap(N, res, use_color)
    om.set_integrator(oracle_mod.TRUNC_INVERSE, 2.0, 1.0, True, 0.05)
    for (depth, pose), p in zip(frames, poses):
        assert np.abs(p - np.asarray(pose)[:3, :4]).max() < 1e-5  # the program's inverse of the recorded tf is the pose, to rounding
        if encoding == "16UC1":
            mm = np.where(np.isfinite(depth), np.round(depth.astype(np.float64) * 1000.0), 0).astype(np.uint16)
            d = (np.float32(1.0) / np.float32(1000.0)) * mm.astype(np.float32)
        else:
            d = depth
        P = np.eye(4, dtype=np.float32)
        P[:3, :4] = p
        if use_color:
            om.integrate_depth_color(d, P, intr, color, near=0.05, far=5.0)
        else:
            om.integrate_depth(d, P, intr, 0.05, 5.0)
    V = N ** 3
    raw = np.fromfile(prefix + ".map", np.uint8)
    recsz = 12 + V * 8 + (V * 4 if use_color else 0)
    assert len(raw) % recsz == 0 and len(raw) > 0
    got = {}
    for i in range(len(raw) // recsz):
        b = raw[i * recsz:(i + 1) * recsz]
        cid = tuple(int(v) for v in b[:12].view(np.int32))
        sw = b[12:12 + V * 8].view(np.float32).reshape(V, 2)
        got[cid] = (sw[:, 0].copy(), sw[:, 1].copy(), b[12 + V * 8:].reshape(V, 4).copy() if use_color else np.zeros((V, 4), np.uint8))
    compare_fields(om.fields(), got, V, use_color, what="replay")
    n = om.num_chunks()
    assert ("map: %d chunks written" % n) in out.stdout and "after Reset: 0 chunks" in out.stdout
    assert ("GetAllChunks %d messages" % n) in out.stdout and "SaveMesh ok" in out.stdout
    assert ("chunk boxes %d," % n) in per_frame[-1]
    assert os.path.getsize(prefix + ".ply") > 1000


def test_public_surface_beyond_chisel_ros():
    """facade_surface.cpp: the ChunkManager / ProjectionIntegrator / Frustum / Plane members a third-party caller of OpenChisel could
    use (ChunkManager.h:61-212, ProjectionIntegrator.h:51-52 / 101-102, Frustum.cpp:41-99), checked against each other on the GPU."""
    tdir = os.path.join(ROOT, "cvids_amd", "open_chisel", "tests")
    subprocess.check_call(["make", "-C", tdir, "build"])
    out = subprocess.run([os.path.join(tdir, "facade_surface")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "facade_surface ok" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    assert "Num Unknown:" in out.stdout and "Theoretical max (MB):" in out.stdout  # PrintMemoryStatistics' three lines


def test_shade_and_generate_against_the_oracle(oracle_mod):
    """chisel_hip_generate_mesh (stages 0 / 3) and chisel_hip_shade_vertices against the oracle's meshes: GenerateMesh's vertices are
    the stored mesh's, stages 3 reproduces the stored normals and colours, shading the bare vertices does too."""
    import ctypes as C
    from cvids_amd import chisel as ch
    from tests.common import make_frames, small_camera
    N, res, W, H = 16, 0.04, 96, 72
    cam = small_camera(W, H)
    intr = (cam.fx, cam.fy, cam.cx, cam.cy)
    color = synth.render_color(W, H, 3)
    om = oracle_mod.OracleMap(N, res, True)
    om.set_integrator(oracle_mod.TRUNC_INVERSE, 2.0, 1.0, True, 0.05)
    gm = ch.Chisel((N,) * 3, res, True, max_chunks=4096)
    integ = ch.ProjectionIntegrator(ch.InverseTruncator(2.0), ch.ConstantWeighter(1.0), 0.05, True)
    for d, p in make_frames("box_room", 3, W, H, nan_fraction=0.01):
        om.integrate_depth_color(d, p, intr, color, near=cam.near_plane, far=cam.far_plane)
        gm.IntegrateDepthScanColor(integ, d, p, cam, color, p, cam)
    om.update_meshes(force=True)
    todo = len(gm.GetMeshesToUpdate())
    assert todo > 0
    L = gm.L
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None
    checked = 0
    for cid in [tuple(int(v) for v in i) for i in om.mesh_ids()][:12]:
        want = om.get_mesh(cid)
        nv_want = len(want["vertices"])
        if nv_want == 0:
            continue
        cap = 15 * N ** 3
        cidc = (C.c_int * 3)(*cid)
        for stages in (0, 3):
            v, n, c, g = (np.zeros((cap, 3), np.float32) for _ in range(4))
            nv, ng = C.c_int64(0), C.c_int64(0)
            ch.check(L.chisel_hip_generate_mesh(gm.h, cidc, stages, cap, cap, fp(v), fp(n), fp(c), fp(g), C.byref(nv), C.byref(ng)))
            assert nv.value == nv_want and ng.value == len(want["grids"])
            assert np.array_equal(v[:nv.value].view(np.uint32), np.asarray(want["vertices"], np.float32).view(np.uint32))
            assert np.array_equal(g[:ng.value].view(np.uint32), np.asarray(want["grids"], np.float32).view(np.uint32))
            if stages == 3:
                assert np.array_equal(n[:nv.value].view(np.uint32), np.asarray(want["normals"], np.float32).view(np.uint32))
                assert np.array_equal(c[:nv.value].view(np.uint32), np.asarray(want["colors"], np.float32).view(np.uint32))
            else:
                face = n[:nv.value].copy()
        # shading the bare vertices: gradient normals over the face normals, colours
        vv = np.ascontiguousarray(want["vertices"], np.float32)
        nn = face.copy()
        cc = np.zeros_like(vv)
        ch.check(L.chisel_hip_shade_vertices(gm.h, fp(vv), len(vv), fp(nn), fp(cc), 3))
        assert np.array_equal(nn.view(np.uint32), np.asarray(want["normals"], np.float32).view(np.uint32))
        assert np.array_equal(cc.view(np.uint32), np.asarray(want["colors"], np.float32).view(np.uint32))
        checked += 1
    assert checked >= 5
    assert len(gm.GetMeshesToUpdate()) == todo and len(gm.GetMeshIDs()) == 0  # neither meshesToUpdate nor allMeshes were touched
    gm.close()
