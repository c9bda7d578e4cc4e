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
