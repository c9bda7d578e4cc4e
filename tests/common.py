"""Shared helpers of the parity tests: run the same frames through the oracle and the HIP path."""
import numpy as np

from cvids_amd import synth

DEFAULT_SDF = np.float32(99999.0)


def small_camera(W=64, H=48, near=0.05, far=5.0):
    from cvids_amd.chisel import PinholeCamera
    fx, fy, cx, cy = synth.intrinsics(W, H)
    return PinholeCamera(fx, fy, cx, cy, W, H, near, far)


def make_frames(scene, n, W, H, agents=1, nan_fraction=0.0, noise=False, start=0):
    return list(synth.stream(scene, n, W, H, agents=agents, nan_fraction=nan_fraction, noise=noise, start=start))


def compare_fields(ref, got, V, use_color, atol=0.0, what=""):
    """ref/got: dict id -> (sdf, w, rgbw).  Absent chunk == all (99999, 0, 0).  Returns max |dsdf|, |dw|."""
    ids = set(ref) | set(got)
    max_ds = 0.0
    max_dw = 0.0
    for cid in sorted(ids):
        rs, rw, rc = ref.get(cid, (None, None, None))
        gs, gw, gc = got.get(cid, (None, None, None))
        if rs is None:
            rs, rw = np.full(V, DEFAULT_SDF), np.zeros(V, np.float32)
            rc = np.zeros((V, 4), np.uint8)
        if gs is None:
            gs, gw = np.full(V, DEFAULT_SDF), np.zeros(V, np.float32)
            gc = np.zeros((V, 4), np.uint8)
        # NaN-aware comparison (the reference can produce NaN/inf, e.g. depth 0 with the inverse truncator)
        same_nan_s = np.isnan(rs) == np.isnan(gs)
        same_nan_w = np.isnan(rw) == np.isnan(gw)
        assert same_nan_s.all() and same_nan_w.all(), "%s chunk %s: NaN pattern differs" % (what, cid)
        fin = np.isfinite(rs) & np.isfinite(gs)
        assert (np.isfinite(rs) == np.isfinite(gs)).all(), "%s chunk %s: inf pattern differs" % (what, cid)
        ds = float(np.abs(rs[fin] - gs[fin]).max()) if fin.any() else 0.0
        finw = np.isfinite(rw) & np.isfinite(gw)
        dw = float(np.abs(rw[finw] - gw[finw]).max()) if finw.any() else 0.0
        max_ds, max_dw = max(max_ds, ds), max(max_dw, dw)
        if atol == 0.0:
            assert np.array_equal(rs[fin], gs[fin]), "%s chunk %s: sdf differs (max %g)" % (what, cid, ds)
            assert np.array_equal(rw[finw], gw[finw]), "%s chunk %s: weight differs (max %g)" % (what, cid, dw)
        else:
            assert ds <= atol and dw <= atol, "%s chunk %s: |dsdf| %g |dw| %g > %g" % (what, cid, ds, dw, atol)
        if use_color:
            assert np.array_equal(rc, gc), "%s chunk %s: colour voxels differ" % (what, cid)
    return max_ds, max_dw


def triangle_multiset(vertices, decimals=5):
    """mesh -> sorted array of triangles, each triangle's 3 vertices rotated to a canonical start."""
    v = np.asarray(vertices, np.float64).reshape(-1, 3, 3).round(decimals) + 0.0
    out = []
    for tri in v:
        keys = [tuple(p) for p in tri]
        k = keys.index(min(keys))
        out.append(tuple(keys[k:] + keys[:k]))
    return sorted(out)
