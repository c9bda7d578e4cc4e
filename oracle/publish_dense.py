"""CPU restatement (numpy) of CollaborativeServer::PublishDenseInfo's depth conditioning -- TEST INFRASTRUCTURE ONLY.

Follows server_pose_graph/src/collaborative_server_system.cpp:199-276: cv::resize of the CV_64F depth map to the publish
size (:213; OpenCV's INTER_LINEAR for 64-bit floats: tap position (dx + 0.5) * scale - 0.5 narrowed to float, float
weights 1 - f and f, products and sums in double; beyond the last column the horizontal pass is S[sx] * 1; equal sizes
are copied), convertTo(CV_32FC1) (:255), NaN outside [0.1, 20] (:262-265), intrinsics rescale (:216-219).
PARITY UNPINNED: OpenCV is not installed here, so no vectors of the real cv::resize could be generated; this is the
published algorithm restated.  Only tests/ may import this module."""
import numpy as np


def _taps(n_dst, n_src):
    scale = np.float64(n_src) / np.float64(n_dst)
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0.0
    s[lo] = 0
    hi = s >= n_src - 1
    f[hi] = 0.0
    s[hi] = n_src - 1
    return s, np.minimum(s + 1, n_src - 1), (np.float32(1.0) - f).astype(np.float32), f, hi


def condition_depth(src, w, h):
    src = np.asarray(src, np.float64)
    h0, w0 = src.shape
    if (w, h) == (w0, h0):
        v = src.copy()
    else:
        x0, x1, a0, a1, xhi = _taps(w, w0)
        y0, y1, b0, b1, _ = _taps(h, h0)
        with np.errstate(invalid="ignore", over="ignore"):
            rows = src[:, x0] * a0.astype(np.float64) + src[:, x1] * a1.astype(np.float64)
            rows[:, xhi] = src[:, x0[xhi]] * 1.0
            v = rows[y0, :] * b0.astype(np.float64)[:, None] + rows[y1, :] * b1.astype(np.float64)[:, None]
    with np.errstate(invalid="ignore", over="ignore"):
        f = v.astype(np.float32)
        f[(f < np.float32(0.1)) | (f > np.float32(20.0))] = np.nan
    return f


def rescale_intrinsics(fx, fy, cx, cy, w0, h0, w, h):
    return (fx / float(w0) * float(w), fy / float(h0) * float(h), cx / float(w0) * float(w), cy / float(h0) * float(h))
