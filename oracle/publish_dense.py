"""CPU restatement (numpy) of CollaborativeServer::PublishDenseInfo's image conditioning -- TEST INFRASTRUCTURE ONLY.

Follows server_pose_graph/src/collaborative_server_system.cpp:199-276: cv::resize of the CV_64F depth map and of the 8-bit colour
image to the publish size (:213-214), convertTo(CV_32FC1) (:255), NaN outside [0.1, 20] (:262-265), intrinsics rescale (:216-219),
and SendPointCloud (:318-381, called at :249): the data array of the organised PointCloud2 sent beside the images.

cv::resize is a third-party dependency that is absent from /root/reference (OpenCV, found by the reference's CMakeLists through
find_package(OpenCV); any 3.x / 4.x build without IPP); its published algorithm (modules/imgproc/src/resize.cpp) restated:
  * scale = 1. / ((double)dst / src) per axis (cv::resize computes the inverse scale first);
  * INTER_LINEAR taps: f = (float)((d + 0.5) * scale - 0.5); s = floor(f); f -= s.
    x: s < 0 -> s = 0, f = 0; s >= width - 1 -> s = width - 1, f = 0 and the horizontal pass there is S[s] * ONE (xmax);
    y: the fraction is kept; the two rows s, s + 1 are clipped to [0, height - 1];
  * CV_64F: float weights (1 - f, f), products and sums in double: S[s] * a0 + S[s + 1] * a1, then S0 * b0 + S1 * b1;
  * CV_8U: short weights saturate_cast<short>(w * 2048) (round half to even), int horizontal pass (ONE = 2048), vertical pass
    uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
  * both axes exactly halved: INTER_LINEAR is replaced by INTER_AREA -- CV_64F: (((a + b) + c) + d) * 0.25f over the 2 x 2 block in
    row-major order; CV_8U: (a + b + c + d + 2) >> 2;
  * equal sizes: copy.
PARITY UNPINNED: OpenCV is not installed here, so no vectors of the real cv::resize could be generated; hand-computed cases are in
tests/test_publish_dense.py.  Only tests/ may import this module."""
import numpy as np


def _scale(n_dst, n_src):
    return np.float64(1.0) / (np.float64(n_dst) / np.float64(n_src))


def _taps_x(n_dst, n_src):
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * _scale(n_dst, n_src) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0.0
    s[lo] = 0
    edge = s >= n_src - 1
    f[edge] = 0.0
    s[edge] = n_src - 1
    return s, np.minimum(s + 1, n_src - 1), f, edge


def _taps_y(n_dst, n_src):
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * _scale(n_dst, n_src) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return np.clip(s, 0, n_src - 1), np.clip(s + 1, 0, n_src - 1), f


def resize_f64(src, w, h):
    """cv::resize(CV_64FC1, Size(w, h)) with the default INTER_LINEAR"""
    src = np.asarray(src, np.float64)
    h0, w0 = src.shape
    if (w, h) == (w0, h0):
        return src.copy()
    with np.errstate(invalid="ignore", over="ignore"):
        if w0 == 2 * w and h0 == 2 * h:
            a, b, c, d = src[0::2, 0::2], src[0::2, 1::2], src[1::2, 0::2], src[1::2, 1::2]
            return (((a + b) + c) + d) * np.float64(np.float32(0.25))
        x0, x1, fx, edge = _taps_x(w, w0)
        y0, y1, fy = _taps_y(h, h0)
        one = np.float32(1.0)
        a0, a1 = (one - fx).astype(np.float64), fx.astype(np.float64)
        b0, b1 = (one - fy).astype(np.float64), fy.astype(np.float64)
        rows = src[:, x0] * a0 + src[:, x1] * a1
        rows[:, edge] = src[:, x0[edge]] * 1.0
        return rows[y0, :] * b0[:, None] + rows[y1, :] * b1[:, None]


def _coef(w):
    return np.rint(w.astype(np.float32) * np.float32(2048.0)).astype(np.int64)  # saturate_cast<short>: round half to even


def resize_u8(src, w, h):
    """cv::resize(CV_8UC1 / C3 / C4, Size(w, h)) with the default INTER_LINEAR; src (H, W) or (H, W, cn)"""
    src = np.asarray(src, np.uint8)
    flat = src.ndim == 2
    s = (src[:, :, None] if flat else src).astype(np.int64)
    h0, w0 = s.shape[:2]
    if (w, h) == (w0, h0):
        out = s
    elif w0 == 2 * w and h0 == 2 * h:
        out = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    else:
        x0, x1, fx, edge = _taps_x(w, w0)
        y0, y1, fy = _taps_y(h, h0)
        one = np.float32(1.0)
        a0, a1, b0, b1 = _coef(one - fx), _coef(fx), _coef(one - fy), _coef(fy)
        rows = s[:, x0] * a0[None, :, None] + s[:, x1] * a1[None, :, None]
        rows[:, edge] = s[:, x0[edge]] * 2048
        out = ((((b0[:, None, None] * (rows[y0] >> 4)) >> 16) + ((b1[:, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2) & 255
    out = out.astype(np.uint8)
    return out[:, :, 0] if flat else out


def condition_depth(src, w, h):
    v = resize_f64(src, w, h)
    with np.errstate(invalid="ignore", over="ignore"):
        f = v.astype(np.float32)
        f[(f < np.float32(0.1)) | (f > np.float32(20.0))] = np.nan
    return f


def condition_color(src, w, h):
    return resize_u8(src, w, h)


def publish_cloud(depth, color):
    """CollaborativeServer::SendPointCloud (collaborative_server_system.cpp:318-381): the PointCloud2 data array as uint32
    (H, W, 4): x = column, y = row, z = (float)depth as float bits, rgb = the grey byte at byte offset `column` of the colour image's
    row (mColorImage.at<uint8_t>(u, v) whatever the channel count) replicated; NaN bits in all four unless 0.1 < z < 10"""
    d = np.asarray(depth, np.float64)
    c = np.ascontiguousarray(color, np.uint8)
    h, w = d.shape
    rows = c.reshape(h, -1)
    with np.errstate(invalid="ignore", over="ignore"):
        dep = d.astype(np.float32)
        ok = (dep < np.float32(10.0)) & (dep > np.float32(0.1))
    g = rows[:, :w].astype(np.uint32)
    out = np.full((h, w, 4), 0x7fc00000, np.uint32)
    xs = np.broadcast_to(np.arange(w, dtype=np.float32)[None, :], (h, w)).copy().view(np.uint32)
    ys = np.broadcast_to(np.arange(h, dtype=np.float32)[:, None], (h, w)).copy().view(np.uint32)
    out[..., 0][ok] = xs[ok]
    out[..., 1][ok] = ys[ok]
    out[..., 2][ok] = dep.view(np.uint32)[ok]
    out[..., 3][ok] = ((g << 16) | (g << 8) | g)[ok]
    return out


def rescale_intrinsics(fx, fy, cx, cy, w0, h0, w, h):
    return (fx / float(w0) * float(w), fy / float(h0) * float(h), cx / float(w0) * float(w), cy / float(h0) * float(h))
