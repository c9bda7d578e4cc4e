"""oracle/publish_dense.py (the numpy restatement of cv::resize as PublishDenseInfo uses it) against cases computed by hand from
the rules its header lists.  CPU only.  OpenCV itself is absent from this image: parity with the real cv::resize is unpinned."""
import numpy as np

from oracle import publish_dense as pd


def test_same_size_is_a_copy():
    rng = np.random.default_rng(1)
    d = rng.uniform(0, 30, (5, 7))
    c = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    assert np.array_equal(pd.resize_f64(d, 7, 5), d)
    assert np.array_equal(pd.resize_u8(c, 7, 5), c)


def test_halving_is_the_block_mean():
    c = np.array([[1, 2, 255, 255], [3, 4, 255, 254]], np.uint8)
    assert pd.resize_u8(c, 2, 1).tolist() == [[3, 255]]  # (1+2+3+4+2)>>2, (1019+2)>>2
    d = np.array([[1.0, 2.0], [3.0, 4.5]])
    assert pd.resize_f64(d, 1, 1).tolist() == [[(((1.0 + 2.0) + 3.0) + 4.5) * 0.25]]


def test_upscale_row_by_two_u8():
    """4 -> 8 columns, rows unchanged: scale 0.5; d=0: f=-0.25 -> s=0, f=0; d=1: f=0.25 -> weights 1536 / 512; d=2: f=0.75;
    d=7: f=3.25 -> s=3 = width-1 -> S[3] * 2048.  Vertical pass with b = (2048, 0)."""
    src = np.array([[100, 200, 50, 8]], np.uint8)
    out = pd.resize_u8(src, 8, 1)

    def v(r):
        return (((2048 * (r >> 4)) >> 16) + 2) >> 2

    want = [v(100 * 2048), v(100 * 1536 + 200 * 512), v(100 * 512 + 200 * 1536), v(200 * 1536 + 50 * 512), v(200 * 512 + 50 * 1536),
            v(50 * 1536 + 8 * 512), v(50 * 512 + 8 * 1536), v(8 * 2048)]
    assert out.tolist() == [want]
    assert want[1] == 125 and want[0] == 100 and want[7] == 8


def test_upscale_rows_keep_their_fraction():
    """2 -> 4 rows: dy=0: f=-0.25 -> floor -1, fraction 0.75, rows clipped to (0, 0): b = (512, 1536) on the same row"""
    src = np.array([[100], [200]], np.uint8)
    out = pd.resize_u8(src, 1, 4)[:, 0].tolist()
    r0, r1 = (100 * 2048) >> 4, (200 * 2048) >> 4

    def v(b0, s0, b1, s1):
        return (((b0 * s0) >> 16) + ((b1 * s1) >> 16) + 2) >> 2

    assert out == [v(512, r0, 1536, r0), v(1536, r0, 512, r1), v(512, r0, 1536, r1), v(1536, r1, 512, r1)]
    d = pd.resize_f64(np.array([[1.0], [3.0]]), 1, 4)[:, 0].tolist()
    assert d == [1.0 * 0.25 + 1.0 * 0.75, 1.0 * 0.75 + 3.0 * 0.25, 1.0 * 0.25 + 3.0 * 0.75, 3.0 * 0.75 + 3.0 * 0.25]


def test_downscale_f64_taps():
    """1241 -> 640 columns: scale = 1 / (640 / 1241); tap of dx = 10 by hand"""
    w0 = 1241
    src = np.arange(w0, dtype=np.float64)[None, :].repeat(2, 0) * 0.01 + 1.0
    out = pd.resize_f64(src, 640, 2)
    scale = 1.0 / (640.0 / 1241.0)
    f = np.float32((10 + 0.5) * scale - 0.5)
    s = int(np.floor(f))
    fx = np.float32(f - np.float32(s))
    want = src[0, s] * np.float64(np.float32(1.0) - fx) + src[0, s + 1] * np.float64(fx)
    assert out[0, 10] == want * 1.0 + want * 0.0


def test_conditioning_masks_out_of_range():
    d = np.array([[0.05, 0.1, 20.0, 20.5, np.nan, np.inf, -1.0]])
    out = pd.condition_depth(d, 7, 1)
    assert np.isnan(out[0, [0, 3, 4, 5, 6]]).all() and out[0, 1] == np.float32(0.1) and out[0, 2] == np.float32(20.0)


def test_publish_cloud_by_hand():
    """SendPointCloud (collaborative_server_system.cpp:318-381): 2 x 3 map, BGR image: the grey byte of pixel (u, v) is byte v of row u"""
    depth = np.array([[0.5, 0.05, 9.99], [10.0, np.nan, 2.0]])
    color = np.arange(18, dtype=np.uint8).reshape(2, 3, 3) + 10  # row 0 bytes 10..18, row 1 bytes 19..27
    out = pd.publish_cloud(depth, color)
    f = lambda x: int(np.float32(x).view(np.uint32))
    nan = 0x7fc00000
    assert out[0, 0].tolist() == [f(0.0), f(0.0), f(0.5), (10 << 16) | (10 << 8) | 10]
    assert out[0, 1].tolist() == [nan] * 4
    assert out[0, 2].tolist() == [f(2.0), f(0.0), f(9.99), (12 << 16) | (12 << 8) | 12]
    assert out[1, 0].tolist() == [nan] * 4 and out[1, 1].tolist() == [nan] * 4
    assert out[1, 2].tolist() == [f(2.0), f(1.0), f(2.0), (21 << 16) | (21 << 8) | 21]
