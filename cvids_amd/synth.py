"""Synthetic depth / colour streams for parity tests and bench.py.

Analytic scenes rendered on the host with numpy (float64 geometry, stored as float32 metres):
  S1 "wall"        plane z = 2.0 m in world coordinates
  S2 "sphere_room" camera inside a sphere of radius 2.5 m centred at the world origin
Pixel (col,row) sees the ray through (col+0.5, row+0.5): the centre of the cell that the
reference's truncating lookup `(int)u,(int)v` (ProjectionIntegrator.h:72,131) maps to that pixel.
Depth is z-depth along the optical axis (what `DepthImage` holds).  Poses are camera->world.
"""
import numpy as np

SEED = 20260102


def intrinsics(W=640, H=480):
    """fx, fy, cx, cy scaled from the 640x480 camera fx=fy=525, cx=319.5, cy=239.5."""
    s = W / 640.0
    return (525.0 * s, 525.0 * s, (W - 1) / 2.0, (H - 1) / 2.0)


def pose_yaw(theta_deg, t=(0.0, 0.0, 0.0)):
    th = np.deg2rad(theta_deg)
    c, s = np.cos(th), np.sin(th)
    T = np.eye(4, dtype=np.float64)
    T[:3, :3] = [[c, 0, s], [0, 1, 0], [-s, 0, c]]
    T[:3, 3] = t
    return T.astype(np.float32)


def trajectory_pose(k, agent=0):
    """frame k: yaw k*0.5 deg (+90 deg per agent), translation (0.01 k, 0, 0) m."""
    return pose_yaw(0.5 * k + 90.0 * agent, (0.01 * k, 0.0, 0.0))


def _rays(pose, intr, W, H):
    fx, fy, cx, cy = intr
    u = (np.arange(W, dtype=np.float64) + 0.5 - cx) / fx
    v = (np.arange(H, dtype=np.float64) + 0.5 - cy) / fy
    d_cam = np.stack(np.broadcast_arrays(u[None, :], v[:, None], np.ones((H, W))), axis=-1)
    R = np.asarray(pose, dtype=np.float64)[:3, :3]
    o = np.asarray(pose, dtype=np.float64)[:3, 3]
    return o, d_cam @ R.T


def render_depth(scene, pose, intr, W, H, noise=False, nan_fraction=0.0, frame_index=0):
    o, d = _rays(pose, intr, W, H)
    if scene == "wall":
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (2.0 - o[2]) / d[..., 2]
        t = np.where((d[..., 2] > 1e-9) & (t > 0), t, np.nan)
    elif scene == "sphere_room":
        a = (d * d).sum(-1)
        b = 2.0 * (d @ o)
        c = float(o @ o) - 2.5 ** 2
        disc = b * b - 4 * a * c
        t = (-b + np.sqrt(np.maximum(disc, 0.0))) / (2 * a)
        t = np.where(disc >= 0, t, np.nan)
    elif scene == "box_room":  # axis-aligned box |x|<=2, |y|<=1.5, |z|<=2.5 seen from inside
        half = np.array([2.0, 1.5, 2.5])
        with np.errstate(divide="ignore", invalid="ignore"):
            t1 = (half - o) / d
            t2 = (-half - o) / d
        t = np.minimum(np.where(d > 0, t1, np.inf), np.where(d < 0, t2, np.inf)).min(-1)
        t = np.where(np.isfinite(t), t, np.nan)
    else:
        raise ValueError(scene)
    depth = t.astype(np.float32)
    if noise or nan_fraction > 0:
        rng = np.random.default_rng(SEED + frame_index)
        if noise:
            depth = (depth + rng.normal(0.0, 1.0, depth.shape).astype(np.float32) * (0.002 * depth * depth)).astype(np.float32)
        if nan_fraction > 0:
            depth = np.where(rng.random(depth.shape) < nan_fraction, np.float32(np.nan), depth).astype(np.float32)
    return np.ascontiguousarray(depth)


def render_color(W, H, channels=3):
    """BGR8 pattern (u mod 256, v mod 256, (u+v) mod 256); mono = u mod 256; 4th channel 255."""
    u = np.arange(W, dtype=np.int32)[None, :].repeat(H, 0)
    v = np.arange(H, dtype=np.int32)[:, None].repeat(W, 1)
    if channels == 1:
        return (u % 256).astype(np.uint8)
    planes = [u % 256, v % 256, (u + v) % 256, np.full_like(u, 255)]
    return np.ascontiguousarray(np.stack(planes[:channels], axis=-1).astype(np.uint8))


def stream(scene, n_frames, W=640, H=480, agents=1, noise=False, nan_fraction=0.0, start=0):
    """Yield (depth, pose) in the global order a0f0, a1f0, ..., a0f1, ...."""
    intr = intrinsics(W, H)
    for k in range(start, start + n_frames):
        for a in range(agents):
            pose = trajectory_pose(k, a)
            yield render_depth(scene, pose, intr, W, H, noise, nan_fraction, frame_index=k * agents + a), pose


def depth_to_cloud(depth, intr, scale=1.0, colors=False):
    """Back-project the valid pixels of a depth image to sensor-frame points (n, 3) float32, in row-major pixel order
    (the shape of cloud the reference's ROSPointCloudToChisel hands to Chisel::IntegratePointCloud);
    `scale` shrinks the scene (the reference skips points deeper than 2 m, or 5 m with colours).  With `colors`,
    also returns (n, 3) float32 colours in [0, 1] taken from render_color()'s pattern."""
    fx, fy, cx, cy = intr
    H, W = depth.shape
    z = (depth.astype(np.float64) * scale)
    u = (np.arange(W, dtype=np.float64) + 0.5 - cx) / fx
    v = (np.arange(H, dtype=np.float64) + 0.5 - cy) / fy
    pts = np.stack([u[None, :] * z, v[:, None] * z, z], axis=-1).reshape(-1, 3)
    ok = np.isfinite(pts).all(-1)
    pts32 = np.ascontiguousarray(pts[ok].astype(np.float32))
    if not colors:
        return pts32
    bgr = render_color(W, H, 3).reshape(-1, 3)[ok]
    rgb = np.ascontiguousarray((bgr[:, ::-1].astype(np.float32) / np.float32(255.0)).astype(np.float32))
    return pts32, rgb
