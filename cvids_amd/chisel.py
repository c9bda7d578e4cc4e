"""Host-side mirror of the reference's operator interface for the TSDF path, over the C ABI.

Names and argument meaning follow OpenChisel (open_chisel/include/open_chisel/*.h) so the parity
tests read like the reference's call sites (chisel_ros/src/ChiselServer.cpp:480-516):

    chisel = Chisel((16, 16, 16), 0.01, use_color=True)
    integrator = ProjectionIntegrator(InverseTruncator(1.0), ConstantWeighter(1), 0.05, True)
    chisel.IntegrateDepthScanColor(integrator, depth, pose, camera, color, pose, camera)
    chisel.UpdateMeshes()

Images may be numpy arrays (host, copied in by the library) or torch CUDA tensors (used in place).
Every call goes to libchisel_hip.so; nothing is computed in Python.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import ColorFrame, Config, DepthFrame, Integrator, check


# ---- strategy objects (truncation/*.h, weighting/ConstantWeighter.h) ---------------------------------
class ConstantTruncator:
    kind = capi.TRUNC_CONSTANT

    def __init__(self, value):
        self.param = float(value)


class InverseTruncator:
    kind = capi.TRUNC_INVERSE

    def __init__(self, scale):
        self.param = float(scale)


class QuadraticTruncator:
    kind = capi.TRUNC_QUADRATIC

    def __init__(self, scale):
        self.param = float(scale)


class ConstantWeighter:
    def __init__(self, weight):
        self.weight = float(weight)


class PinholeCamera:
    """camera/PinholeCamera.h:35-69 + Intrinsics.h:40-47"""

    def __init__(self, fx, fy, cx, cy, width, height, near_plane=0.05, far_plane=5.0):
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)
        self.width, self.height = int(width), int(height)
        self.near_plane, self.far_plane = float(near_plane), float(far_plane)


class ProjectionIntegrator:
    """ProjectionIntegrator.h:43-44, 185-217 (centroids are implicit: the kernel recomputes them)."""

    def __init__(self, truncator=None, weighter=None, carving_dist=0.05, enable_carving=True):
        self.truncator = truncator or InverseTruncator(8.0)
        self.weighter = weighter or ConstantWeighter(1.0)
        self.carving_dist = float(carving_dist)
        self.enable_carving = bool(enable_carving)

    def SetTruncator(self, t):
        self.truncator = t

    def SetWeighter(self, w):
        self.weighter = w

    def SetCarvingDist(self, d):
        self.carving_dist = float(d)

    def SetCarvingEnabled(self, e):
        self.enable_carving = bool(e)

    def _struct(self):
        return Integrator(self.truncator.kind, self.truncator.param, self.weighter.weight, int(self.enable_carving),
                          self.carving_dist)


def _image_pointer(img, np_dtype):
    """-> (address, on_device, keepalive) for a numpy array or a torch tensor."""
    if isinstance(img, np.ndarray):
        a = np.ascontiguousarray(img, dtype=np_dtype)
        return a.ctypes.data, 0, a
    # torch tensor
    t = img.contiguous()
    if t.is_cuda:
        return t.data_ptr(), 1, t
    a = np.ascontiguousarray(t.numpy(), dtype=np_dtype)
    return a.ctypes.data, 0, a


def _pose12(pose):
    p = np.ascontiguousarray(np.asarray(pose, dtype=np.float32)[:3, :4]).reshape(12)
    return (C.c_float * 12)(*p.tolist())


def depth_frame(depth, pose, camera):
    addr, dev, keep = _image_pointer(depth, np.float32)
    H, W = depth.shape[-2], depth.shape[-1]
    f = DepthFrame(addr, W, H, dev, _pose12(pose), camera.fx, camera.fy, camera.cx, camera.cy, camera.near_plane,
                   camera.far_plane)
    return f, keep


def color_frame(color, pose, camera):
    addr, dev, keep = _image_pointer(color, np.uint8)
    shp = tuple(color.shape)
    H, W = shp[0], shp[1]
    ch = 1 if len(shp) == 2 else shp[2]
    f = ColorFrame(addr, W, H, ch, dev, _pose12(pose), camera.fx, camera.fy, camera.cx, camera.cy)
    return f, keep


class PointCloud:
    """chisel::PointCloud (pointcloud/PointCloud.h:33-82): points in the sensor frame and, optionally, one colour per point."""

    def __init__(self, points=None, colors=None):
        self.points = np.zeros((0, 3), np.float32) if points is None else points
        self.colors = colors

    def HasColor(self):
        return self.colors is not None and len(self.colors) > 0

    def GetPoints(self):
        return self.points

    def GetColors(self):
        return self.colors

    def Clear(self):
        self.points = np.zeros((0, 3), np.float32)
        self.colors = None


class Chisel:
    """chisel::Chisel (Chisel.h:38-230) + the ChunkManager queries its callers use."""

    def __init__(self, chunk_size=(16, 16, 16), voxel_resolution=0.03, use_color=False, device_id=-1, max_chunks=0,
                 n_shards=1, shard_rank=0, shard_block=0, devices=None):
        """devices: a list of HIP device ordinals -> one map spread over these GPUs inside this process
        (chisel_hip_create_group: one shard per entry; an ordinal may repeat)."""
        self.L = capi.load_library()
        cs = (chunk_size,) * 3 if isinstance(chunk_size, int) else tuple(int(v) for v in chunk_size)
        self.chunk_size = cs
        self.V = cs[0] * cs[1] * cs[2]
        self.voxel_resolution = float(voxel_resolution)
        self.use_color = bool(use_color)
        cfg = Config((C.c_int * 3)(*cs), float(voxel_resolution), int(use_color), int(device_id), int(max_chunks),
                     int(n_shards), int(shard_rank), int(shard_block))
        self.h = C.c_void_p()
        if devices is not None:
            ids = (C.c_int * len(devices))(*[int(d) for d in devices])
            check(self.L.chisel_hip_create_group(C.byref(cfg), ids, len(devices), C.byref(self.h)))
        else:
            check(self.L.chisel_hip_create(C.byref(cfg), C.byref(self.h)))
        self._integrator = None
        self._keep = []

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.chisel_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- Chisel.h ---------------------------------------------------------------------------------
    def _use(self, integrator):
        s = integrator._struct()
        check(self.L.chisel_hip_set_integrator(self.h, C.byref(s)))

    def IntegrateDepthScan(self, integrator, depth_image, extrinsic, camera):
        self._use(integrator)
        f, keep = depth_frame(depth_image, extrinsic, camera)
        check(self.L.chisel_hip_integrate_depth(self.h, C.byref(f)))
        self._keep = [keep]

    def IntegrateDepthScanColor(self, integrator, depth_image, depth_extrinsic, depth_camera, color_image,
                                color_extrinsic, color_camera):
        self._use(integrator)
        f, k1 = depth_frame(depth_image, depth_extrinsic, depth_camera)
        c, k2 = color_frame(color_image, color_extrinsic, color_camera)
        check(self.L.chisel_hip_integrate_depth_color(self.h, C.byref(f), C.byref(c)))
        self._keep = [k1, k2]

    def IntegrateBatch(self, integrator, frames, colors=None):
        """frames: list of (depth, pose, camera); colors: list of (color, pose, camera) or None."""
        self._use(integrator)
        n = len(frames)
        fa = (DepthFrame * n)()
        keep = []
        for i, (d, p, cam) in enumerate(frames):
            fa[i], k = depth_frame(d, p, cam)
            keep.append(k)
        ca = None
        if colors is not None:
            ca = (ColorFrame * n)()
            for i, (c, p, cam) in enumerate(colors):
                ca[i], k = color_frame(c, p, cam)
                keep.append(k)
        check(self.L.chisel_hip_integrate_batch(self.h, n, fa, ca))
        self._keep = keep

    def IntegratePointCloud(self, integrator, cloud, extrinsic, truncation, max_dist):
        """Chisel::IntegratePointCloud (Chisel.h:57, Chisel.cpp:107-157).  `cloud`: a PointCloud, or a tuple (points, colors or None);
        points (n, 3) in the sensor frame, colours (n, 3) in [0, 1]; numpy arrays or torch CUDA tensors."""
        self._use(integrator)
        points, colors = (cloud.points, cloud.colors) if isinstance(cloud, PointCloud) else cloud
        pa, pdev, k1 = _image_pointer(points, np.float32)
        n = int(np.prod(tuple(points.shape))) // 3
        ca, cdev, k2 = (None, pdev, None)
        if colors is not None and int(np.prod(tuple(colors.shape))) > 0:  # PointCloud::HasColor
            ca, cdev, k2 = _image_pointer(colors, np.float32)
            assert int(np.prod(tuple(colors.shape))) // 3 == n, "one colour per point"
        assert cdev == pdev, "points and colours must live in the same memory space"
        pc = capi.PointCloud(pa, ca, n, pdev, _pose12(extrinsic), float(truncation), float(max_dist))
        check(self.L.chisel_hip_integrate_pointcloud(self.h, C.byref(pc)))
        self._keep = [k1, k2]

    def CloudCandidates(self, cloud, extrinsic, truncation, max_dist):
        """ChunkManager::GetChunkIDsIntersecting(cloud, cameraTransform, truncation, maxDist, chunkList) (ChunkManager.cpp:214-257): ids of
        the chunks the cloud's truncated rays pass through, ascending."""
        points, colors = (cloud.points, cloud.colors) if isinstance(cloud, PointCloud) else cloud
        pa, pdev, k1 = _image_pointer(points, np.float32)
        n = int(np.prod(tuple(points.shape))) // 3
        pc = capi.PointCloud(pa, None, n, pdev, _pose12(extrinsic), float(truncation), float(max_dist))
        cnt = C.c_int64(0)
        check(self.L.chisel_hip_cloud_candidates(self.h, C.byref(pc), None, 0, C.byref(cnt)))
        ids = np.zeros((max(1, cnt.value), 3), np.int32)
        check(self.L.chisel_hip_cloud_candidates(self.h, C.byref(pc), ids.ctypes.data_as(C.POINTER(C.c_int)), cnt.value, C.byref(cnt)))
        return ids[:cnt.value]

    def GarbageCollect(self, chunk_ids):
        ids = np.ascontiguousarray(np.asarray(chunk_ids, dtype=np.int32).reshape(-1, 3))
        check(self.L.chisel_hip_garbage_collect(self.h, ids.ctypes.data_as(C.POINTER(C.c_int)), len(ids)))

    def UpdateMeshes(self, force=False):
        check(self.L.chisel_hip_update_meshes(self.h, int(force)))

    def Reset(self):
        check(self.L.chisel_hip_reset(self.h))

    # ---- meshing a sharded map: halo chunks (chisel_hip.h "meshing a sharded map") ------------------------------
    def ExportChunks(self, ids, device=False):
        """-> (sdf [n, V] float32, weight [n, V] float32, rgbw [n, V, 4] uint8 or None, found [n] int32) of the listed chunks;
        device=True: the three payload arrays are torch CUDA tensors (they never visit the host), `found` stays a numpy array"""
        ids = np.ascontiguousarray(np.asarray(ids, np.int32).reshape(-1, 3))
        n = len(ids)
        found = np.zeros(n, np.int32)
        if device:
            import torch
            dev = torch.device("cuda", torch.cuda.current_device())
            sdf = torch.empty((n, self.V), dtype=torch.float32, device=dev)
            wgt = torch.empty((n, self.V), dtype=torch.float32, device=dev)
            col = torch.empty((n, self.V, 4), dtype=torch.uint8, device=dev) if self.use_color else None
            ptr = lambda t: t.data_ptr() if t is not None else None
        else:
            sdf = np.empty((n, self.V), np.float32)
            wgt = np.empty((n, self.V), np.float32)
            col = np.empty((n, self.V, 4), np.uint8) if self.use_color else None
            ptr = lambda a: a.ctypes.data if a is not None else None
        if n:
            check(self.L.chisel_hip_export_chunks(self.h, ids.ctypes.data_as(C.POINTER(C.c_int)), n, ptr(sdf), ptr(wgt), ptr(col),
                                                  found.ctypes.data_as(C.POINTER(C.c_int)), int(bool(device))))
        return sdf, wgt, col, found

    def ImportGhostChunks(self, ids, sdf, wgt, col=None, found=None):
        """payload: numpy arrays, or torch CUDA tensors (used in place)"""
        ids = np.ascontiguousarray(np.asarray(ids, np.int32).reshape(-1, 3))
        n = len(ids)
        if not n:
            return
        on_device = not isinstance(sdf, np.ndarray) and getattr(sdf, "is_cuda", False)
        if on_device:
            sdf, wgt = sdf.contiguous(), wgt.contiguous()
            col = col.contiguous() if col is not None else None
            ptr = lambda t: t.data_ptr() if t is not None else None
        else:
            sdf = np.ascontiguousarray(sdf, np.float32)
            wgt = np.ascontiguousarray(wgt, np.float32)
            col = np.ascontiguousarray(col, np.uint8) if col is not None else None
            ptr = lambda a: a.ctypes.data if a is not None else None
        found = np.ascontiguousarray(found, np.int32) if found is not None else None
        check(self.L.chisel_hip_import_ghost_chunks(self.h, ids.ctypes.data_as(C.POINTER(C.c_int)), n, ptr(sdf), ptr(wgt), ptr(col),
                                                    found.ctypes.data_as(C.POINTER(C.c_int)) if found is not None else None, int(on_device)))
        self._keep = [sdf, wgt, col]

    # ---- meshing a sharded map with shells (chisel_hip.h "meshing a sharded map with shells") --------------------
    def DirtyIdsDevice(self, out):
        """out: torch int32 CUDA tensor of 1 + 4 * capacity elements -> [n, (x, y, z, flag) * n] (filled on the map's stream: no wait)"""
        check(self.L.chisel_hip_dirty_ids_device(self.h, out.data_ptr(), (out.numel() - 1) // 4))

    def DirtyEntries(self):
        """the same list on the host: (n, 4) int32"""
        import torch
        cap = 1 << 14
        while True:
            buf = torch.zeros((1 + 4 * cap,), dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
            torch.cuda.current_stream().synchronize()  # the buffer is zero before the map's stream writes into it
            self.DirtyIdsDevice(buf)
            self.synchronize()
            h = buf.cpu().numpy()
            if h[0] <= cap:
                return h[1:1 + 4 * int(h[0])].reshape(-1, 4).copy()
            cap = 2 * int(h[0])

    def ExportShells(self, items, device=False):
        """items: (n, 4) int32 (x, y, z, box) -> (sdf [vox], weight [vox], rgbw [vox, 4] or None, found [n]) packed box after box;
        device=True: torch CUDA tensors, nothing has been waited for (record_event orders the consumer)"""
        items = np.ascontiguousarray(np.asarray(items, np.int32).reshape(-1, 4))
        n = len(items)
        edge = self.chunk_size[0]
        vox = int(shell_volumes(items[:, 3], edge).sum()) if n else 0
        if device:
            import torch
            dev = torch.device("cuda", torch.cuda.current_device())
            sdf = torch.empty((vox,), dtype=torch.float32, device=dev)
            wgt = torch.empty((vox,), dtype=torch.float32, device=dev)
            col = torch.empty((vox, 4), dtype=torch.uint8, device=dev) if self.use_color else None
            found = torch.empty((n,), dtype=torch.int32, device=dev)  # (every entry is written by the kernel)
            ptr = lambda t: t.data_ptr() if t is not None else None
        else:
            sdf, wgt = np.empty(vox, np.float32), np.empty(vox, np.float32)
            col = np.empty((vox, 4), np.uint8) if self.use_color else None
            found = np.zeros(n, np.int32)
            ptr = lambda a: a.ctypes.data if a is not None else None
        if n:
            check(self.L.chisel_hip_export_shells(self.h, items.ctypes.data_as(C.POINTER(C.c_int)), n, ptr(sdf), ptr(wgt), ptr(col), ptr(found),
                                                  int(bool(device))))
        self._keep = [items]
        return sdf, wgt, col, found

    def ImportGhostShells(self, items, sdf, wgt, col, found):
        """payload as ExportShells returns it (numpy arrays, or torch CUDA tensors used in place behind wait_event)"""
        items = np.ascontiguousarray(np.asarray(items, np.int32).reshape(-1, 4))
        n = len(items)
        if not n:
            return
        on_device = not isinstance(sdf, np.ndarray) and getattr(sdf, "is_cuda", False)
        if on_device:
            ptr = lambda t: t.data_ptr() if t is not None else None
        else:
            sdf, wgt = np.ascontiguousarray(sdf, np.float32), np.ascontiguousarray(wgt, np.float32)
            col = np.ascontiguousarray(col, np.uint8) if col is not None else None
            found = np.ascontiguousarray(found, np.int32)
            ptr = lambda a: a.ctypes.data if a is not None else None
        check(self.L.chisel_hip_import_ghost_shells(self.h, items.ctypes.data_as(C.POINTER(C.c_int)), n, ptr(sdf), ptr(wgt), ptr(col), ptr(found),
                                                    int(on_device)))
        # the import is only queued (device payloads are read in place, nothing is waited for): the payload must outlive it.  Kept until
        # DropGhostChunks, which returns after the recompute behind the imports has started (it looks at that recompute's totals).
        self._ghost_keep = getattr(self, "_ghost_keep", []) + [(items, sdf, wgt, col, found)]
        self._imports_fenced = False  # (only a recompute queued BEHIND this import tells the host that it has read its payload)

    # ---- the sharded recompute planned on the device (chisel_hip.h: chisel_hip_shell_plan_device ...) -------------------------------
    def PlanShellsDevice(self, gathered, world, cap):
        """gathered: int32 CUDA tensor, per rank 1 + 4 * cap ints (count, then (x, y, z, flag) entries).  -> dict: jobs (of this shard),
        max_count (largest per-rank count: > cap means entries were cut off and nothing else counts), send / recv: (world, 2) int64 arrays
        of (items, voxels) per peer.  Waits for those figures: the one host wait of a sharded recompute."""
        out = np.zeros(4 + 4 * world, np.int64)
        check(self.L.chisel_hip_shell_plan_device(self.h, gathered.data_ptr(), int(world), int(cap), out.ctypes.data_as(C.POINTER(C.c_int64))))
        self._plan_keep = gathered  # (read by the plan kernels, which the call has waited for; kept for symmetry with the buffers below)
        self._packed_keep = []      # the previous recompute's buffers: its drop kernel ran before the wait above returned
        return {"jobs": int(out[0]), "ghosts_before": int(out[1]), "max_count": int(out[2]), "send_items": int(out[3]),
                "send": out[4:4 + 2 * world].reshape(world, 2).copy(), "recv": out[4 + 2 * world:4 + 4 * world].reshape(world, 2).copy()}

    def ShellSegmentBytes(self, items, voxels):
        return int(self.L.chisel_hip_shell_segment_bytes(self.h, int(items), int(voxels)))

    def ExportShellsPacked(self, out):
        """out: uint8 CUDA tensor of the plan's send size; nothing is waited for (record_event orders the collective)"""
        check(self.L.chisel_hip_export_shells_packed(self.h, out.data_ptr() if out.numel() else None, int(out.numel())))
        self._packed_keep = getattr(self, "_packed_keep", []) + [out]

    def ImportShellsPacked(self, buf):
        """buf: uint8 CUDA tensor holding the received segments (read in place behind wait_event; kept until the next plan)"""
        check(self.L.chisel_hip_import_shells_packed(self.h, buf.data_ptr() if buf.numel() else None, int(buf.numel())))
        self._packed_keep = getattr(self, "_packed_keep", []) + [buf]

    # ---- ... and its wait-free form (chisel_hip.h: chisel_hip_shell_plan_queue ...): every tensor below stays the caller's until ShellCommit
    def PlanShellsQueue(self, gathered, world, cap, stride, status, out, send_items_hint=0):
        """queues the plan, the export of `world` segments of `stride` bytes into `out` (uint8 CUDA tensor) and this rank's status (int32 CUDA
        tensor of SHELL_STATUS_INTS entries); nothing is waited for"""
        check(self.L.chisel_hip_shell_plan_queue(self.h, gathered.data_ptr(), int(world), int(cap), int(stride), status.data_ptr(), out.data_ptr(), int(send_items_hint)))

    def ImportShellsFixed(self, buf, stride, status, jobs_hint=0, items_hint=0):
        """buf: the received segments; status: the ALL-REDUCED status vector (word 0 != 0: nothing below happens on the device)"""
        check(self.L.chisel_hip_import_shells_fixed(self.h, buf.data_ptr(), int(stride), status.data_ptr(), int(jobs_hint), int(items_hint)))

    def ShellCommit(self, aborted):
        check(self.L.chisel_hip_shell_commit(self.h, int(bool(aborted))))

    SHELL_STATUS_INTS = 8

    def UpdateMeshesPlanned(self):
        check(self.L.chisel_hip_update_meshes_planned(self.h))
        self._imports_fenced = True

    def DropGhostChunks(self):
        check(self.L.chisel_hip_drop_ghost_chunks(self.h))
        if getattr(self, "_ghost_keep", None):
            # a recompute queued behind the imports (UpdateMeshesOf): the call above has looked at its totals, which the device publishes
            # after everything in front of it -- the imports have read their payload.  Without one nothing has told the host so: wait.
            if not getattr(self, "_imports_fenced", False):
                self.synchronize()
            self._ghost_keep = []
        self._imports_fenced = False

    def UpdateMeshesOf(self, ids):
        ids = np.ascontiguousarray(np.asarray(ids, np.int32).reshape(-1, 3))
        check(self.L.chisel_hip_update_meshes_of(self.h, ids.ctypes.data_as(C.POINTER(C.c_int)), len(ids)))
        self._imports_fenced = True

    def SaveMap(self, filename):
        """binary dump of every resident chunk (chisel_hip_save_map): checkpoint"""
        check(self.L.chisel_hip_save_map(self.h, str(filename).encode()))

    def LoadMap(self, filename):
        """replace the map's contents by a dump written by SaveMap: resume"""
        check(self.L.chisel_hip_load_map(self.h, str(filename).encode()))

    def SaveAllMeshesToPLY(self, filename):
        rc = self.L.chisel_hip_save_ply(self.h, str(filename).encode())
        if rc == 6:
            return False
        check(rc)
        return True

    def GetMeshesToUpdate(self):
        n = C.c_int64(0)
        check(self.L.chisel_hip_meshes_to_update(self.h, None, 0, C.byref(n)))
        ids = np.zeros((n.value, 3), np.int32)
        if n.value:
            check(self.L.chisel_hip_meshes_to_update(self.h, ids.ctypes.data_as(C.POINTER(C.c_int)), n.value, C.byref(n)))
        return ids

    def PrefetchMeshesToUpdate(self, cursor):
        """chisel_hip_meshes_to_update_prefetch: queue the listing behind the integration just issued (no wait)"""
        check(self.L.chisel_hip_meshes_to_update_prefetch(self.h, cursor))

    def GetMeshesToUpdateSince(self, cursor, capacity=8192):
        """chisel_hip_meshes_to_update_since: (ids that joined the set since `cursor` [n, 3], cleared) -- `cursor` is a (C.c_uint64 * 2)
        the caller keeps (zero before the first call); what the C++ facade's GetMeshesToUpdate is built on"""
        n, cleared = C.c_int64(0), C.c_int(0)
        while True:
            ids = np.zeros((capacity, 3), np.int32)
            check(self.L.chisel_hip_meshes_to_update_since(self.h, cursor, ids.ctypes.data_as(C.POINTER(C.c_int)), capacity, C.byref(n), C.byref(cleared)))
            if n.value <= capacity:
                return ids[:n.value], bool(cleared.value)
            capacity = n.value + 64

    # ---- ChunkManager.h -----------------------------------------------------------------------------
    def synchronize(self):
        check(self.L.chisel_hip_synchronize(self.h))

    def set_stream(self, hip_stream):
        check(self.L.chisel_hip_set_stream(self.h, C.c_void_p(hip_stream)))

    def wait_event(self, hip_event):
        """the device frames of the next Integrate* call are complete behind this hipEvent_t (e.g. torch.cuda.Event.cuda_event)"""
        check(self.L.chisel_hip_wait_event(self.h, C.c_void_p(hip_event)))

    def record_event(self, hip_event):
        """record the hipEvent_t behind everything queued on the map: after it the frames of earlier calls have been read"""
        check(self.L.chisel_hip_record_event(self.h, C.c_void_p(hip_event)))

    def order_stream_after_map(self, hip_stream):
        """whatever the hipStream_t is given next starts after what the map has queued so far (one call, the map's own event)"""
        check(self.L.chisel_hip_order_stream_after_map(self.h, C.c_void_p(hip_stream)))

    def order_map_after_stream(self, hip_stream):
        """the map's next call starts after what the hipStream_t has been given so far"""
        check(self.L.chisel_hip_order_map_after_stream(self.h, C.c_void_p(hip_stream)))

    def NumChunks(self):
        n = C.c_int64(0)
        check(self.L.chisel_hip_num_chunks(self.h, C.byref(n)))
        return n.value

    def GetChunkIDs(self):
        n = C.c_int64(0)
        check(self.L.chisel_hip_list_chunks(self.h, None, 0, C.byref(n)))
        ids = np.zeros((n.value, 3), np.int32)
        if n.value:
            check(self.L.chisel_hip_list_chunks(self.h, ids.ctypes.data_as(C.POINTER(C.c_int)), n.value, C.byref(n)))
        return ids

    def HasChunk(self, cid):
        cid = (C.c_int * 3)(*[int(v) for v in cid])
        out = C.c_int(0)
        check(self.L.chisel_hip_has_chunk(self.h, cid, C.byref(out)))
        return bool(out.value)

    def GetChunk(self, cid):
        """-> (sdf[V], weight[V], rgbw[V,4] or None); raises KeyError like ChunkMap::at."""
        cid_c = (C.c_int * 3)(*[int(v) for v in cid])
        sdf = np.empty(self.V, np.float32)
        w = np.empty(self.V, np.float32)
        rgbw = np.empty((self.V, 4), np.uint8) if self.use_color else None
        rc = self.L.chisel_hip_download_chunk(self.h, cid_c, sdf.ctypes.data_as(C.POINTER(C.c_float)),
                                              w.ctypes.data_as(C.POINTER(C.c_float)),
                                              rgbw.ctypes.data_as(C.POINTER(C.c_uint8)) if rgbw is not None else None)
        if rc == 4:
            raise KeyError(tuple(int(v) for v in cid))
        check(rc)
        return sdf, w, rgbw

    def AddChunk(self, cid, sdf, weight, rgbw=None):
        cid_c = (C.c_int * 3)(*[int(v) for v in cid])
        s = np.ascontiguousarray(sdf, np.float32)
        w = np.ascontiguousarray(weight, np.float32)
        c = np.ascontiguousarray(rgbw, np.uint8) if rgbw is not None else None
        check(self.L.chisel_hip_upload_chunk(self.h, cid_c, s.ctypes.data_as(C.POINTER(C.c_float)),
                                             w.ctypes.data_as(C.POINTER(C.c_float)),
                                             c.ctypes.data_as(C.POINTER(C.c_uint8)) if c is not None else None))

    def fields(self):
        return {tuple(int(v) for v in cid): self.GetChunk(cid) for cid in self.GetChunkIDs()}

    def GetMeshIDs(self):
        n = C.c_int64(0)
        check(self.L.chisel_hip_list_meshes(self.h, None, 0, C.byref(n)))
        ids = np.zeros((n.value, 3), np.int32)
        if n.value:
            check(self.L.chisel_hip_list_meshes(self.h, ids.ctypes.data_as(C.POINTER(C.c_int)), n.value, C.byref(n)))
        return ids

    def GetMesh(self, cid):
        cid_c = (C.c_int * 3)(*[int(v) for v in cid])
        nv, ng = C.c_int64(0), C.c_int64(0)
        rc = self.L.chisel_hip_mesh_size(self.h, cid_c, C.byref(nv), C.byref(ng))
        if rc == 4:
            raise KeyError(tuple(int(v) for v in cid))
        check(rc)
        v = np.zeros((nv.value, 3), np.float32)
        n = np.zeros((nv.value, 3), np.float32)
        c = np.zeros((nv.value, 3), np.float32) if self.use_color else None
        g = np.zeros((ng.value, 3), np.float32)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None
        check(self.L.chisel_hip_download_mesh(self.h, cid_c, fp(v), fp(n), fp(c), fp(g)))
        return {"vertices": v, "normals": n, "colors": c, "grids": g}

    def GetSDF(self, pos):
        p = (C.c_float * 3)(*[float(v) for v in pos])
        d, found = C.c_double(0), C.c_int(0)
        check(self.L.chisel_hip_get_sdf(self.h, p, C.byref(d), C.byref(found)))
        return bool(found.value), d.value

    def GetSDFAndGradient(self, pos):
        p = (C.c_float * 3)(*[float(v) for v in pos])
        g = (C.c_float * 3)()
        d, found = C.c_double(0), C.c_int(0)
        check(self.L.chisel_hip_get_sdf_and_gradient(self.h, p, C.byref(d), g, C.byref(found)))
        return bool(found.value), d.value, np.array(list(g), np.float32)

    def MemoryStatistics(self):
        """ChunkManager::PrintMemoryStatistics (ChunkManager.cpp:641-678) as numbers: the voxel census of Chunk::ComputeStatistics over
        the resident chunks, the weight sum, the bounds of the chunk boxes and the two memory figures the reference prints (it
        prices a voxel at sizeof(DistVoxel) = 16 bytes)."""
        st = capi.Statistics()
        check(self.L.chisel_hip_memory_statistics(self.h, C.byref(st)))
        n, res = self.chunk_size, np.float32(self.voxel_resolution)
        out = {"numUnknown": st.n_unknown, "numKnownInside": st.n_known_inside, "numKnownOutside": st.n_known_outside,
               "totalWeight": st.total_weight, "chunks": st.n_chunks}
        if st.n_chunks:
            lo = np.array([np.float32(n[a] * st.id_min[a]) * res for a in range(3)], np.float32)                       # Chunk.cpp:43
            hi = np.array([np.float32(n[a] * st.id_max[a]) * res + np.float32(n[a]) * res for a in range(3)], np.float32)  # Chunk.cpp:65-70
            out["bounds"] = (lo, hi)
            ext = (hi - lo) * np.float32(0.5)                                                                         # AABB::GetExtents
            nv = ext * np.float32(2) / res
            out["max_memory_mb"] = float(nv[0] * nv[1] * nv[2] * np.float32(16) / np.float32(1000000.0))
        out["current_memory_mb"] = float(np.float32(st.n_chunks * n[0] * n[1] * n[2] * 16) / np.float32(1000000.0))
        return out

    # ---- measurement ----------------------------------------------------------------------------------
    def counters(self, reset=False):
        out = (C.c_uint64 * capi.NUM_COUNTERS)()
        check(self.L.chisel_hip_get_counters(self.h, out, int(reset)))
        return dict(zip(capi.COUNTER_NAMES, [int(v) for v in out]))

    def set_profiling(self, enable):
        check(self.L.chisel_hip_set_profiling(self.h, int(enable)))

    def profile(self, reset=False):
        ms = (C.c_double * capi.NUM_KERNELS)()
        n = (C.c_int64 * capi.NUM_KERNELS)()
        check(self.L.chisel_hip_get_profile(self.h, ms, n, int(reset)))
        return {k: {"ms": ms[i], "launches": int(n[i])} for i, k in enumerate(capi.KERNEL_NAMES)}

    LAUNCH_STATS = ("integrate_2_per_lane", "integrate_4_per_lane", "integrate_4_with_2_tail", "cull_4_waves", "cull_wave_per_frame",
                    "unordered_worklists", "single_stream_sets", "launch_sets", "behind_unseen_recompute", "replayed")

    def pool_info(self):
        """chisel_hip_pool_info: chunks committed now, the pool's limit, times it has grown, whether it can"""
        out = (C.c_int64 * 4)()
        check(self.L.chisel_hip_pool_info(self.h, out))
        return {"committed": int(out[0]), "limit": int(out[1]), "grown": int(out[2]), "growable": bool(out[3])}

    def launch_stats(self, reset=False):
        """which shapes the launch heuristics picked (chisel_hip_get_launch_stats)"""
        out = (C.c_int64 * len(self.LAUNCH_STATS))()
        check(self.L.chisel_hip_get_launch_stats(self.h, out, int(reset)))
        return dict(zip(self.LAUNCH_STATS, [int(v) for v in out]))


class DepthFilter:
    """DepthFilter of the dense-mapping thread (server_pose_graph/src/dense_mapping/depth_filter.cpp) with its state in HBM."""
    A, B, INV_DEPTH, COV, RATIO, INV_DEPTH_MASKED, DEPTH = range(7)

    def __init__(self, height, width, device_id=-1):
        self.L = capi.load_library()
        self.shape = (int(height), int(width))
        self.h = C.c_void_p()
        check(self.L.chisel_hip_depth_filter_create(int(height), int(width), int(device_id), C.byref(self.h)))
        self._keep = None

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.chisel_hip_depth_filter_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def Update(self, update_mu, update_cov, reciprocal=False):
        """DepthFilter::Update(mUpdateMu, mUpdateCov): float64 maps (numpy or torch CUDA tensors); update_cov may be a scalar."""
        mu_addr, dev, k1 = _image_pointer(update_mu, np.float64)
        assert tuple(update_mu.shape) == self.shape
        if np.isscalar(update_cov):
            check(self.L.chisel_hip_depth_filter_update(self.h, mu_addr, None, float(update_cov), int(reciprocal), dev))
            self._keep = [k1]
        else:
            cov_addr, dev2, k2 = _image_pointer(update_cov, np.float64)
            assert dev2 == dev and tuple(update_cov.shape) == self.shape
            check(self.L.chisel_hip_depth_filter_update(self.h, mu_addr, cov_addr, 0.0, int(reciprocal), dev))
            self._keep = [k1, k2]

    def read(self, which, out=None):
        """-> float64 map; `out`: a torch CUDA tensor to fill in place (stays in HBM), default a new numpy array"""
        if out is not None:
            check(self.L.chisel_hip_depth_filter_read(self.h, int(which), out.data_ptr(), 1))
            return out
        a = np.empty(self.shape, np.float64)
        check(self.L.chisel_hip_depth_filter_read(self.h, int(which), a.ctypes.data, 0))
        return a

    def GetA(self):
        return self.read(self.A)

    def GetB(self):
        return self.read(self.B)

    def GetInvDepth(self):
        return self.read(self.INV_DEPTH)

    def GetCov(self):
        return self.read(self.COV)

    def GetRatio(self):
        return self.read(self.RATIO)


def condition_depth(depth64, width=640, height=480, intrinsics=None):
    """CollaborativeServer::PublishDenseInfo's depth conditioning (chisel_hip_condition_depth): float64 depth map of any size
    -> (float32 depth of the publish size with NaN outside [0.1, 20] m, rescaled (fx, fy, cx, cy) or None)"""
    L = capi.load_library()
    src = np.ascontiguousarray(depth64, np.float64)
    h0, w0 = src.shape
    dst = np.empty((height, width), np.float32)
    K = (C.c_double * 4)(*(intrinsics if intrinsics is not None else (0.0, 0.0, 0.0, 0.0)))
    check(L.chisel_hip_condition_depth(src.ctypes.data, w0, h0, 0, dst.ctypes.data, width, height, 0, K, None))
    return dst, (tuple(K) if intrinsics is not None else None)


def condition_color(image, width=640, height=480):
    """PublishDenseInfo's cv::resize of the colour image (chisel_hip_condition_color): uint8 (H, W) or (H, W, 1 / 3 / 4) ->
    uint8 of the publish size, same channel count"""
    L = capi.load_library()
    src = np.ascontiguousarray(image, np.uint8)
    cn = 1 if src.ndim == 2 else src.shape[2]
    h0, w0 = src.shape[:2]
    dst = np.empty((height, width) if src.ndim == 2 else (height, width, cn), np.uint8)
    check(L.chisel_hip_condition_color(src.ctypes.data, w0, h0, cn, 0, dst.ctypes.data, width, height, 0, None))
    return dst


def publish_cloud(depth64, color):
    """CollaborativeServer::SendPointCloud (chisel_hip_publish_cloud): float64 depth (H, W) + uint8 colour image (H, W[, C]) ->
    the PointCloud2 data array as uint32 (H, W, 4): x, y, z float bits and the packed grey rgb"""
    L = capi.load_library()
    d = np.ascontiguousarray(depth64, np.float64)
    c = np.ascontiguousarray(color, np.uint8)
    h, w = d.shape
    step = c.size // h
    out = np.empty((h, w, 4), np.uint32)
    check(L.chisel_hip_publish_cloud(d.ctypes.data, c.ctypes.data, w, h, step, 0, out.ctypes.data, 0, None))
    return out


def chunk_owner(cid, n_shards, shard_block=2):
    c = (C.c_int * 3)(*[int(v) for v in cid])
    return capi.load_library().chisel_hip_chunk_owner(c, int(n_shards), int(shard_block))


def mesh_shell_plan(entries, n_shards, rank, shard_block=2):
    """chisel_hip_mesh_shell_plan: entries (n, 4) int32 (x, y, z, flag) -> (jobs (nj, 3), items (ni, 5): owner, x, y, z, box)"""
    L = capi.load_library()
    e = np.ascontiguousarray(np.asarray(entries, np.int32).reshape(-1, 4))
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int)) if a is not None else None
    nj, ni = C.c_int64(0), C.c_int64(0)
    check(L.chisel_hip_mesh_shell_plan(ip(e), len(e), int(n_shards), int(rank), int(shard_block), None, 0, C.byref(nj), None, 0, C.byref(ni)))
    jobs, items = np.zeros((nj.value, 3), np.int32), np.zeros((ni.value, 5), np.int32)
    check(L.chisel_hip_mesh_shell_plan(ip(e), len(e), int(n_shards), int(rank), int(shard_block), ip(jobs), nj.value, C.byref(nj), ip(items), ni.value,
                                       C.byref(ni)))
    return jobs, items


def shell_volume(box, chunk_edge):
    return int(capi.load_library().chisel_hip_shell_volume(int(box), int(chunk_edge)))


_SHELL_VOLUMES = {}


def shell_volumes(boxes, chunk_edge):
    """voxels in the payload of every box code of an array (a table of the 64 codes per chunk edge: one library call per code and
    edge, not one per item -- a recompute lists thousands of items)"""
    table = _SHELL_VOLUMES.get(int(chunk_edge))
    if table is None:
        table = _SHELL_VOLUMES[int(chunk_edge)] = np.array([shell_volume(b, chunk_edge) for b in range(64)], np.int64)
    return table[np.asarray(boxes, np.int64)]
