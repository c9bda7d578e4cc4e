"""ctypes bindings of include/chisel_hip.h (the drop-in C ABI)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# CHISEL_HIP_LIB selects a diagnostic build (e.g. libchisel_hip_stamps.so); the default is the product library
_LIB = os.path.join(_HERE, os.environ.get("CHISEL_HIP_LIB", "libchisel_hip.so"))

ABI_VERSION = 2  # CHISEL_HIP_ABI_VERSION of include/chisel_hip.h this mirror was written against (tests/test_abi.py compares)
NUM_COUNTERS = 9
COUNTER_NAMES = ["sdf", "col", "col_sat", "probe", "carved", "work_chunks", "new_chunks", "updated_chunks", "frames"]
NUM_KERNELS = 6
KERNEL_NAMES = ["pyramid", "cull", "integrate", "mesh", "resolve", "cloud"]
TRUNC_CONSTANT, TRUNC_INVERSE, TRUNC_QUADRATIC = 0, 1, 2
STATUS = {0: "OK", 1: "ERR_INVALID", 2: "ERR_HIP", 3: "ERR_POOL_FULL", 4: "ERR_NOT_FOUND", 5: "ERR_UNSUPPORTED", 6: "ERR_IO"}


class ChiselHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("chisel_hip: %s (%d): %s" % (STATUS.get(code, "?"), code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("chunk_size", C.c_int * 3), ("voxel_resolution", C.c_float), ("use_color", C.c_int),
                ("device_id", C.c_int), ("max_chunks", C.c_int64), ("n_shards", C.c_int), ("shard_rank", C.c_int),
                ("shard_block", C.c_int)]


class Integrator(C.Structure):
    _fields_ = [("truncator_kind", C.c_int), ("truncator_param", C.c_float), ("weight", C.c_float),
                ("carving_enabled", C.c_int), ("carving_dist", C.c_float)]


class DepthFrame(C.Structure):
    _fields_ = [("depth", C.c_void_p), ("width", C.c_int), ("height", C.c_int), ("on_device", C.c_int),
                ("pose", C.c_float * 12), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("near_plane", C.c_float), ("far_plane", C.c_float)]


class ColorFrame(C.Structure):
    _fields_ = [("color", C.c_void_p), ("width", C.c_int), ("height", C.c_int), ("channels", C.c_int),
                ("on_device", C.c_int), ("pose", C.c_float * 12), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float)]


class PointCloud(C.Structure):
    _fields_ = [("points", C.c_void_p), ("colors", C.c_void_p), ("n_points", C.c_int64), ("on_device", C.c_int),
                ("pose", C.c_float * 12), ("truncation", C.c_float), ("max_dist", C.c_float)]


# every symbol include/chisel_hip.h declares (tests/test_abi.py checks the header against this list)
class Statistics(C.Structure):
    """chisel_hip_statistics (include/chisel_hip.h): ChunkManager::PrintMemoryStatistics' census"""
    _fields_ = [("n_unknown", C.c_int64), ("n_known_inside", C.c_int64), ("n_known_outside", C.c_int64), ("total_weight", C.c_double),
                ("n_chunks", C.c_int64), ("id_min", C.c_int32 * 3), ("id_max", C.c_int32 * 3)]


EXPORTS = [
    "chisel_hip_abi_version", "chisel_hip_last_error", "chisel_hip_device_count", "chisel_hip_host_alloc", "chisel_hip_host_free", "chisel_hip_create",
    "chisel_hip_destroy", "chisel_hip_reset", "chisel_hip_set_integrator", "chisel_hip_set_stream",
    "chisel_hip_synchronize", "chisel_hip_wait_event", "chisel_hip_record_event", "chisel_hip_order_stream_after_map", "chisel_hip_order_map_after_stream", "chisel_hip_integrate_depth", "chisel_hip_integrate_depth_color",
    "chisel_hip_integrate_batch", "chisel_hip_integrate_pointcloud", "chisel_hip_garbage_collect", "chisel_hip_update_meshes", "chisel_hip_num_chunks",
    "chisel_hip_list_chunks", "chisel_hip_has_chunk", "chisel_hip_download_chunk", "chisel_hip_upload_chunk",
    "chisel_hip_meshes_to_update", "chisel_hip_meshes_to_update_since", "chisel_hip_meshes_to_update_prefetch", "chisel_hip_shell_plan_device", "chisel_hip_shell_segment_bytes", "chisel_hip_export_shells_packed", "chisel_hip_import_shells_packed", "chisel_hip_update_meshes_planned", "chisel_hip_shell_plan_queue", "chisel_hip_import_shells_fixed", "chisel_hip_shell_commit", "chisel_hip_num_meshes", "chisel_hip_list_meshes", "chisel_hip_mesh_size",
    "chisel_hip_download_mesh", "chisel_hip_get_sdf", "chisel_hip_get_sdf_and_gradient", "chisel_hip_save_ply",
    "chisel_hip_save_map", "chisel_hip_load_map", "chisel_hip_export_chunks", "chisel_hip_import_ghost_chunks",
    "chisel_hip_drop_ghost_chunks", "chisel_hip_update_meshes_of", "chisel_hip_condition_depth", "chisel_hip_condition_color", "chisel_hip_publish_cloud",
    "chisel_hip_depth_filter_create", "chisel_hip_depth_filter_destroy", "chisel_hip_depth_filter_update", "chisel_hip_depth_filter_read",
    "chisel_hip_get_counters", "chisel_hip_memory_statistics", "chisel_hip_topology_epoch", "chisel_hip_candidates", "chisel_hip_cloud_candidates", "chisel_hip_mesh_cube", "chisel_hip_write_mesh_ply", "chisel_hip_shade_vertices", "chisel_hip_generate_mesh", "chisel_hip_recompute_mesh", "chisel_hip_integrate_chunk", "chisel_hip_dirty_ids_device", "chisel_hip_mesh_shell_plan",
    "chisel_hip_shell_volume", "chisel_hip_export_shells", "chisel_hip_import_ghost_shells", "chisel_hip_set_profiling", "chisel_hip_get_profile", "chisel_hip_get_launch_stats", "chisel_hip_pool_info", "chisel_hip_mc_tables", "chisel_hip_mesh_cube_values", "chisel_hip_interpolate_vertex", "chisel_hip_raycast", "chisel_hip_chunk_owner", "chisel_hip_frustum", "chisel_hip_frustum_from_vectors", "chisel_hip_create_group",
]
# the device self-tests and debug read-outs include/chisel_hip_selftest.h declares
SELFTEST_EXPORTS = [
    "chisel_hip_kat_truncation", "chisel_hip_kat_dist", "chisel_hip_kat_color", "chisel_hip_kat_color_fresh", "chisel_hip_kat_color_any",
    "chisel_hip_kat_reciprocal", "chisel_hip_kat_floor", "chisel_hip_kat_raycast", "chisel_hip_debug_cloud_stats",
    "chisel_hip_debug_cull_space", "chisel_hip_debug_frustum_range",
]


def library_path():
    return _LIB


def build_library(force=False):
    """hipcc --offload-arch=gfx950 build of cvids_amd/csrc (cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    if not force and os.path.exists(_LIB):
        newest = max(os.path.getmtime(os.path.join(src_dir, f)) for f in os.listdir(src_dir))
        newest = max(newest, os.path.getmtime(os.path.join(_HERE, "..", "include", "chisel_hip.h")))
        if os.path.getmtime(_LIB) >= newest:
            return _LIB
    subprocess.check_call(["make", "-C", src_dir])
    return _LIB


_lib = None


def load_library():
    """Load libchisel_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise ChiselHipError(2, "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(there is no CPU fallback)" % _LIB)
    # torch bundles its own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP runtimes in one process do not
    # work ("No HIP GPUs are available" in whichever comes second), so when torch is installed let it load its
    # runtime first; libchisel_hip.so then binds to that copy.  A C++ host without torch uses /opt/rocm's.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(_LIB)
    if L.chisel_hip_abi_version() != ABI_VERSION:  # array lengths and signatures below are this version's: no call into another
        raise ChiselHipError(2, "%s has ABI version %d, these bindings are for %d: rebuild (`make -C cvids_amd/csrc`)" % (_LIB, L.chisel_hip_abi_version(), ABI_VERSION))
    vp, i32p, f32p, u8p, i64p = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int64)
    L.chisel_hip_last_error.restype = C.c_char_p
    L.chisel_hip_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.chisel_hip_create_group.argtypes = [C.POINTER(Config), i32p, C.c_int, C.POINTER(vp)]
    L.chisel_hip_destroy.argtypes = [vp]
    L.chisel_hip_reset.argtypes = [vp]
    L.chisel_hip_set_integrator.argtypes = [vp, C.POINTER(Integrator)]
    L.chisel_hip_set_stream.argtypes = [vp, vp]
    L.chisel_hip_synchronize.argtypes = [vp]
    L.chisel_hip_integrate_depth.argtypes = [vp, C.POINTER(DepthFrame)]
    L.chisel_hip_integrate_depth_color.argtypes = [vp, C.POINTER(DepthFrame), C.POINTER(ColorFrame)]
    L.chisel_hip_integrate_batch.argtypes = [vp, C.c_int, C.POINTER(DepthFrame), C.POINTER(ColorFrame)]
    L.chisel_hip_integrate_pointcloud.argtypes = [vp, C.POINTER(PointCloud)]
    L.chisel_hip_garbage_collect.argtypes = [vp, i32p, C.c_int]
    L.chisel_hip_update_meshes.argtypes = [vp, C.c_int]
    L.chisel_hip_num_chunks.argtypes = [vp, i64p]
    L.chisel_hip_list_chunks.argtypes = [vp, i32p, C.c_int64, i64p]
    L.chisel_hip_has_chunk.argtypes = [vp, i32p, i32p]
    L.chisel_hip_download_chunk.argtypes = [vp, i32p, f32p, f32p, u8p]
    L.chisel_hip_upload_chunk.argtypes = [vp, i32p, f32p, f32p, u8p]
    L.chisel_hip_meshes_to_update.argtypes = [vp, i32p, C.c_int64, i64p]
    L.chisel_hip_meshes_to_update_since.argtypes = [vp, C.POINTER(C.c_uint64), i32p, C.c_int64, i64p, i32p]
    L.chisel_hip_meshes_to_update_prefetch.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.chisel_hip_host_alloc.argtypes = [C.c_size_t]
    L.chisel_hip_host_alloc.restype = C.c_void_p
    L.chisel_hip_host_free.argtypes = [vp]
    L.chisel_hip_host_free.restype = None
    L.chisel_hip_num_meshes.argtypes = [vp, i64p]
    L.chisel_hip_list_meshes.argtypes = [vp, i32p, C.c_int64, i64p]
    L.chisel_hip_mesh_size.argtypes = [vp, i32p, i64p, i64p]
    L.chisel_hip_download_mesh.argtypes = [vp, i32p, f32p, f32p, f32p, f32p]
    L.chisel_hip_get_sdf.argtypes = [vp, f32p, C.POINTER(C.c_double), i32p]
    L.chisel_hip_get_sdf_and_gradient.argtypes = [vp, f32p, C.POINTER(C.c_double), f32p, i32p]
    L.chisel_hip_save_ply.argtypes = [vp, C.c_char_p]
    L.chisel_hip_export_chunks.argtypes = [vp, i32p, C.c_int, vp, vp, vp, i32p, C.c_int]
    L.chisel_hip_import_ghost_chunks.argtypes = [vp, i32p, C.c_int, vp, vp, vp, i32p, C.c_int]
    L.chisel_hip_drop_ghost_chunks.argtypes = [vp]
    L.chisel_hip_condition_depth.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), vp]
    L.chisel_hip_condition_color.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]
    L.chisel_hip_publish_cloud.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]
    L.chisel_hip_update_meshes_of.argtypes = [vp, i32p, C.c_int]
    L.chisel_hip_depth_filter_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.chisel_hip_depth_filter_destroy.argtypes = [vp]
    L.chisel_hip_depth_filter_update.argtypes = [vp, vp, vp, C.c_double, C.c_int, C.c_int]
    L.chisel_hip_depth_filter_read.argtypes = [vp, C.c_int, vp, C.c_int]
    L.chisel_hip_save_map.argtypes = [vp, C.c_char_p]
    L.chisel_hip_load_map.argtypes = [vp, C.c_char_p]
    L.chisel_hip_get_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
    L.chisel_hip_set_profiling.argtypes = [vp, C.c_int]
    if hasattr(L, "chisel_hip_candidates"):
        L.chisel_hip_topology_epoch.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.chisel_hip_candidates.argtypes = [f32p, f32p, i32p, C.c_float, i32p, C.c_int64, i64p]
        L.chisel_hip_cloud_candidates.argtypes = [vp, C.POINTER(PointCloud), i32p, C.c_int64, i64p]
        L.chisel_hip_mesh_cube.argtypes = [vp, i32p, i32p, f32p, f32p, f32p, i32p, i32p]
        L.chisel_hip_write_mesh_ply.argtypes = [C.c_char_p, f32p, f32p, C.c_int64, i64p, C.c_int64]
        L.chisel_hip_shade_vertices.argtypes = [vp, f32p, C.c_int64, f32p, f32p, C.c_int]
        L.chisel_hip_integrate_chunk.argtypes = [vp, i32p, C.POINTER(DepthFrame), C.POINTER(ColorFrame), i32p]
        L.chisel_hip_recompute_mesh.argtypes = [vp, i32p]
        L.chisel_hip_dirty_ids_device.argtypes = [vp, vp, C.c_int]
        L.chisel_hip_shell_plan_device.argtypes = [vp, vp, C.c_int, C.c_int, i64p]
        L.chisel_hip_shell_segment_bytes.argtypes = [vp, C.c_int64, C.c_int64]
        L.chisel_hip_shell_segment_bytes.restype = C.c_int64
        L.chisel_hip_export_shells_packed.argtypes = [vp, vp, C.c_int64]
        L.chisel_hip_import_shells_packed.argtypes = [vp, vp, C.c_int64]
        L.chisel_hip_update_meshes_planned.argtypes = [vp]
        L.chisel_hip_shell_plan_queue.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int64, vp, vp, C.c_int]
        L.chisel_hip_import_shells_fixed.argtypes = [vp, vp, C.c_int64, vp, C.c_int, C.c_int]
        L.chisel_hip_shell_commit.argtypes = [vp, C.c_int]
        L.chisel_hip_mesh_shell_plan.argtypes = [i32p, C.c_int64, C.c_int, C.c_int, C.c_int, i32p, C.c_int64, i64p, i32p, C.c_int64, i64p]
        L.chisel_hip_shell_volume.argtypes = [C.c_int, C.c_int]
        L.chisel_hip_shell_volume.restype = C.c_int64
        L.chisel_hip_export_shells.argtypes = [vp, i32p, C.c_int, vp, vp, vp, vp, C.c_int]
        L.chisel_hip_import_ghost_shells.argtypes = [vp, i32p, C.c_int, vp, vp, vp, vp, C.c_int]
        L.chisel_hip_generate_mesh.argtypes = [vp, i32p, C.c_int, C.c_int64, C.c_int64, f32p, f32p, f32p, f32p, i64p, i64p]
    if hasattr(L, "chisel_hip_memory_statistics"):
        L.chisel_hip_memory_statistics.argtypes = [vp, C.POINTER(Statistics)]
    L.chisel_hip_get_profile.argtypes = [vp, C.POINTER(C.c_double), i64p, C.c_int]
    L.chisel_hip_get_launch_stats.argtypes = [vp, i64p, C.c_int]
    L.chisel_hip_pool_info.argtypes = [vp, i64p]
    L.chisel_hip_chunk_owner.argtypes = [i32p, C.c_int, C.c_int]
    L.chisel_hip_kat_truncation.argtypes = [C.c_int, C.c_float, f32p, C.c_int, f32p, f32p]
    L.chisel_hip_kat_dist.argtypes = [f32p, C.c_int, f32p]
    L.chisel_hip_kat_color.argtypes = [u8p, C.c_int, u8p]
    # entry points added after ABI version 1 was first built (an older library simply lacks them: A/B runs of tools/)
    for name, types in (("chisel_hip_wait_event", [vp, vp]), ("chisel_hip_record_event", [vp, vp]),
                        ("chisel_hip_order_stream_after_map", [vp, vp]), ("chisel_hip_order_map_after_stream", [vp, vp]),
                        ("chisel_hip_kat_color_fresh", [C.POINTER(C.c_uint)]),
                        ("chisel_hip_kat_color_any", [C.POINTER(C.c_uint)]),
                        ("chisel_hip_debug_cloud_stats", [vp, i64p]),
                        ("chisel_hip_kat_raycast", [f32p, C.c_int, i32p, i32p, i32p, C.c_int, i32p]),
                        ("chisel_hip_kat_reciprocal", [C.POINTER(C.c_ulonglong), C.POINTER(C.c_uint)]),
                        ("chisel_hip_kat_floor", [C.POINTER(C.c_ulonglong), C.POINTER(C.c_uint)])):
        try:
            getattr(L, name).argtypes = types
        except AttributeError:
            pass
    L.chisel_hip_frustum.argtypes = [f32p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, C.c_float, f32p, f32p, f32p]
    L.chisel_hip_debug_frustum_range.argtypes = [f32p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                                                 C.c_int, C.c_float, i32p, i32p, f32p, f32p]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise ChiselHipError(rc, load_library().chisel_hip_last_error().decode(errors="replace"))
