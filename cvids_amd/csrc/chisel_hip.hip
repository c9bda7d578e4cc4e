// chisel_hip.hip -- host side of libchisel_hip.so: the C ABI of include/chisel_hip.h on top of the
// gfx950 kernels in kernels_*.h.  No CPU fallback: every entry point either runs on the GPU or fails
// with CHISEL_HIP_ERR_HIP.
#include "../../include/chisel_hip.h"
#include "../../include/chisel_hip_selftest.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <unordered_map>
#include <set>
#include <unordered_set>
#include <vector>

#include "chisel_device.h"
#include "host_frustum.h"
#include "kernels_cull.h"
#include "kernels_integrate.h"
#include "kernels_map.h"
#include "kernels_mesh.h"
#include "kernels_cloud.h"
#include "kernels_filter.h"

using namespace chisel_hip;

// kernel arguments travel in a 4 KiB segment
static_assert(sizeof(IntegrateParams) + sizeof(MapView) + 6 * sizeof(void *) <= 4096, "integrate_kernel arguments");
static_assert(sizeof(CullParams) + sizeof(PyramidView) + 6 * sizeof(void *) <= 4096, "cull_kernel arguments");
static_assert(sizeof(PyramidParams) + sizeof(PyramidView) + 2 * sizeof(void *) <= 4096, "depth_pyramid_kernel arguments");

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(CHISEL_HIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

struct IdHash {
    size_t operator()(uint64_t k) const { return (size_t)(k * 0x9E3779B97F4A7C15ull); }
};

// ChunkManager::allMeshes lives on the device: every recompute emits the meshes of its chunks into one arena
// (vertices | normals | colors | grids, 3 floats per entry) and a chunk's mesh is a window of it (mesh/Mesh.h:54-58;
// indices are implicit 0..n-1).  An arena is copied to the host when a caller first reads one of its meshes and is
// released when no chunk points into it any more (every one of its chunks has been meshed again since).
struct MeshArena {
    float *dev = nullptr;
    size_t capacity = 0;       // floats allocated at dev
    size_t nv = 0, ng = 0;     // vertices / grids of the whole arena
    bool color = false;
    int live = 0;              // meshes of the map that point into this arena
    std::vector<float> host;   // lazily filled copy
    bool host_valid = false;
    size_t floats() const { return nv * 3 * (color ? 3 : 2) + ng * 3; }
};
constexpr int MESH_INFO_PREFETCH = 8192;  // per-job records copied to the host with every recompute (the rest on demand)
struct MeshRef {
    int arena = -1;            // -1: empty mesh
    size_t v_off = 0, n_v = 0, g_off = 0, n_g = 0;
};

// host-side phase timer (CHISEL_HIP_HOST_TIMING=1): where does an integrate call spend its host time?
struct HostTimer {
    bool on = getenv("CHISEL_HIP_HOST_TIMING") != nullptr;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long calls = 0;
    std::chrono::steady_clock::time_point t;
    void start() { if (on) t = std::chrono::steady_clock::now(); }
    void lap(int i) {
        if (!on) return;
        auto n = std::chrono::steady_clock::now();
        acc[i] += std::chrono::duration<double, std::micro>(n - t).count();
        t = n;
    }
    ~HostTimer() {
        if (on && calls)
            fprintf(stderr, "chisel_hip host us/group: setup %.2f frames %.2f alloc %.2f pyramid-launch %.2f cull-launch %.2f integrate-launch %.2f (groups %ld)\n",
                    acc[0] / calls, acc[1] / calls, acc[2] / calls, acc[3] / calls, acc[4] / calls, acc[5] / calls, calls);
    }
};
HostTimer g_host_timer;
// the same for chisel_hip_update_meshes of a single map: waiting for the previous recompute's totals / job records, the host's
// bookkeeping of that recompute (which chunk's mesh is where), queueing the new one
struct MeshHostTimer {
    double acc[4] = {0, 0, 0, 0};
    long calls = 0;
    std::chrono::steady_clock::time_point t;
    void start() { if (g_host_timer.on) t = std::chrono::steady_clock::now(); }
    void lap(int i) {
        if (!g_host_timer.on) return;
        auto n = std::chrono::steady_clock::now();
        acc[i] += std::chrono::duration<double, std::micro>(n - t).count();
        t = n;
    }
    ~MeshHostTimer() {
        if (g_host_timer.on && calls)
            fprintf(stderr, "chisel_hip host us/recompute: waiting for the previous one's totals %.2f, for its job records %.2f | bookkeeping %.2f | queueing %.2f (recomputes %ld)\n",
                    acc[0] / calls, acc[1] / calls, acc[2] / calls, acc[3] / calls, calls);
    }
};
MeshHostTimer g_mesh_timer;

// roctx ranges (rocprofv3 --marker-trace) around the two halves of a launch set and the mesh recompute: CHISEL_HIP_ROCTX=1.  The
// library is looked up at run time (libroctx64.so of the ROCm installation): no link-time dependency, nothing when the variable is unset.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        if (!getenv("CHISEL_HIP_ROCTX")) return;
        void *h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
Roctx g_roctx;
struct RoctxRange {
    explicit RoctxRange(const char *name) { if (g_roctx.push) g_roctx.push(name); }
    ~RoctxRange() { if (g_roctx.pop) g_roctx.pop(); }
};

struct ProfEvent {
    int kernel;
    hipEvent_t start, stop;
};

}  // namespace

// Buffer sets of the front half (a batch's front may start once the batch that used its set has been integrated), pending sets
// (batch b reads those of b - 1 and b - 2; its pyramid kernel clears its own, last used by b - CHISEL_PENDING_RING) and front streams.
// 4 / 8 / 2 since the end of round 4: with a pending ring of SETS + 3 or more a front half no longer waits for the one two batches
// back, and its buffer set comes free a batch earlier (one rank of eight + 6-8 %, nothing elsewhere; 3 / 4 / 2 before).  A third
// front stream (4 / 8 / 3) costs what it gains (profiles/r04_front_sets.txt).
#ifndef CHISEL_FRONT_SETS
#define CHISEL_FRONT_SETS 4
#endif
#ifndef CHISEL_PENDING_RING
#define CHISEL_PENDING_RING 8
#endif
#ifndef CHISEL_FRONT_STREAMS
#define CHISEL_FRONT_STREAMS 2
#endif
static_assert(CHISEL_FRONT_SETS >= 3 && (CHISEL_PENDING_RING & (CHISEL_PENDING_RING - 1)) == 0 && CHISEL_PENDING_RING >= 4, "front-half rings");
struct chisel_hip_map {
    // a group handle (chisel_hip_create_group, host_group.h): no device state of its own, one shard map per GPU
    bool is_group = false;
    std::vector<chisel_hip_map *> shards;
    void *stages = nullptr;  // std::vector<group::Stage>*: staging of device frames for shards on other devices
    void *host_fanout = nullptr;  // group::HostFanout*: host frames staged once per launch set and broadcast to the group's devices
    void *pool = nullptr;    // group::Pool*: one issuing host thread per shard
    void *mesh_stages_group = nullptr;  // group::MeshStages*: persistent shell staging of update_meshes, per (meshing shard, owner)
    chisel_hip_config cfg;
    int N = 0, V = 0;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    MapView view{};
    MapView *view_dev = nullptr;  // device-resident copy of `view` (cold fields are read from here by the kernels)
    uint64_t hash_capacity = 0;
    chisel_hip_integrator integ{CHISEL_HIP_TRUNC_INVERSE, 8.0f, 1.0f, 1, 0.05f};  // ChiselNode.cpp:54-64 defaults
    // A batch (up to KMAX frames) is handled by one launch set in two halves:
    //   front (auxiliary stream): host->device staging, depth_pyramid_kernel, cull_kernel (these read the frames only),
    //         (which also reads the chunk hash; it tolerates the previous batches' insertions, see kernels_cull.h), brick_kernel
    //   back  (the map's stream): integrate_kernel
    // The fronts of batches b+1 and b+2 run while the back of batch b is still executing -- on two auxiliary streams, so that
    // consecutive fronts overlap each other too (each is a chain of four short kernels: one stream runs them at half the rate the
    // chip could) -- so every buffer the front writes exists CHISEL_FRONT_SETS times (sets rotate per batch); events order the halves.
    // The pending sets (chunks a batch may create) rotate over CHISEL_PENDING_RING buffers of their own: batch b reads those of b-1
    // and b-2 while later batches fill theirs; a batch's pyramid kernel clears the one it is about to use.
    struct BatchSet {
        float2 *pyr_data = nullptr;      // [KMAX][pyr_stride]
        PixelRec *rec_data = nullptr;    // [KMAX][2 + W*H]: per frame two all-NaN records, then the image
        float *depth_stage = nullptr;    // [KMAX][depth_stage_elems]: host frames are copied here
        uint8_t *color_stage = nullptr;  // [KMAX][color_stage_bytes]
        FrameBox *boxes = nullptr;       // [items_capacity][KMAX]: the cull kernel's flags of every (work item, frame), in work-list order
        unsigned short *brick_masks = nullptr;  // [items_capacity][bricks per chunk]: the frames that can touch each brick of a work item (cull_kernel's brick phase)
        int *cand_count = nullptr;       // [COUNT_INTS] device counters of the batch (COUNT_* in kernels_cull.h)
        WorkItem *items = nullptr;       // [items_capacity]: the work-list
        ItemSync *sync = nullptr;        // [items_capacity]: chunk-level state of the work items while the integration kernel runs
        uint64_t *pending = nullptr;     // chunks this batch may create: one of pending_ring (assigned per batch)
        hipStream_t front_stream = nullptr;  // where this batch's front half runs: aux, or the map's stream when nothing is in flight
        hipEvent_t front_done = nullptr;  // recorded on the front stream after the set's front half (work-list, brick masks) is complete
        hipEvent_t cull_done = nullptr;   // ... and after its cull kernel: the set's pending set is complete (what the NEXT batch's cull kernel waits for: it need not
                                          // wait for this batch's brick kernel too -- one rank of eight on the 4-agent stream 253 -> 28x k frames/s)
        // what launch_back needs to launch the set's integration again (a set queued behind a recompute that then did not fit: deferred_set)
        IntegrateParams replay_ip;
        bool replay_color = false, replay_inline = false;
        int replay_total = 0;
        bool staged = false;             // host frames of this set were copied with hipMemcpyAsync (not read over the bus, not on the device)
        unsigned lseq = 0;               // launch number of the set's integration (chisel_hip_map::launch_seq), 0 = never launched
        bool caller_color = false;       // a colour image of the set lies in the caller's device memory (not staged)
        bool front_inline = false;       // the set's front half ran on the map's stream, in front of its integration: no front_done event
    } sets[CHISEL_FRONT_SETS];
    int deferred_set = -1;               // the set whose integration was queued behind a recompute the host has not sized yet (launch_back), or -1
    uint64_t *pending_ring[CHISEL_PENDING_RING] = {};  // [PENDING_CAPACITY + 1] each: the set, then its overflow flag
    hipStream_t aux = nullptr, aux2 = nullptr, aux3 = nullptr;  // the front halves of consecutive batches take them in turn (aux3: CHISEL_FRONT_SETS >= 4 only)
    hipEvent_t call_event = nullptr;     // caller-provided stream: orders the front after the caller's producers
    hipEvent_t mutation_event = nullptr; // map changed outside the integration path (reset, upload): the next front waits
    bool mutation_pending = false;
    hipEvent_t input_event = nullptr;    // chisel_hip_wait_event: the next batch's frames are ready behind this (caller's) event
    bool force_pipeline = false;         // test hook (CHISEL_HIP_FORCE_PIPELINE at creation): the front half always runs on the auxiliary stream
    bool mesh_tiny = false;              // test hook (CHISEL_HIP_MESH_TINY at creation): triangle list and arena start far too small, so every recompute takes the grow-and-emit-again paths
    // The launch heuristics in one place, with the hooks that override them; read ONCE, in chisel_hip_create (they sat in the per-launch
    // host path as getenv calls).  The numbers were fitted on the 640x480 / 1 cm streams of bench.py (sphere room, one and four agents)
    // and are checked on the other scenes, with noise and missing pixels, by tools/measure_table.sh -> profiles/r04_scene_table.txt.
    struct Tuning {
        int fine_below = INTEGRATE_FINE_BELOW;  // work items (16^3-chunk equivalents) below which a launch of >= 4 frames runs at 2 voxels per lane ...
        int fine_min_frames_per_item = 6;       // ... if its items see at least this many frames on average (chains worth shortening)
        int tail_percent = 15;                  // share of a large 4-voxel launch's cost-ordered work-list (the cheap end) that runs at 2 voxels per lane
        double narrow_cull_ratio = 1.5;         // frames look at different parts of space (union id range > ratio x the largest frame's): cull with four waves per workgroup ...
        int narrow_cull_max_shards = 1 << 30;   // ... on maps of at most this many shards (round 4: 2 -- only the cull changed then, no gain at 8; round 5, with the
                                                // refinement's shape going with it: one rank of 2 / 4 / 8 on the 4-agent stream + 3 / + 16 / + 1 %)
        // test / A-B hooks (environment, at creation)
        int force_vpl = 0;                      // CHISEL_HIP_VPL=2|4
        int force_cull_waves = 0;               // CHISEL_HIP_CULL_WAVES=1|4|16
        int defer_totals = 1;                   // CHISEL_HIP_DEFER_TOTALS=0: every integration launch waits for the totals of the recompute in front of it; 2 (test hook): a
                                                // launch is queued unseen even when the totals are already there
        int force_cull_contig = -1;             // CHISEL_HIP_CULL_CONTIG=0|1: which frames a wave of the four-wave cull takes (default: by the frames' ranges)
        int persistent_grid = 0;                // CHISEL_HIP_PERSISTENT=n: a resident grid of n workgroups per SIMD (1 = the build's INTEGRATE_BLOCKS_PER_CU) pulling units from the queue heads
        bool no_zero_copy = false;              // CHISEL_HIP_NO_ZERO_COPY: page-locked host frames are copied like pageable ones
        bool always_wait_packet = false;
        bool bricks_for_one_frame = false;      // CHISEL_HIP_BRICKS_K1=1 (A/B hook): one-frame launch sets of the short form run brick_kernel too (a caller that waits after every frame: 48.7 / 53.6 us per frame with, 50.3 / 48.8 without)
        int mesh_stage_mask = 3;                // CHISEL_HIP_MESH_STAGES=0..3 (timing diagnostic, wrong normals / colours): which of ComputeNormalsFromGradients (1) and ColorizeMesh (2) the triangle kernel runs
        int front_poll_after_publish_us = 18;   // CHISEL_HIP_FRONT_POLL_US: how long after a recompute's triangle kernel has started the host keeps looking for the front half's end before it queues a wait packet (launch_back)
        bool ext_events = false;                // CHISEL_HIP_EXT_EVENTS=0|1: a set's events ride on its last kernels (hipExtLaunchKernelGGL's stop event) instead of separate records; default: on        // CHISEL_HIP_ALWAYS_WAIT_PACKET: no event query before a stream wait
    } tune;
    int64_t launch_stats[CHISEL_HIP_NUM_LAUNCH_STATS] = {};  // chisel_hip_get_launch_stats
    bool refine_off = false;             // test / A-B hook (CHISEL_HIP_REFINE=0 at creation): every cell of every frame of an item's mask counts as needed
    bool force_uncertain = false;        // test hook (CHISEL_HIP_FORCE_UNCERTAIN at creation): every candidate without a slot takes the SLOT_LOOKUP path
    unsigned batch_seq = 0;              // batches issued so far: batch b uses sets[b % CHISEL_FRONT_SETS] and pending_ring[b % CHISEL_PENDING_RING]
    // How far the map's stream has come, WITHOUT events on it: a kernel that carries an event (or an event record, or a wait packet) holds the
    // kernel behind it back by 5-6 us (rocprofv3 timelines of the driver's window with and without them: 309 -> 294 us).  Every integration
    // launch has a number (launch_seq, from 1); its first thread stores it into pinned word [4] (PROGRESS_STARTED: this launch has started, so
    // everything queued before it on the stream is complete -- its own front half when that ran on the map's stream, and the launch before it),
    // the count kernel of a recompute stores the number of the launch in front of it into word [5] (PROGRESS_DONE).  complete_seq: what the
    // host itself has seen complete (a wait for the stream).
    unsigned launch_seq = 0, complete_seq = 0;
    // A pool that grows with the scene, as the reference's unordered_map of heap chunks does (ChunkManager.h:40-55, ChunkManager.cpp:171-174).
    // The voxel arrays are address ranges reserved for view.max_chunks slots (hipMemAddressReserve) of which view.committed have physical
    // memory mapped (hipMemCreate / hipMemMap, in steps of the allocation granularity); the per-slot arrays and the hash (76 bytes per slot
    // against 48 KiB of voxels) are allocated for max_chunks at once, so nothing is ever moved or rehashed.  When the free slots fall under a
    // low-water mark -- checked at launch-set boundaries from the count the integration kernels report in pinned word [7] -- grow_pool() maps
    // more memory and two small kernels on the map's stream give the new slots default voxels and push them onto the free list.
    bool growable = false;
    struct PoolArray {
        char *base = nullptr;
        size_t reserved = 0, mapped = 0;  // bytes
        std::vector<hipMemGenericAllocationHandle_t> handles;
    } pool_mem[3];                        // sdf, wgt, rgbw
    size_t vmm_granularity = 0;
    int64_t grow_events = 0;              // (chisel_hip_pool_info)
    unsigned recomputes = 0, recomputes_seen = 0;  // mesh recomputes issued / as of the previous batch (front-stream choice)
    int items_capacity = 0;
    int pyr_w = 0, pyr_h = 0, pyr_stride = 0;
    PyramidView pyr{};                   // level geometry; .data is set per batch
    size_t depth_stage_elems = 0;
    size_t color_stage_bytes = 0;
    // scratch for queries
    int *scratch_i = nullptr;   // device ints
    size_t scratch_i_elems = 0;
    // meshing state
    int update_meshes_calls = 0;                                       // Chisel.cpp:53 "static int cnt"
    std::unordered_map<uint64_t, MeshRef, IdHash> meshes;              // ChunkManager::allMeshes
    std::vector<MeshArena> arenas;
    std::vector<std::pair<float *, size_t>> arena_pool;                // released arena buffers (floats), reused by later recomputes
    size_t mesh_need_hint = 0;                                         // floats the previous recompute needed (sizes the next arena)
    struct PendingMeshes {                                             // a recompute whose per-chunk results are still on the device
        bool unchecked = false;                                        // its totals have not been looked at yet (check_mesh_totals)
        bool active = false;                                           // its per-chunk bookkeeping is outstanding (resolve_pending_meshes)
        int n = 0;                                                     // jobs
        int arena = -1;
    } pending_meshes;
    int *mesh_totals_host = nullptr;                                   // pinned: [0-3] totals of the recompute in flight and its sequence number (one 16-byte store of the device), [6] sequence number of the job records, [7] check_device_error
    int *mesh_totals_dev = nullptr;                                    // the same memory as the device addresses it
    int *error_flag_host = nullptr;                                    // pinned: the map's error flag (view.error_flag is its device address)
    int *mesh_info_dev = nullptr;                                      // mesh_info_host as the device addresses it
    int mesh_seq = 0;                                                  // recomputes queued so far
    JobInfo *mesh_info_host = nullptr;                                 // pinned: its first MESH_INFO_PREFETCH per-job records
    hipStream_t copy_stream = nullptr;                                 // small device->host copies that must not wait for queued batches
    std::vector<int> ghost_ids;                                        // chunks of other shards imported for meshing (x, y, z triples)
    // the plan of a sharded recompute, made on the device (kernels_map.h: ShellPlan; chisel_hip_shell_plan_device ...)
    ShellPlan shell_plan{};
    int *shell_plan_host = nullptr, *shell_plan_host_dev = nullptr;    // pinned: where the plan's figures reach the host (the one wait of a sharded recompute)
    int64_t shell_send[SHELL_MAX_SHARDS][2] = {}, shell_recv[SHELL_MAX_SHARDS][2] = {};  // (items, voxels) per peer of the latest plan
    int shell_jobs = 0, shell_send_items = 0;
    const unsigned char *ghost_packed = nullptr;                       // the received segments the current ghosts came from (chisel_hip_import_shells_packed): dropped from there
    ShellSegments ghost_segments{};
    int ghost_packed_items = 0;
    // the wait-free form (chisel_hip_shell_plan_queue ...): the segments' stride, the all-reduced status word on the device, whether a drop
    // is queued that leaves the ghosts alone while the mesh step has to be emitted again (check_mesh_totals launches it again), whether the
    // recompute's host-side bookkeeping (pending_mesh_ids) waits for chisel_hip_shell_commit
    hipEvent_t order_events[2] = {nullptr, nullptr};  // chisel_hip_order_stream_after_map / chisel_hip_order_map_after_stream
    int64_t shell_stride = 0;
    unsigned shell_items_grid = 2048;
    const int *shell_abort_dev = nullptr;
    bool shell_fixed_ghosts = false, shell_redrop = false, shell_uncommitted = false;
    std::unordered_set<uint64_t, IdHash> pending_mesh_ids;             // meshesToUpdate entries whose source chunk is gone
    uint64_t pending_version = 1;                                      // bumped when entries join pending_mesh_ids (chisel_hip_meshes_to_update_since)
    uint32_t dirty_epoch = 0;                                          // bumped when the device's dirty list is emptied (recompute, reset)
    bool dirty_tail_queued = false;                                    // chisel_hip_meshes_to_update_prefetch: the listing kernel is queued (or done) for ...
    uint64_t dirty_tail_cursor = 0;                                    // ... this cursor word and ...
    unsigned dirty_tail_batch = 0;                                     // ... this many batches issued
    int *dirty_tail_host = nullptr, *dirty_tail_dev = nullptr;         // pinned: where list_dirty_tail_kernel leaves the new entries of the dirty list
    int batch_frames = KMAX;                                           // frames per launch set in chisel_hip_integrate_batch
    uint64_t ghost_bytes = 0;                                          // group handle: ghost voxel bytes its recomputes have moved between shards
    int *shell_items_dev = nullptr;                                    // staging of the items / offsets of chisel_hip_export_shells / import_ghost_shells
    long long *shell_offs_dev = nullptr;
    int shell_capacity = 0;
    int *shell_first_dev = nullptr;                                    // chisel_hip_import_ghost_shells: first item of every distinct ghost
    int shell_first_capacity = 0;
    bool single_chunk = false;                                         // chisel_hip_integrate_chunk: the next integrate call sees this id only
    int single_id[3] = {0, 0, 0};
    bool mesh_mark_needed = false;                                     // slots were dirtied by something other than integrate_kernel (point clouds), or the kept job list was given up: the next recompute runs mesh_mark_kernel
    bool mesh_totals_clean = true;                                     // an integration launch has been queued since the last recompute (it zeroes the recompute totals)
    int64_t removed_since_recompute = 0;                               // chunks removed since: each may have left a dead entry in the kept job list
    int mesh_jobs_hint = 0;                                            // jobs of the previous recompute (sizes the count kernel's grid)
    int mesh_stages = 3;                                               // MeshParams::stages of the next recompute (chisel_hip_generate_mesh lowers it)
    bool mesh_detached = false;                                        // the next recompute leaves meshesToUpdate alone (chisel_hip_generate_mesh)
    uint64_t topology_epoch = 0;                                       // bumped by everything but integration that adds or removes chunks (chisel_hip_topology_epoch)
    MeshBuffers mesh_buf{};
    struct CloudBuffers {                                              // point-cloud fusion mode (host_cloud.h), allocated on first use
        float *points = nullptr, *colors = nullptr;                    // staging of host clouds
        int64_t capacity = 0;                                          // points
        size_t zeroed_bytes = 0;                                       // table keys | control | counts | cursors: one allocation, one memset per cloud
        CloudView view{};
    } cloud;
    // profiling
    bool profiling = false;
    std::vector<ProfEvent> prof_live;
    std::vector<hipEvent_t> event_pool;
    double prof_ms[CHISEL_HIP_NUM_KERNELS] = {};
    int64_t prof_launches[CHISEL_HIP_NUM_KERNELS] = {};
};

namespace {

// everything queued by the map, on both of its streams (needed before a batch buffer is reallocated)
int sync_all(chisel_hip_map *m) {
    for (int pass = 0; pass < 2; pass++) {
        if (m->aux) HIP_TRY(hipStreamSynchronize(m->aux));
        if (m->aux2) HIP_TRY(hipStreamSynchronize(m->aux2));
        if (m->aux3) HIP_TRY(hipStreamSynchronize(m->aux3));
        if (!pass) HIP_TRY(hipStreamSynchronize(m->stream));
    }
    m->complete_seq = m->launch_seq;  // (note_stream_idle, declared below)
    return CHISEL_HIP_OK;
}

// ---- progress of the map's stream (chisel_hip_map::launch_seq) ---------------------------------------------------------------------------
constexpr int PROGRESS_STARTED = 4, PROGRESS_DONE = 5;  // words of the pinned error-flag block (chisel_device.h)
inline void note_stream_idle(chisel_hip_map *m) { m->complete_seq = m->launch_seq; }  // the host has just waited for the map's stream
// has integration launch L started / ended?  (pinned words the kernels store into; no runtime call, nothing queued)
// (launch numbers are compared through their signed difference: a map that lives through 2^32 launches keeps working)
inline bool seq_le(unsigned a, unsigned b) { return (int)(a - b) <= 0; }
inline bool integrate_started(chisel_hip_map *m, unsigned L) {
    if (seq_le(L, m->complete_seq)) return true;
    volatile int *w = reinterpret_cast<volatile int *>(m->error_flag_host);
    const unsigned started = (unsigned)w[PROGRESS_STARTED], done = (unsigned)w[PROGRESS_DONE];
    // (a word counts only if it names a launch of this map that is not already known to be over: they start at zero)
    if (seq_le(started, m->launch_seq) && !seq_le(started - 1u, m->complete_seq)) m->complete_seq = started - 1u;
    if (seq_le(done, m->launch_seq) && !seq_le(done, m->complete_seq)) m->complete_seq = done;
    return seq_le(L, m->complete_seq) || (seq_le(started, m->launch_seq) && seq_le(L, started));
}
inline bool integrate_done(chisel_hip_map *m, unsigned L) {
    if (seq_le(L, m->complete_seq)) return true;
    (void)integrate_started(m, L);  // (refreshes complete_seq)
    return seq_le(L, m->complete_seq);
}
// The host waits (polling the pinned words) until launch L has started (ended): back-pressure for a caller that runs more than the buffer
// sets ahead of the device, and the order between an inline front half and the next batch's resolve step.  Everything waited for here has
// been queued and needs nothing more from the host; should the words never move (a failed launch) the stream is waited for instead.
int wait_integrate(chisel_hip_map *m, unsigned L, bool until_done) {
    if (L == 0) return CHISEL_HIP_OK;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; spin++) {
        if (until_done ? integrate_done(m, L) : integrate_started(m, L)) return CHISEL_HIP_OK;
        if ((spin & 63u) == 63u) {
            if (until_done && L == m->launch_seq) break;  // nothing queued behind it can report its end: ask the stream
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
        }
    }
    HIP_TRY(hipStreamSynchronize(m->stream));
    note_stream_idle(m);
    return CHISEL_HIP_OK;
}

// ---- growable pool (chisel_hip_map::growable) ---------------------------------------------------------------------------------------------
constexpr int PROGRESS_USED = 7;  // pinned word: committed - free slots as the latest integration launch found them when it started
// physical memory behind the first `bytes` of a reserved array
int pool_map_upto(chisel_hip_map *m, chisel_hip_map::PoolArray &A, size_t bytes) {
    bytes = (bytes + m->vmm_granularity - 1) / m->vmm_granularity * m->vmm_granularity;
    if (bytes > A.reserved) bytes = A.reserved;
    if (bytes <= A.mapped) return CHISEL_HIP_OK;
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = m->device;
    const size_t add = bytes - A.mapped;
    hipMemGenericAllocationHandle_t h;
    HIP_TRY(hipMemCreate(&h, add, &prop, 0));
    hipError_t e = hipMemMap(A.base + A.mapped, add, 0, h, 0);
    if (e == hipSuccess) {
        hipMemAccessDesc acc;
        memset(&acc, 0, sizeof(acc));
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        // (over everything mapped so far, from the reservation's base: on this driver the call fails now and then -- "invalid argument" --
        // for a range that begins behind an earlier mapping, never for one that begins at the base: tools/micro/vmm_probe.hip)
        e = hipMemSetAccess(A.base, A.mapped + add, &acc, 1);
        if (e != hipSuccess) (void)hipMemUnmap(A.base + A.mapped, add);
    }
    if (e != hipSuccess) {
        (void)hipMemRelease(h);
        return fail(CHISEL_HIP_ERR_HIP, std::string("growing the chunk pool: ") + hipGetErrorString(e));
    }
    A.handles.push_back(h);
    A.mapped = bytes;
    return CHISEL_HIP_OK;
}
void pool_release(chisel_hip_map::PoolArray &A) {
    if (!A.base) return;
    if (A.mapped) (void)hipMemUnmap(A.base, A.mapped);
    for (auto h : A.handles) (void)hipMemRelease(h);
    (void)hipMemAddressFree(A.base, A.reserved);
    A = chisel_hip_map::PoolArray();
}
// more committed slots (up to `want`, at most view.max_chunks): memory, default voxels, free list -- queued on the map's stream, nothing waited for
int grow_pool(chisel_hip_map *m, int64_t want) {
    MapView &v = m->view;
    if (!m->growable || v.committed >= v.max_chunks) return CHISEL_HIP_OK;
    const size_t per_chunk = (size_t)m->V * sizeof(float);  // (floats and packed colours alike: 4 bytes per voxel)
    int64_t step = (int64_t)std::max<size_t>(1, m->vmm_granularity / per_chunk);  // whole mapping granules
    int64_t target = std::min<int64_t>(v.max_chunks, (std::max<int64_t>(want, v.committed + 1) + step - 1) / step * step);
    for (int a = 0; a < 3; a++) {
        if (a == 2 && !m->cfg.use_color) continue;
        int rc = pool_map_upto(m, m->pool_mem[a], (size_t)target * per_chunk);
        if (rc) return rc;
    }
    const int first = v.committed, n = (int)(target - v.committed);
    hipLaunchKernelGGL(grow_pool_kernel, dim3(std::min(n, 4096)), dim3(256), 0, m->stream, m->view, m->V, first, n);
    hipLaunchKernelGGL(grow_commit_kernel, dim3(1), dim3(1), 0, m->stream, m->view, n);
    HIP_TRY(hipGetLastError());
    v.committed = (int)target;
    HIP_TRY(hipMemcpyAsync(&m->view_dev->committed, &v.committed, sizeof(int), hipMemcpyHostToDevice, m->stream));
    m->grow_events++;
    return CHISEL_HIP_OK;
}
// Called where chunks are about to be created: grows the pool when what is left of it -- by the latest report of an integration launch,
// less what the launches queued since may have taken -- is less than a quarter of the pool, or than `expect_new` with room to spare.
int maybe_grow(chisel_hip_map *m, int64_t expect_new) {
    if (!m->growable || m->view.committed >= m->view.max_chunks) return CHISEL_HIP_OK;
    volatile int *w = reinterpret_cast<volatile int *>(m->error_flag_host);
    const int64_t used = w[PROGRESS_USED];
    const unsigned started = (unsigned)w[PROGRESS_STARTED];
    const int64_t in_flight = (int64_t)((int)(m->launch_seq - started) > 0 ? (int)(m->launch_seq - started) : 0) + 1;  // launches whose allocations the report does not hold
    const int64_t free_est = (int64_t)m->view.committed - used - in_flight * expect_new;
    if (free_est >= (int64_t)m->view.committed / 4 && free_est >= 2 * expect_new) return CHISEL_HIP_OK;
    return grow_pool(m, std::max<int64_t>(2 * (int64_t)m->view.committed, (int64_t)m->view.committed + 4 * in_flight * expect_new));
}

// ... and for the entry points that create chunks outside the integration path and wait for the stream anyway (point clouds, uploads): the
// free list's true height, at least `n` free slots afterwards (or the pool at its limit)
int ensure_free_exact(chisel_hip_map *m, int64_t n) {
    if (!m->growable || m->view.committed >= m->view.max_chunks) return CHISEL_HIP_OK;
    int top = 0;
    HIP_TRY(hipMemcpyAsync(&top, m->view.free_top, sizeof(int), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    note_stream_idle(m);
    if ((int64_t)top >= n + (int64_t)m->view.committed / 8) return CHISEL_HIP_OK;
    return grow_pool(m, std::max<int64_t>(2 * (int64_t)m->view.committed, (int64_t)m->view.committed + 2 * n));
}

// the chunk hash was changed on the map's stream outside the integration path: the next batch's front half must see it
hipError_t note_map_mutation(chisel_hip_map *m) {
    m->mutation_pending = true;
    m->dirty_tail_queued = false;  // (a prefetched listing of meshesToUpdate predates what this change dirtied or removed)
    return hipEventRecord(m->mutation_event, m->stream);
}

int ensure_scratch(chisel_hip_map *m, size_t elems) {
    if (elems <= m->scratch_i_elems) return CHISEL_HIP_OK;
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (m->scratch_i) HIP_TRY(hipFree(m->scratch_i));
    m->scratch_i = nullptr;
    size_t n = std::max<size_t>(elems, 1024);
    HIP_TRY(hipMalloc(&m->scratch_i, n * sizeof(int)));
    m->scratch_i_elems = n;
    return CHISEL_HIP_OK;
}

hipEvent_t take_event(chisel_hip_map *m) {
    if (!m->event_pool.empty()) {
        hipEvent_t e = m->event_pool.back();
        m->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
struct ProfScope {
    chisel_hip_map *m;
    int k;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(chisel_hip_map *m_, int k_, hipStream_t st_ = nullptr) : m(m_), k(k_), st(st_ ? st_ : m_->stream) {
        if (m->profiling) {
            a = take_event(m);
            b = take_event(m);
            (void)hipEventRecord(a, st);
        }
    }
    ~ProfScope() {
        if (m->profiling) {
            (void)hipEventRecord(b, st);
            m->prof_live.push_back(ProfEvent{k, a, b});
        }
    }
};
int drain_profile(chisel_hip_map *m) {
    if (m->prof_live.empty()) return CHISEL_HIP_OK;
    int rc_sync = sync_all(m);
    if (rc_sync) return rc_sync;
    for (const ProfEvent &p : m->prof_live) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
            m->prof_ms[p.kernel] += ms;
            m->prof_launches[p.kernel] += 1;
        }
        m->event_pool.push_back(p.start);
        m->event_pool.push_back(p.stop);
    }
    m->prof_live.clear();
    return CHISEL_HIP_OK;
}

// Wait for the stream by polling: the wake-up of a blocking wait costs more than the kernels being waited for.
hipError_t wait_stream_spinning(hipStream_t st) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) return hipStreamSynchronize(st);
    }
}

// Waits for everything queued on the map's stream (a caller that waits after every frame pays this per frame: the flag
// lands in pinned memory and the wait polls) and reports a chunk that could not be allocated.
int check_device_error(chisel_hip_map *m) {
    HIP_TRY(wait_stream_spinning(m->stream));
    note_stream_idle(m);
    std::atomic_thread_fence(std::memory_order_acquire);
    volatile int *flags = m->error_flag_host;  // written by the device (raise_error): no copy
    const int cloud = flags[1];
    if (cloud != 0) {  // a property of one cloud, not of the map: reported once
        flags[1] = 0;
        if (flags[0] == 0)
            return fail(CHISEL_HIP_ERR_UNSUPPORTED, cloud == CLOUD_ERR_CAPACITY ? "point cloud: too many chunks or (chunk, point) pairs for one call"
                                                                                : "point cloud: a ray leaves the supported chunk-id range or is too long");
    }
    if (flags[0] != 0)  // the map is incomplete: every wait reports it until chisel_hip_reset
        return fail(CHISEL_HIP_ERR_POOL_FULL, flags[0] == 1 ? "chunk pool exhausted: raise chisel_hip_config.max_chunks"
                                                            : "chunk hash table exhausted: raise chisel_hip_config.max_chunks");
    return CHISEL_HIP_OK;
}

void fill_camera(CameraParams &c, const float *pose, float fx, float fy, float cx, float cy, int W, int H) {
    for (int r = 0; r < 3; r++) {
        for (int k = 0; k < 3; k++) c.R[3 * r + k] = pose[4 * r + k];
        c.t[r] = pose[4 * r + 3];
    }
    c.fx = fx; c.fy = fy; c.cx = cx; c.cy = cy; c.W = W; c.H = H;
}

int ensure_pyramid(chisel_hip_map *m, int W, int H) {
    if (m->sets[0].pyr_data && m->pyr_w == W && m->pyr_h == H) return CHISEL_HIP_OK;
    int rc = sync_all(m);
    if (rc) return rc;
    int off = 0;
    for (int l = 0; l < PYR_LEVELS; l++) {
        int s = PYR_L0 + l;
        m->pyr.w[l] = (W + (1 << s) - 1) >> s;
        m->pyr.h[l] = (H + (1 << s) - 1) >> s;
        m->pyr.off[l] = off;
        off += m->pyr.w[l] * m->pyr.h[l];
    }
    m->pyr_stride = off;
    for (auto &bs : m->sets) {
        if (bs.pyr_data) HIP_TRY(hipFree(bs.pyr_data));
        if (bs.rec_data) HIP_TRY(hipFree(bs.rec_data));
        bs.pyr_data = nullptr;
        bs.rec_data = nullptr;
        HIP_TRY(hipMalloc(&bs.pyr_data, (size_t)off * KMAX * sizeof(float2)));
        // two records of padding in front of every frame's image, NaN once and for all (only pixels are ever written): the
        // integration kernel points voxels that are off the image at record -1
        const size_t rec_floats = ((size_t)W * H + 2) * KMAX * (sizeof(PixelRec) / sizeof(float));
        HIP_TRY(hipMalloc(&bs.rec_data, rec_floats * sizeof(float)));
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(bs.rec_data), 0x7fc00000, rec_floats, m->stream));
    }
    HIP_TRY(hipStreamSynchronize(m->stream));  // the padding is in place before a front half (possibly on the other stream) writes pixels
    m->pyr.data = nullptr;
    m->pyr_w = W;
    m->pyr_h = H;
    return CHISEL_HIP_OK;
}

int check_mesh_totals(chisel_hip_map *m);  // host_mesh.h
void give_up_job_list(chisel_hip_map *m);  // host_mesh.h
// First thing in every entry point that looks at the map or queues work on it (the depth-integration calls see to it themselves, in
// launch_back): a launch set that was queued behind a recompute the host had not sized yet is settled -- the totals are read, and had
// the recompute not fitted it is emitted again and the set's integration replayed -- before anybody can observe the difference.
namespace group {
int settle(chisel_hip_map *g);  // host_group.h: a group's wait-free recompute in flight -- its status, the shards' commits
void forget_recompute(chisel_hip_map *g);
}
inline int settle(chisel_hip_map *m) {
    if (m && m->is_group) return group::settle(m);
    return (m && m->deferred_set >= 0) ? check_mesh_totals(m) : CHISEL_HIP_OK;
}
#define SETTLE(m) do { const int rc_settle_ = settle(m); if (rc_settle_) return rc_settle_; } while (0)

bool mesh_totals_published(const chisel_hip_map *m);  // host_mesh.h
// `stream` is about to read what the front half of set `ps` (an earlier batch) wrote -- its pending set, its work-list buffers: ordered behind
// it.  A front half that ran on an auxiliary stream has its event (no wait packet for what is over); one that ran on the map's stream in
// front of its integration (the short form) is over once that integration has started: the host looks at the progress words, and waits the
// few microseconds itself in the rare case that it got here first.
int wait_for_front_of(chisel_hip_map *m, chisel_hip_map::BatchSet &ps, hipStream_t stream) {
    if (ps.front_inline) return stream == m->stream ? CHISEL_HIP_OK : wait_integrate(m, ps.lseq, false);
    if (!m->tune.always_wait_packet && hipEventQuery(ps.cull_done) == hipSuccess) return CHISEL_HIP_OK;
    (void)hipGetLastError();  // (hipErrorNotReady is not an error)
    HIP_TRY(hipStreamWaitEvent(stream, ps.cull_done, 0));
    return CHISEL_HIP_OK;
}
// The back half of a launch set: the integration kernel on the map's stream.  replay: the launch again, for a set whose first launch
// left the map alone (MC_LATCH; check_mesh_totals).
template <int N>
int launch_back(chisel_hip_map *m, chisel_hip_map::BatchSet &bs, const IntegrateParams &IP, bool color, int total, bool inline_resolve, bool replay) {
    RoctxRange back_range("chisel_hip back half: integrate");
    // ---- back half: the map's stream.  A mesh recompute still in flight must have been sized first (it may have to be
    // emitted again from the voxels as they are now); its front-half work above did not depend on that.
    bool recompute_in_flight = m->pending_meshes.unchecked;
    if (!inline_resolve && !replay) {
        // A front half that has already finished needs no wait packet in front of the integration kernel (the packet is only looked at
        // once the kernel in front of it has ended, and the dispatch behind it only once the packet has retired: 5-6 us of idle chip).
        // With a recompute in flight the map's stream has its count and triangle kernels to chew on (40-50 us) and the host nothing
        // else to do: it looks until the front half is over -- or until the triangle kernel has been running for a while (its first thread
        // publishes the totals: then the integration kernel must get into the queue, with a wait packet if need be).  Without a recompute
        // the host runs batches ahead of the device and the packet is the cheaper wait.
        const bool never = m->tune.always_wait_packet;
        bool done = !never && hipEventQuery(bs.front_done) == hipSuccess;
        if (!done && !never && recompute_in_flight) {
            const auto t0 = std::chrono::steady_clock::now();
            auto t_pub = t0;
            bool published = false;
            for (;;) {
                if ((done = hipEventQuery(bs.front_done) == hipSuccess)) break;
                const auto now = std::chrono::steady_clock::now();
                if (!published && mesh_totals_published(m)) { published = true; t_pub = now; }
                if (published && now - t_pub > std::chrono::microseconds(m->tune.front_poll_after_publish_us)) break;
                if (now - t0 > std::chrono::microseconds(300)) break;
            }
        }
        (void)hipGetLastError();  // (hipErrorNotReady is not an error)
        if (!done) HIP_TRY(hipStreamWaitEvent(m->stream, bs.front_done, 0));
    }
    if (!replay) {
        // A recompute whose totals the host has not seen yet.  Waiting for them here (the triangle kernel publishes them when it STARTS)
        // put the host into the device's loop once per recompute: the launch below reached the queue 10-30 us after that kernel had ended.
        // Instead the launch is queued behind it unseen -- ONE launch set, of frames whose integration reads nothing of the caller's
        // (bs.replay_*) -- and protected on the device: a recompute that did not fit sets MC_LATCH, every wave of the integration kernel
        // leaves at once, and the host, when it next looks (any entry point: settle()), emits again from the untouched map and replays the
        // launch.  Otherwise the totals are looked at as before (by now they are usually there).
        bool same_cam_all = color;
        for (int k = 0; k < IP.n_frames; k++) same_cam_all = same_cam_all && IP.f[k].same_cam;
        // (a replay reads the frames' colour images again.  Staged copies are the library's own; images in the caller's device memory are
        // safe as long as the caller cannot know the integration is over without passing an entry point that settles first -- true on the
        // map's own stream (chisel_hip_synchronize / record_event), not on a stream of the caller's, where it may overwrite them in stream order)
        const bool caller_may_overwrite = color && bs.caller_color && m->stream != m->own_stream;
        const bool may_defer = recompute_in_flight && m->tune.defer_totals && m->deferred_set < 0 && m->cfg.n_shards <= 1 && (!color || same_cam_all) && !caller_may_overwrite &&
                               (m->tune.defer_totals == 2 || !mesh_totals_published(m));
        if (may_defer) {
            m->deferred_set = (int)(&bs - m->sets);
            bs.replay_ip = IP;
            bs.replay_color = color;
            bs.replay_total = total;
            bs.replay_inline = inline_resolve;
            m->launch_stats[8]++;
        } else {
            int rc_m = check_mesh_totals(m);
            if (rc_m) return rc_m;
        }
    }
    int *wc = bs.cand_count + COUNT_ITEMS;
    {
        ProfScope ps(m, CHISEL_HIP_KERNEL_INTEGRATE);
        // Grid.  One unit per wave and the hardware's in-order workgroup dispatch over the cost-ordered work-list is the
        // schedule that works best (the SIMDs issue from their oldest wave first: a persistent wave that pulls a second unit keeps
        // its age and starves the younger waves' first units).  The number of work items is only known on the device, so the grid
        // is sized from the count a recent launch of this map reported (pinned word [2] beside the error flags, written by the
        // integration kernel; it lags by the launches in flight) plus 1/8; surplus workgroups find no unit and leave, a shortfall
        // is pulled from the queue heads by the workgroups as they finish.  Without a report yet: what the chip holds at once.
        // The same figure picks the granularity: 2 voxels per lane for launches of several frames below INTEGRATE_FINE_BELOW items
        // (about three rounds of the chip at 4 voxels per lane), 4 otherwise (kernels_integrate.h).
        const long long hint = (long long)reinterpret_cast<volatile int *>(m->error_flag_host)[2];
        const long long pairs = (long long)reinterpret_cast<volatile int *>(m->error_flag_host)[3];  // (item, frame) pairs of that launch, 0 = unknown
        int vpl = (IP.n_frames >= 4 && hint > 0 && hint * (long long)(N * N * N) < (long long)m->tune.fine_below * 4096 &&
                   (pairs == 0 || pairs >= (long long)m->tune.fine_min_frames_per_item * hint)) ? 2 : 4;
        if (m->tune.force_vpl) vpl = m->tune.force_vpl;
        const int wpc = vpl == 2 ? Geom<N, 2>::WPC : Geom<N, 4>::WPC;
        const int step = vpl == 2 ? Geom<N, 2>::GRID_STEP : Geom<N, 4>::GRID_STEP;
        const long long units = (long long)total * wpc;
        constexpr int WPB = Geom<N, 4>::WPB;
        long long blocks = hint > 0 ? ((hint + hint / 8 + 8) * wpc + WPB - 1) / WPB : (long long)Geom<N, 4>::GRID;
        if (m->tune.persistent_grid) blocks = m->tune.persistent_grid == 1 ? (long long)Geom<N, 4>::GRID : 256ll * m->tune.persistent_grid * (4 / WPB);
        blocks = std::min<long long>(blocks, (units + WPB - 1) / WPB);
        // Two granularities in one launch: the last seventh of the (cost-ordered) items at 2 voxels per lane -- units of half the length
        // where the launch drains -- when the launch is long enough to have a tail worth shortening (>= 2 rounds of the chip).  The
        // kernel takes the split only if the grid covers every unit (it knows the true item count), so the grid is sized for it.
        int split = -1;
        if (vpl == 4 && hint > 0 && hint * (long long)Geom<N, 4>::WPC >= 2ll * Geom<N, 4>::GRID * WPB) {
            const int tail_percent = m->tune.tail_percent;  // (driver window, round 3: 97.8 us without, 95.7 with 15 %, 98.9 with 25 %, 104 with 50 %; no effect late in the stream)
            if (tail_percent > 0) {
                const long long n_est = hint + hint / 8 + 8;
                split = (int)std::max<long long>(0, hint - hint * tail_percent / 100);
                const long long u0 = ((split + 7) / 8) * (long long)Geom<N, 4>::WPC + ((n_est + 7) / 8 - (split + 7) / 8) * (long long)Geom<N, 2>::WPC;
                const long long need = 8 * ((u0 + WPB - 1) / WPB);
                if (need <= (long long)INTEGRATE_GRID_CAP) blocks = std::max(blocks, need);
                else split = -1;
            }
        }
        blocks = std::min<long long>(blocks, (long long)INTEGRATE_GRID_CAP);
        const int grid = (int)std::max<long long>(step, (blocks + step - 1) / step * step);
        m->launch_stats[vpl == 2 ? 0 : (split >= 0 ? 2 : 1)]++;
        m->launch_stats[7]++;
        if (inline_resolve) m->launch_stats[6]++;
        int *queues = bs.cand_count + COUNT_QUEUE0;
        bool same_cam = color;
        for (int k = 0; k < IP.n_frames; k++) same_cam = same_cam && IP.f[k].same_cam;
        if (++m->launch_seq == 0u) { m->launch_seq = 1u; m->complete_seq = 0u; }  // (0 means "never launched"; the wait in integrate_group for a buffer set's last launch has long passed)
        bs.lseq = m->launch_seq;
        const int lseq = (int)bs.lseq;
#define CHISEL_LAUNCH_INTEGRATE(COLOR, SAMECAM, VPL)                                                                                 \
    hipLaunchKernelGGL((integrate_kernel<N, COLOR, SAMECAM, VPL>), dim3(grid), dim3(64 * WPB), 0, m->stream, IP, m->view, m->view_dev, bs.items, \
                       bs.boxes, bs.sync, wc, queues, m->items_capacity, split, lseq, bs.brick_masks)
        if (color && same_cam) {  // CVIDS: depth and colour share one camera (sample.launch:19-20)
            if (vpl == 2) CHISEL_LAUNCH_INTEGRATE(true, true, 2);
            else CHISEL_LAUNCH_INTEGRATE(true, true, 4);
        } else if (color) {
            if (vpl == 2) CHISEL_LAUNCH_INTEGRATE(true, false, 2);
            else CHISEL_LAUNCH_INTEGRATE(true, false, 4);
        } else {
            if (vpl == 2) CHISEL_LAUNCH_INTEGRATE(false, false, 2);
            else CHISEL_LAUNCH_INTEGRATE(false, false, 4);
        }
#undef CHISEL_LAUNCH_INTEGRATE
    }
    HIP_TRY(hipGetLastError());
    m->mesh_totals_clean = true;  // (integrate_kernel's first thread zeroes the next recompute's totals)
    if (replay) return CHISEL_HIP_OK;
    m->batch_seq++;
    if (m->cfg.n_shards <= 1) {  // (the shards of a group are issued by a thread each: one unsynchronised timer would only record their race)
        g_host_timer.lap(5);
        if (g_host_timer.on) g_host_timer.calls++;
    }
    return CHISEL_HIP_OK;
}

// The launch set of one batch.  Front half on the auxiliary stream, back half on the map's stream (see BatchSet).
template <int N>
int launch_group(chisel_hip_map *m, chisel_hip_map::BatchSet &bs, const PyramidParams &PP, const CullParams &CP, const IntegrateParams &IP,
                 bool color) {
    const int total = CullSpace(CP).total;  // candidate slots: every id of the union range, or the owned ones of a sharded map
    g_host_timer.lap(2);
    PyramidView pyr = m->pyr;
    pyr.data = bs.pyr_data;
    // Nothing in flight (the previous batch has been integrated, e.g. a caller that waits after every frame): no second
    // stream to run beside, so the short form -- pyramid, cull with the lookup inline, integrate -- on the map's stream.
    // `front` was decided in integrate_group (it also routes the staging copies).
    hipStream_t front = bs.front_stream;
    const bool inline_resolve = front == m->stream;
    bool front_recorded = false;
    bool narrow_cull_set = false;  // this launch's frames look at different parts of the space (decided where the cull kernel is launched)
    const bool skip_bricks = IP.n_frames == 1 && !m->refine_off && !(inline_resolve && m->tune.bricks_for_one_frame);
    {
    RoctxRange front_range("chisel_hip front half: pyramid, cull, bricks");
    {
        ProfScope ps(m, CHISEL_HIP_KERNEL_PYRAMID, front);
        dim3 grid((PP.W + 63) / 64, (PP.H + 63) / 64, IP.n_frames);
        hipLaunchKernelGGL(depth_pyramid_kernel, grid, dim3(256), 0, front, PP, pyr, bs.cand_count, bs.pending);
    }
    g_host_timer.lap(3);
    // the chunks the batches in flight may create: the pending sets of the previous two (complete once the previous batch's cull kernel
    // is through: the one wait between consecutive front halves).  Nothing in flight in the short form.
    const unsigned b = m->batch_seq;
    const uint64_t *prev_pending = nullptr, *prev2_pending = nullptr;
    if (!inline_resolve) {
        prev_pending = (b >= 1 || m->force_uncertain) ? m->pending_ring[(b + CHISEL_PENDING_RING - 1u) & (CHISEL_PENDING_RING - 1u)] : nullptr;
        prev2_pending = b >= 2 ? m->pending_ring[(b + CHISEL_PENDING_RING - 2u) & (CHISEL_PENDING_RING - 1u)] : nullptr;
        if (b >= 1) {
            int rc_w = wait_for_front_of(m, m->sets[(b + CHISEL_FRONT_SETS - 1u) % CHISEL_FRONT_SETS], front);
            if (rc_w) return rc_w;
        }
    }
    {
        ProfScope ps(m, CHISEL_HIP_KERNEL_CULL, front);
        const CullSpace cspace(CP);
        static_assert(CULL_BLOCK * CULL_BLOCK * CULL_BLOCK == 64, "one wave per block of ids");
        const dim3 cgrid = cspace.sharded ? dim3((total + 63) / 64) : dim3(cspace.nsb[2], cspace.nsb[1], cspace.nsb[0]);
        // frames that look at different parts of the space (several agents in one launch): four waves per workgroup (kernels_cull.h)
        bool narrow_cull = false;
        {
            double vmax = 0.0;
            for (int k2 = 0; k2 < CP.n_frames; k2++) vmax = std::max(vmax, (double)CP.f[k2].range_dim[0] * CP.f[k2].range_dim[1] * CP.f[k2].range_dim[2]);
            narrow_cull = m->cfg.n_shards <= m->tune.narrow_cull_max_shards && (double)CP.range_dim[0] * CP.range_dim[1] * CP.range_dim[2] > m->tune.narrow_cull_ratio * vmax;
            if (m->tune.force_cull_waves) narrow_cull = m->tune.force_cull_waves != 16;
            m->launch_stats[(narrow_cull && IP.n_frames > 4) ? 3 : 4]++;
        }
        // a wave of the four-wave form takes several frames: the ones that share least (kernels_cull.h).  Frames 0 and 1 against frames 0
        // and `waves` by the ids their ranges have in common: interleaved agents share less with their neighbour in the launch
        // ... and one wave per workgroup where the front half runs beside an integration kernel that refills every slot as it frees up: a
        // single-wave workgroup gets in at once, one of four waits for four free slots on ONE CU (tools/micro/beside.hip)
        const bool cull_one_wave = m->tune.force_cull_waves ? m->tune.force_cull_waves == 1 : (!inline_resolve && m->cfg.n_shards <= 1);  // (a shard's cull is an n-th of the map's: the one wave's chain is what is left of it -- one rank of eight 276 -> 206 k)
        int cull_contig = 0;
        if (narrow_cull && CP.n_frames > 4) {
            const int waves = 4, far = waves < CP.n_frames ? waves : CP.n_frames - 1;
            const auto common = [&](const CullFrame &a, const CullFrame &b2) {
                double v = 1.0;
                for (int ax = 0; ax < 3; ax++) {
                    const int lo = std::max(a.range_min[ax], b2.range_min[ax]), hi = std::min(a.range_min[ax] + a.range_dim[ax], b2.range_min[ax] + b2.range_dim[ax]);
                    v *= hi > lo ? (double)(hi - lo) : 0.0;
                }
                return v;
            };
            cull_contig = common(CP.f[0], CP.f[1]) < common(CP.f[0], CP.f[far]) ? 1 : 0;
            if (m->tune.force_cull_contig >= 0) cull_contig = m->tune.force_cull_contig;
        }
        const int *force_flag = m->force_uncertain ? m->sets[0].cand_count + COUNT_ONE : nullptr;
        front_recorded = false;
        // (pipelined form: the cull kernel's completion is the set's cull_done event -- what the next batch's cull kernel waits for)
        const bool cull_recorded = m->tune.ext_events && !m->profiling && !bs.staged && !inline_resolve;
#define CHISEL_LAUNCH_CULL_W(KLV, WV)                                                                                                          \
    do {                                                                                                                                       \
        if (cull_recorded)                                                                                                                     \
            hipExtLaunchKernelGGL((cull_kernel<N, KLV, WV>), cgrid, dim3(64 * CullGeom<KLV, WV>::WAVES), 0, front, nullptr, bs.cull_done, 0, CP, pyr, bs.items, \
                                  bs.boxes, bs.cand_count, m->items_capacity, m->view, prev_pending, prev2_pending, force_flag, bs.pending, bs.sync, \
                                  cull_contig, skip_bricks ? bs.brick_masks : nullptr);                                                        \
        else                                                                                                                                   \
            hipLaunchKernelGGL((cull_kernel<N, KLV, WV>), cgrid, dim3(64 * CullGeom<KLV, WV>::WAVES), 0, front, CP, pyr, bs.items, bs.boxes, bs.cand_count, \
                               m->items_capacity, m->view, prev_pending, prev2_pending, force_flag, bs.pending, bs.sync, cull_contig,        \
                               skip_bricks ? bs.brick_masks : nullptr);                                                                       \
    } while (0)
#define CHISEL_LAUNCH_CULL(KLV) do { if (narrow_cull && cull_one_wave) CHISEL_LAUNCH_CULL_W(KLV, 1); else if (narrow_cull) CHISEL_LAUNCH_CULL_W(KLV, 4); else CHISEL_LAUNCH_CULL_W(KLV, 16); } while (0)
        if (IP.n_frames <= 1) CHISEL_LAUNCH_CULL_W(1, 16);
        else if (IP.n_frames <= 2) CHISEL_LAUNCH_CULL_W(2, 16);
        else if (IP.n_frames <= 4) CHISEL_LAUNCH_CULL_W(4, 16);
        else if (IP.n_frames <= 8) CHISEL_LAUNCH_CULL(8);
        else CHISEL_LAUNCH_CULL(16);
#undef CHISEL_LAUNCH_CULL_W
#undef CHISEL_LAUNCH_CULL
        m->launch_stats[5]++;
        narrow_cull_set = narrow_cull;
    }
    if (!inline_resolve && !(m->tune.ext_events && !m->profiling && !bs.staged)) HIP_TRY(hipEventRecord(bs.cull_done, front));  // the set's pending set is complete (wait_for_front_of)
    if (!skip_bricks) {
        // per work item: which frames can touch which of its bricks (the cull test again at the scale of what a wave of the integration kernel
        // owns, kernels_cull.h).  One wave per item; their number is only known on the device: a persistent grid sized from a recent launch.
        ProfScope ps(m, CHISEL_HIP_KERNEL_RESOLVE, front);
        const int items_hint = reinterpret_cast<volatile int *>(m->error_flag_host)[2];
        // Shape.  Beside an integration kernel (80 registers, six single-wave workgroups per SIMD, every slot refilled the moment it frees up)
        // a workgroup of four waves waits for four free slots on ONE CU, i.e. for that kernel to drain (tools/micro/beside.hip: 145 us
        // instead of 17); single-wave workgroups get in at once -- and take the slots from the integration kernel for as long as they run.
        // Worth it when the front half is what the stream waits for: launches whose frames look at different parts of the space.
        const int rblock = narrow_cull_set ? 64 : BRICK_BLOCK;
        const int rwaves = rblock / 64;
        const long long want = (items_hint > 0 ? items_hint + items_hint / 4 + 16 : 2048) * (IP.n_frames > 8 ? 2 : 1);  // (two waves per item for launches of more than eight frames)
        const int rgrid = (int)std::max<long long>(64, std::min<long long>(8192 / rwaves, (want + rwaves - 1) / rwaves));
        // The brick kernel is the front half's last: in the pipelined form its completion IS the set's front_done event (no record packet
        // behind it); the short form has no event at all (launch_seq).
        front_recorded = m->tune.ext_events && !m->profiling && !bs.staged && !inline_resolve;
        if (front_recorded)
            hipExtLaunchKernelGGL((brick_kernel<N>), dim3(rgrid), dim3(rblock), 0, front, nullptr, bs.front_done, 0, IP, pyr, m->pyr_stride, (const WorkItem *)bs.items,
                                  (const int *)(bs.cand_count + COUNT_ITEMS), m->items_capacity, bs.brick_masks, m->refine_off ? 1 : 0);
        else
            hipLaunchKernelGGL((brick_kernel<N>), dim3(rgrid), dim3(rblock), 0, front, IP, pyr, m->pyr_stride, bs.items, bs.cand_count + COUNT_ITEMS, m->items_capacity,
                               bs.brick_masks, m->refine_off ? 1 : 0);
    }
    // The short form has no event: its kernels sit in front of the set's integration on the map's stream, and a later batch's front half on
    // another stream learns that they are over from the progress words (wait_for_front_of) -- an event carried by (or recorded behind) the
    // last of them would hold the integration kernel back by 5 us, on every frame of a caller that waits after each.
    bs.front_inline = inline_resolve;
    if (!front_recorded && !inline_resolve) HIP_TRY(hipEventRecord(bs.front_done, front));
    }
    g_host_timer.lap(4);
    return launch_back<N>(m, bs, IP, color, total, inline_resolve, false);
}

// the ghosts of a wait-free sharded recompute go (kernels_map.h: the drop in two passes); latch: leave them while the mesh step has to be emitted again
void launch_fixed_drop(chisel_hip_map *m, const int *latch) {
    hipLaunchKernelGGL(shell_reset_boxes_kernel, dim3(m->shell_items_grid), dim3(256), 0, m->stream, m->view, m->ghost_packed, (long long)m->shell_stride, ShellSegments{}, m->cfg.n_shards,
                       m->N, m->shell_abort_dev, latch);
    hipLaunchKernelGGL(shell_remove_ghosts_kernel, dim3((m->shell_items_grid + 255) / 256), dim3(256), 0, m->stream, m->view, m->ghost_packed, (long long)m->shell_stride, ShellSegments{},
                       m->cfg.n_shards, m->shell_abort_dev, latch);
}
// the integration of a set that was queued behind a recompute which then did not fit: its kernel left at once (MC_LATCH), here it is again
int replay_deferred_set(chisel_hip_map *m, int set) {
    chisel_hip_map::BatchSet &bs = m->sets[set];
    m->launch_stats[9]++;
    switch (m->N) {
        case 8: return launch_back<8>(m, bs, bs.replay_ip, bs.replay_color, bs.replay_total, bs.replay_inline, true);
        case 16: return launch_back<16>(m, bs, bs.replay_ip, bs.replay_color, bs.replay_total, bs.replay_inline, true);
        default: return launch_back<32>(m, bs, bs.replay_ip, bs.replay_color, bs.replay_total, bs.replay_inline, true);
    }
}

int check_frame(chisel_hip_map *m, const chisel_hip_depth_frame *f, const chisel_hip_color_frame *c) {
    if (!f || !f->depth || f->width <= 0 || f->height <= 0) return fail(CHISEL_HIP_ERR_INVALID, "bad depth frame");
    if (f->width > 32767 || f->height > 32767) return fail(CHISEL_HIP_ERR_INVALID, "image larger than 32767 pixels per side");
    if ((long long)f->width * f->height > (1ll << 27))  // record offsets are 32 bits
        return fail(CHISEL_HIP_ERR_INVALID, "image larger than 2^27 pixels");
    if (c && (!c->color || c->channels < 1 || c->channels > 4 || c->width <= 0 || c->height <= 0))
        return fail(CHISEL_HIP_ERR_INVALID, "bad colour frame");
    if (c && !m->cfg.use_color)
        return fail(CHISEL_HIP_ERR_INVALID, "map was created without colour voxels (Chunk::GetColorVoxelMutable would throw)");
    return CHISEL_HIP_OK;
}

// n <= KMAX frames of one image size, all with or all without colour, in one launch set
int integrate_group(chisel_hip_map *m, int n, const chisel_hip_depth_frame *frames, const chisel_hip_color_frame *colors) {
    g_host_timer.start();
    HIP_TRY(hipSetDevice(m->device));
    const int W = frames[0].width, H = frames[0].height;
    const size_t npx = (size_t)W * H;
    const bool color = colors != nullptr;
    int rc = ensure_pyramid(m, W, H);
    if (rc) return rc;

    PyramidParams PP;
    CullParams CP;
    IntegrateParams IP;
    memset(&PP, 0, sizeof(PP));
    memset(&CP, 0, sizeof(CP));
    memset(&IP, 0, sizeof(IP));
    IntegratorParams ip;
    ip.trunc_kind = m->integ.truncator_kind;
    ip.trunc_param = m->integ.truncator_param;
    ip.weight = m->integ.weight;
    ip.carving = m->integ.carving_enabled ? 1 : 0;
    ip.carving_dist = m->integ.carving_dist;
    ip.res = m->cfg.voxel_resolution;
    ip.half_res = m->cfg.voxel_resolution * 0.5f;                                           // ChunkManager.cpp:52
    ip.diag = (float)(2.0 * ::sqrt((double)3.0f) * (double)m->cfg.voxel_resolution);        // ProjectionIntegrator.h:58,109
    ip.max_depth = color ? 100.0f : 50.0f;
    ip.n_shards = m->cfg.n_shards;
    ip.shard_rank = m->cfg.shard_rank;
    ip.shard_block = m->cfg.shard_block;
    ip.single_chunk = m->single_chunk ? 1 : 0;
    PP.ip = CP.ip = IP.ip = ip;
    PP.W = W;
    PP.H = H;
    chisel_hip_map::BatchSet &bs = m->sets[m->batch_seq % CHISEL_FRONT_SETS];
    bs.pending = m->pending_ring[m->batch_seq & (CHISEL_PENDING_RING - 1u)];
    bs.staged = false;
    bs.caller_color = false;
    PP.rec_stride = (int)npx + 2;
    PP.pyr_stride = m->pyr_stride;
    PP.rec = bs.rec_data + 2;
    CP.n_frames = IP.n_frames = n;
    CP.pyr_stride = m->pyr_stride;

    // host-resident images: grow the staging rings first (a reallocation must not race with queued copies)
    bool any_host_depth = false, any_host_color = false;
    size_t color_bytes = 0;
    for (int k = 0; k < n; k++) {
        any_host_depth |= !frames[k].on_device;
        if (color) {
            any_host_color |= !colors[k].on_device;
            color_bytes = std::max(color_bytes, (size_t)colors[k].width * colors[k].height * colors[k].channels);
        }
    }
    if (any_host_depth && npx > m->depth_stage_elems) {
        rc = sync_all(m);
        if (rc) return rc;
        for (auto &b2 : m->sets) {
            if (b2.depth_stage) HIP_TRY(hipFree(b2.depth_stage));
            b2.depth_stage = nullptr;
            HIP_TRY(hipMalloc(&b2.depth_stage, npx * KMAX * sizeof(float)));
        }
        m->depth_stage_elems = npx;
    }
    if (any_host_color && color_bytes > m->color_stage_bytes) {
        rc = sync_all(m);
        if (rc) return rc;
        for (auto &b2 : m->sets) {
            if (b2.color_stage) HIP_TRY(hipFree(b2.color_stage));
            b2.color_stage = nullptr;
            HIP_TRY(hipMalloc(&b2.color_stage, color_bytes * KMAX));
        }
        m->color_stage_bytes = color_bytes;
    }
    // Where does the front half run?  On the auxiliary stream, beside the batch that is being integrated -- unless nothing
    // is in flight (a caller that waits after every frame; the first batch): then there is nothing to run beside and the
    // short form on the map's stream has the lower latency (launch_group).  A stream given by the caller orders the front
    // half after whatever the caller queued there (the producers of device frames), which is the same stream order.
    const bool meshing = m->recomputes != m->recomputes_seen;  // a mesh recompute was issued since the previous batch
    m->recomputes_seen = m->recomputes;
    {
        const chisel_hip_map::BatchSet &prev = m->sets[(m->batch_seq + CHISEL_FRONT_SETS - 1u) % CHISEL_FRONT_SETS];  // the previous batch's (integrations complete in order)
        const bool idle = (m->batch_seq == 0 || integrate_done(m, prev.lseq)) && !m->pending_meshes.unchecked && !m->force_pipeline;
        // Two auxiliary streams, taken in turn, double the rate of the front halves -- which is what a stream without meshing
        // hangs on (98 -> 122 k frames/s) -- but a stream that recomputes meshes between its batches is paced by integration +
        // meshing on the map's stream, and a second front half beside them only takes 3 % from it: one stream then.
        hipStream_t side = (m->batch_seq & 1u) && m->aux2 && !meshing ? m->aux2 : m->aux;
        if (m->aux3 && !meshing) side = (m->batch_seq % 3u) == 2u ? m->aux3 : ((m->batch_seq % 3u) == 1u ? m->aux2 : m->aux);
        bs.front_stream = (idle || m->stream != m->own_stream || m->aux == m->own_stream) ? m->stream : side;
    }
    hipStream_t front = bs.front_stream;
    if (front != m->stream) {
        // the front half may start as soon as the batch that last used this buffer set (CHISEL_FRONT_SETS batches ago) has been integrated;
        // only its resolve step waits for the previous batch's (launch_group).  The previous front half may have run on the map's
        // stream (short form): what it wrote must be complete before this one's kernels read the candidates' neighbours' state
        // (the integration that last read this buffer set, CHISEL_FRONT_SETS batches ago: the HOST waits -- a caller that far ahead of the
        // device has nothing to gain from queueing more, and the map's stream carries no event for it, see chisel_hip_map::launch_seq)
        if (m->batch_seq >= (unsigned)CHISEL_FRONT_SETS) {
            rc = wait_integrate(m, bs.lseq, true);
            if (rc) return rc;
        }
        // this batch's pyramid kernel clears the pending buffer that the resolve step of batch b-2 still reads (as its b-4)
        // (with a pending ring of NSETS + 3 or more the buffer's last readers finished before the set came free: no such wait)
        if (m->batch_seq >= 2 && CHISEL_PENDING_RING < CHISEL_FRONT_SETS + 3) {
            rc = wait_for_front_of(m, m->sets[(m->batch_seq + CHISEL_FRONT_SETS - 2u) % CHISEL_FRONT_SETS], front);
            if (rc) return rc;
        }
        // a stream paced by integration + meshing gains nothing from a front half that starts a batch earlier (it only runs
        // beside more of the kernels that set the pace): the two-set rule for it
        if (meshing && m->batch_seq >= 2) {
            rc = wait_integrate(m, m->sets[(m->batch_seq + CHISEL_FRONT_SETS - 2u) % CHISEL_FRONT_SETS].lseq, true);
            if (rc) return rc;
        }
        if (m->mutation_pending) HIP_TRY(hipStreamWaitEvent(front, m->mutation_event, 0));
    }
    m->mutation_pending = false;
    if (m->input_event) {  // depth is read by the front half, colour by the integration kernel
        if (front != m->stream) HIP_TRY(hipStreamWaitEvent(front, m->input_event, 0));
        HIP_TRY(hipStreamWaitEvent(m->stream, m->input_event, 0));
        m->input_event = nullptr;
    }

    g_host_timer.lap(0);
    int umin[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, umax[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    for (int k = 0; k < n; k++) {
        const chisel_hip_depth_frame *f = &frames[k];
        FrameCam &F = IP.f[k];
        fill_camera(F.cam, f->pose, f->fx, f->fy, f->cx, f->cy, W, H);
        CP.f[k].cam = F.cam;
        if (f->on_device) {
            PP.depth[k] = f->depth;
        } else {
            // Page-locked host memory is addressable by the device: the pyramid kernel reads the frame straight over the bus, once
            // and coalesced (a copy-engine transfer per frame costs more in launch latency than it moves).  Pageable memory is
            // copied by the runtime during the call.
            hipPointerAttribute_t attr;
            const float *mapped = nullptr;
            if (hipPointerGetAttributes(&attr, f->depth) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer)
                mapped = static_cast<const float *>(attr.devicePointer);
            else
                (void)hipGetLastError();  // not a registered pointer: plain pageable memory
            if (mapped && !m->tune.no_zero_copy) {
                PP.depth[k] = mapped;
            } else {
                float *dst = bs.depth_stage + (size_t)k * m->depth_stage_elems;
                HIP_TRY(hipMemcpyAsync(dst, f->depth, npx * sizeof(float), hipMemcpyHostToDevice, front));
                PP.depth[k] = dst;
                bs.staged = true;
            }
        }
        F.rec = PP.rec + (size_t)k * PP.rec_stride;
        if (color) {
            const chisel_hip_color_frame *c = &colors[k];
            fill_camera(F.ccam, c->pose, c->fx, c->fy, c->cx, c->cy, c->width, c->height);
            F.color_channels = c->channels;
            F.same_cam = memcmp(&F.cam, &F.ccam, sizeof(CameraParams)) == 0 ? 1 : 0;
            if (c->on_device) {
                F.color = c->color;
                bs.caller_color = true;
            } else {
                uint8_t *dst = bs.color_stage + (size_t)k * m->color_stage_bytes;
                HIP_TRY(hipMemcpyAsync(dst, c->color, (size_t)c->width * c->height * c->channels, hipMemcpyHostToDevice, front));
                F.color = dst;
                bs.staged = true;
            }
        }
        hostmath::FrustumRange fr = hostmath::frustum_range(f->pose, f->near_plane, f->far_plane, f->fy, f->cy, W, H, m->N,
                                                            m->cfg.voxel_resolution);
        if (m->single_chunk)  // ProjectionIntegrator::Integrate(..., chunk): that chunk, whether or not the frustum's range holds it
            for (int a = 0; a < 3; a++) {
                fr.range_min[a] = m->single_id[a];
                fr.range_dim[a] = 1;
            }
        for (int a = 0; a < 3; a++) {
            CP.f[k].range_min[a] = fr.range_min[a];
            CP.f[k].range_dim[a] = fr.range_dim[a];
            if (fr.range_dim[a] <= 0 || fr.range_min[a] < -ID_BIAS + 2 || fr.range_min[a] + fr.range_dim[a] > ID_BIAS - 2)
                return fail(CHISEL_HIP_ERR_INVALID, "frustum outside the addressable chunk-id range (pose not finite?)");
            umin[a] = std::min(umin[a], fr.range_min[a]);
            umax[a] = std::max(umax[a], fr.range_min[a] + fr.range_dim[a]);
        }
        memcpy(CP.f[k].planes, fr.planes, sizeof(fr.planes));
    }
    g_host_timer.lap(1);
    double total_d = 1.0;
    for (int a = 0; a < 3; a++) {
        CP.range_min[a] = umin[a];
        CP.range_dim[a] = umax[a] - umin[a];
        total_d *= (double)CP.range_dim[a];
    }
    if (total_d > 2.0e8) return fail(CHISEL_HIP_ERR_INVALID, "frusta cover more than 2e8 chunks: far plane / resolution mismatch");
    const int total = (int)total_d;
    if (total > m->items_capacity) {
        rc = sync_all(m);
        if (rc) return rc;
        // geometric growth: the candidate range changes with every pose, a reallocation (stream sync) must stay rare
        int cap = std::max(1 << 17, m->items_capacity);
        while (cap < total) cap *= 2;
        for (auto &b2 : m->sets) {
            if (b2.boxes) HIP_TRY(hipFree(b2.boxes));
            if (b2.brick_masks) HIP_TRY(hipFree(b2.brick_masks));
            b2.brick_masks = nullptr;
            if (b2.items) HIP_TRY(hipFree(b2.items));
            if (b2.sync) HIP_TRY(hipFree(b2.sync));
            b2.boxes = nullptr;
            b2.items = nullptr;
            b2.sync = nullptr;
            HIP_TRY(hipMalloc(&b2.items, (size_t)cap * sizeof(WorkItem)));
            HIP_TRY(hipMalloc(&b2.sync, (size_t)cap * sizeof(ItemSync)));
            HIP_TRY(hipMalloc(&b2.boxes, (size_t)cap * KMAX * sizeof(FrameBox)));
            HIP_TRY(hipMalloc(&b2.brick_masks, (size_t)cap * (m->N / 8) * (m->N / 8) * (m->N / 4) * sizeof(unsigned short)));
        }
        m->items_capacity = cap;
    }
    {
        // (a launch set creates at most its work items' worth of chunks; the hint is a recent launch's count)
        const int items_hint = reinterpret_cast<volatile int *>(m->error_flag_host)[2];
        rc = maybe_grow(m, std::max<int64_t>(256, 2 * (int64_t)items_hint));
        if (rc) return rc;
    }
    switch (m->N) {
        case 8: return launch_group<8>(m, bs, PP, CP, IP, color);
        case 16: return launch_group<16>(m, bs, PP, CP, IP, color);
        case 32: return launch_group<32>(m, bs, PP, CP, IP, color);
    }
    return fail(CHISEL_HIP_ERR_UNSUPPORTED, "chunk size");
}

// frames in order; consecutive frames of equal image size are grouped KMAX at a time
int integrate_frames(chisel_hip_map *m, int n, const chisel_hip_depth_frame *frames, const chisel_hip_color_frame *colors,
                     int max_group) {
    for (int i = 0; i < n; i++) {
        int rc = check_frame(m, &frames[i], colors ? &colors[i] : nullptr);
        if (rc) return rc;
    }
    const int kmax = std::max(1, std::min(max_group, KMAX));
    // chisel_hip_wait_event covers every frame of this call: each launch set waits for it (a later set's front half runs on the
    // other auxiliary stream and is not ordered behind the first set's pyramid kernel)
    const hipEvent_t call_input = m->input_event;
    struct Disarm {
        chisel_hip_map *m;
        ~Disarm() { m->input_event = nullptr; }
    } disarm{m};
    int i = 0;
    while (i < n) {
        // the run of frames of one image size, cut into launch sets of equal length (10 frames: 5 + 5, not 8 + 2)
        int run = 1;
        while (i + run < n && frames[i + run].width == frames[i].width && frames[i + run].height == frames[i].height) run++;
        int sets = (run + kmax - 1) / kmax;
        while (run > 0) {
            const int g = (run + sets - 1) / sets;
            m->input_event = call_input;
            int rc = integrate_group(m, g, frames + i, colors ? colors + i : nullptr);
            if (rc) return rc;
            i += g;
            run -= g;
            sets--;
        }
    }
    return CHISEL_HIP_OK;
}

// ids of dirty slots -> host vector (packed keys)
int fetch_listed(chisel_hip_map *m, bool dirty_only, std::vector<int> &ids, std::vector<int> *slots) {
    int rc = ensure_scratch(m, (size_t)m->view.max_chunks * 4 + 16);
    if (rc) return rc;
    int *d_count = m->scratch_i;
    int *d_ids = m->scratch_i + 16;
    int *d_slots = d_ids + (size_t)m->view.max_chunks * 3;
    HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(int), m->stream));
    const int blocks = (m->view.max_chunks + 255) / 256;
    if (dirty_only)
        hipLaunchKernelGGL(list_slots_kernel<true>, dim3(blocks), dim3(256), 0, m->stream, m->view, d_ids, d_slots,
                           m->view.max_chunks, d_count);
    else
        hipLaunchKernelGGL(list_slots_kernel<false>, dim3(blocks), dim3(256), 0, m->stream, m->view, d_ids, d_slots,
                           m->view.max_chunks, d_count);
    int count = 0;
    HIP_TRY(hipMemcpyAsync(&count, d_count, sizeof(int), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    ids.resize((size_t)count * 3);
    if (slots) slots->resize(count);
    if (count) {
        HIP_TRY(hipMemcpy(ids.data(), d_ids, (size_t)count * 3 * sizeof(int), hipMemcpyDeviceToHost));
        if (slots) HIP_TRY(hipMemcpy(slots->data(), d_slots, (size_t)count * sizeof(int), hipMemcpyDeviceToHost));
    }
    return CHISEL_HIP_OK;
}

int lookup_slots(chisel_hip_map *m, const int *ids, int n, std::vector<int> &slots) {
    slots.assign(n, -1);
    if (n == 0) return CHISEL_HIP_OK;
    int rc = ensure_scratch(m, (size_t)n * 4);
    if (rc) return rc;
    int *d_ids = m->scratch_i, *d_slots = m->scratch_i + (size_t)n * 3;
    HIP_TRY(hipMemcpyAsync(d_ids, ids, (size_t)n * 3 * sizeof(int), hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(lookup_kernel, dim3((n + 255) / 256), dim3(256), 0, m->stream, m->view, d_ids, n, d_slots);
    HIP_TRY(hipMemcpyAsync(slots.data(), d_slots, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return CHISEL_HIP_OK;
}

void expand27(const std::vector<int> &ids, std::unordered_set<uint64_t, IdHash> &out) {
    for (size_t i = 0; i + 2 < ids.size(); i += 3)
        for (int dx = -1; dx <= 1; dx++)
            for (int dy = -1; dy <= 1; dy++)
                for (int dz = -1; dz <= 1; dz++) out.insert(pack_id(ids[i] + dx, ids[i + 1] + dy, ids[i + 2] + dz));
}

// header of the binary map dump (chisel_hip_save_map / load_map; layout documented in chisel_hip.h)
struct MapFileHeader {
    char magic[8];
    int32_t chunk_edge;
    float resolution;
    int32_t has_color;
    int32_t spare;
    int64_t n_chunks;
};
static_assert(sizeof(MapFileHeader) == 32, "map file header");

// group handles (host_group.h): what the entry points defined in host_mesh.h / host_cloud.h dispatch to
namespace group {
int update_meshes(chisel_hip_map *g, int force);
int get_sdf(chisel_hip_map *g, const float pos[3], double *dist, int *found);
int get_sdf_and_gradient(chisel_hip_map *g, const float pos[3], double *dist, float grad[3], int *found);
int integrate_cloud(chisel_hip_map *g, const chisel_hip_pointcloud *cloud);
int list_meshes(chisel_hip_map *g, int *ids, int64_t max_ids, int64_t *count);
int save_ply(chisel_hip_map *g, const char *path);
inline chisel_hip_map *owner_map(chisel_hip_map *g, const int id[3]) {
    return g->shards[chunk_owner(id[0], id[1], id[2], (int)g->shards.size(), g->cfg.shard_block)];
}
}  // namespace group

}  // namespace

#include "host_mesh.h"
#include "host_cloud.h"
#include "host_group.h"

extern "C" {

int chisel_hip_abi_version(void) { return CHISEL_HIP_ABI_VERSION; }
const char *chisel_hip_last_error(void) { return g_last_error.c_str(); }

int chisel_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strstr(p.gcnArchName, "gfx950")) ok++;
    }
    return ok;
}

// DepthImage / ColorImage buffers (DepthImage.h:42-52, ColorImage.h:44-58): page-locked, so that an integrate call reads them without the
// staged copy of pageable memory (integrate_group: hipPointerGetAttributes finds the mapping).  Which pointers came from hipHostMalloc
// is remembered here (free() of the others).
namespace {
std::mutex g_host_alloc_mutex;
std::unordered_set<void *> g_host_pinned;
}  // namespace
void *chisel_hip_host_alloc(size_t bytes) {
    if (bytes == 0) bytes = 1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) {
        void *p = nullptr;
        if (hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess && p) {
            std::lock_guard<std::mutex> lock(g_host_alloc_mutex);
            g_host_pinned.insert(p);
            return p;
        }
    }
    (void)hipGetLastError();
    return malloc(bytes);
}
void chisel_hip_host_free(void *p) {
    if (!p) return;
    bool pinned = false;
    {
        std::lock_guard<std::mutex> lock(g_host_alloc_mutex);
        pinned = g_host_pinned.erase(p) != 0;
    }
    if (pinned) (void)hipHostFree(p);
    else free(p);
}

int chisel_hip_chunk_owner(const int id[3], int n_shards, int shard_block) {
    return chunk_owner(id[0], id[1], id[2], n_shards < 1 ? 1 : n_shards, shard_block < 1 ? 2 : shard_block);
}

int chisel_hip_create(const chisel_hip_config *cfg, chisel_hip_map **out) {
    if (!cfg || !out) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    *out = nullptr;
    const int N = cfg->chunk_size[0];
    if (cfg->chunk_size[1] != N || cfg->chunk_size[2] != N || (N != 8 && N != 16 && N != 32))
        return fail(CHISEL_HIP_ERR_UNSUPPORTED,
                    "chunk_size must be cubic 8, 16 or 32 (the reference's Chunk::GetVoxelID, Chunk.h:81-84, is only valid for cubic chunks)");
    if (!(cfg->voxel_resolution > 0.0f)) return fail(CHISEL_HIP_ERR_INVALID, "voxel_resolution must be > 0");
    const int n_shards = cfg->n_shards < 1 ? 1 : cfg->n_shards;
    if (cfg->shard_rank < 0 || cfg->shard_rank >= n_shards) return fail(CHISEL_HIP_ERR_INVALID, "shard_rank out of range");

    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(CHISEL_HIP_ERR_HIP, "no HIP device visible: libchisel_hip has no CPU path");
    int dev = cfg->device_id;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    if (dev >= ndev) return fail(CHISEL_HIP_ERR_INVALID, "device_id out of range");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (!strstr(prop.gcnArchName, "gfx950"))
        return fail(CHISEL_HIP_ERR_HIP, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 (MI355X) only");
    HIP_TRY(hipSetDevice(dev));

    chisel_hip_map *m = new chisel_hip_map();
    m->cfg = *cfg;
    m->cfg.n_shards = n_shards;
    m->cfg.shard_block = cfg->shard_block < 1 ? 2 : cfg->shard_block;
    m->N = N;
    m->V = N * N * N;
    m->device = dev;
    const size_t bytes_per_chunk = (size_t)m->V * (8 + (cfg->use_color ? 4 : 0));
    // max_chunks > 0: a pool of exactly that many chunks (exhaustion = CHISEL_HIP_ERR_POOL_FULL); 0: the default size, growing with the scene;
    // < 0: -max_chunks to begin with, growing.  A growing pool is laid out for 16 times its first size, or for what three quarters of
    // the device's memory hold, whichever is less (C_max); CHISEL_HIP_GROW=0 keeps every pool at its first size.
    int64_t C0 = cfg->max_chunks;
    bool growable = C0 <= 0 && !(getenv("CHISEL_HIP_GROW") && atoi(getenv("CHISEL_HIP_GROW")) == 0);
    if (C0 == 0) C0 = (int64_t)((6ull << 30) / bytes_per_chunk);
    else if (C0 < 0) C0 = -C0;
    if (C0 > (1 << 30)) C0 = 1 << 30;
    int64_t C = C0;  // C_max
    if (growable) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = 0;
        const int64_t by_memory = (int64_t)((double)total_b * 0.75 / (double)bytes_per_chunk);
        const int64_t step = std::max<int64_t>(1, ((int64_t)2 << 20) / ((int64_t)m->V * 4));  // chunks per 2 MiB page of a voxel array: what the pool grows in
        C0 = (C0 + step - 1) / step * step;
        C = std::max<int64_t>(C0, std::min<int64_t>(std::max<int64_t>(16 * C0, 8 * step), by_memory));
        if (C > (1 << 30)) C = 1 << 30;
        if (C == C0) growable = false;
    }
    m->cfg.max_chunks = C;
    m->growable = growable;
    uint64_t hc = 1024;
    while (hc < (uint64_t)C * 2) hc <<= 1;
    m->hash_capacity = hc;

    auto cleanup = [&](int rc) {
        chisel_hip_destroy(m);
        return rc;
    };
#define HIP_TRY_C(expr)                                                                                       \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            return cleanup(fail(CHISEL_HIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)));      \
    } while (0)
    // Diagnostic CHISEL_HIP_FRONT_CUS=k: k compute units of every XCD set aside for the front half's streams, the map's own stream on
    // the others (bit i of a CU mask = CU i / 8 of XCD i % 8 on this part).  Measured in round 4 (profiles/r04_cu_partition.txt,
    // tools/ab_cus.sh): slower at every k on every stream -- the front half is a third of the chip's work, not a latency problem.
    const int front_cus = getenv("CHISEL_HIP_FRONT_CUS") ? atoi(getenv("CHISEL_HIP_FRONT_CUS")) : 0;
    if (front_cus > 0 && front_cus < 32 && !getenv("CHISEL_HIP_SERIAL")) {
        uint32_t mask_front[8] = {0}, mask_back[8] = {0};
        for (int i = 0; i < 256; i++) ((i / 8 < front_cus) ? mask_front : mask_back)[i / 32] |= 1u << (i % 32);
        if (getenv("CHISEL_HIP_BACK_ALL_CUS")) for (auto &w : mask_back) w = ~0u;
        HIP_TRY_C(hipExtStreamCreateWithCUMask(&m->own_stream, 8, mask_back));
        HIP_TRY_C(hipExtStreamCreateWithCUMask(&m->aux, 8, mask_front));
        HIP_TRY_C(hipExtStreamCreateWithCUMask(&m->aux2, 8, mask_front));
    } else if (getenv("CHISEL_HIP_SERIAL")) {
        HIP_TRY_C(hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking));
        m->aux = m->own_stream;  // diagnostic: both halves on one stream, nothing overlaps
    } else {
        HIP_TRY_C(hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking));
        // the front half is short and feeds the long integration kernel of the next batch: let its workgroups go first
        int least = 0, greatest = 0;
        HIP_TRY_C(hipDeviceGetStreamPriorityRange(&least, &greatest));
        const char *pr = getenv("CHISEL_HIP_AUX_PRIORITY");  // diagnostic: "low" / "normal" instead of the default high
        const int prio = pr && !strcmp(pr, "low") ? least : (pr && !strcmp(pr, "normal") ? (least + greatest) / 2 : greatest);
        HIP_TRY_C(hipStreamCreateWithPriority(&m->aux, hipStreamNonBlocking, prio));
        if (!getenv("CHISEL_HIP_ONE_FRONT_STREAM")) HIP_TRY_C(hipStreamCreateWithPriority(&m->aux2, hipStreamNonBlocking, prio));
        if (CHISEL_FRONT_SETS >= 4 && CHISEL_FRONT_STREAMS >= 3 && m->aux2) HIP_TRY_C(hipStreamCreateWithPriority(&m->aux3, hipStreamNonBlocking, prio));
    }
    m->stream = m->own_stream;
    HIP_TRY_C(hipStreamCreateWithFlags(&m->copy_stream, hipStreamNonBlocking));
    HIP_TRY_C(hipHostMalloc((void **)&m->mesh_totals_host, 16 * sizeof(int), hipHostMallocDefault));  // ([8..]: staging of small host values that are copied to the device asynchronously)
    memset(m->mesh_totals_host, 0, 16 * sizeof(int));
    HIP_TRY_C(hipHostGetDevicePointer((void **)&m->mesh_totals_dev, m->mesh_totals_host, 0));
    HIP_TRY_C(hipHostMalloc((void **)&m->mesh_info_host, (size_t)MESH_INFO_PREFETCH * sizeof(JobInfo), hipHostMallocDefault));
    HIP_TRY_C(hipHostGetDevicePointer((void **)&m->mesh_info_dev, m->mesh_info_host, 0));
    static_assert(sizeof(JobInfo) == 8 * sizeof(int), "the triangle kernel moves the records to the host as ints");
    HIP_TRY_C(hipEventCreateWithFlags(&m->call_event, hipEventDisableTiming));
    // A launch set's two events order kernels of ONE device across its streams: the agent-scope release every kernel ends with is all they
    // need.  Without hipEventDisableSystemFence the kernel that carries the event (hipExtLaunchKernelGGL's stop event) ends with a
    // system-scope release -- the L2s written back for the host's sake -- and the kernel behind it starts 5 us later (rocprofv3 timeline
    // of the driver's window: refine -> integrate 5.3 us, integrate -> mesh_count 4.9 us, against 0.1 us between kernels without an event).
    // What the host reads of the device travels through pinned memory behind the kernels' own system-scope fences or through copies.
    // CHISEL_HIP_EVENT_FENCE=system: the events as they were (A/B).
    const char *ev_fence = getenv("CHISEL_HIP_EVENT_FENCE");
    const unsigned set_event_flags = hipEventDisableTiming | ((ev_fence && !strcmp(ev_fence, "system")) ? 0u : (unsigned)hipEventDisableSystemFence);
    for (auto &bs : m->sets) {
        HIP_TRY_C(hipEventCreateWithFlags(&bs.front_done, set_event_flags));
        HIP_TRY_C(hipEventCreateWithFlags(&bs.cull_done, set_event_flags));
        HIP_TRY_C(hipMalloc(&bs.cand_count, COUNT_INTS * sizeof(int)));
        HIP_TRY_C(hipMemsetAsync(bs.cand_count, 0, COUNT_INTS * sizeof(int), m->own_stream));
    }
    for (auto &pr : m->pending_ring) {
        HIP_TRY_C(hipMalloc(&pr, ((size_t)PENDING_CAPACITY + 1) * sizeof(uint64_t)));
        HIP_TRY_C(hipMemsetAsync(pr, 0xff, (size_t)PENDING_CAPACITY * sizeof(uint64_t), m->own_stream));
        HIP_TRY_C(hipMemsetAsync(pr + PENDING_CAPACITY, 0, sizeof(uint64_t), m->own_stream));
    }
    HIP_TRY_C(hipEventCreateWithFlags(&m->mutation_event, hipEventDisableTiming));
    m->mesh_tiny = getenv("CHISEL_HIP_MESH_TINY") != nullptr;
    m->force_uncertain = getenv("CHISEL_HIP_FORCE_UNCERTAIN") != nullptr;
    if (const char *e = getenv("CHISEL_HIP_REFINE")) {
        m->refine_off = atoi(e) == 0;
    }
    if (const char *e = getenv("CHISEL_HIP_VPL")) m->tune.force_vpl = atoi(e) == 2 ? 2 : (atoi(e) == 4 ? 4 : 0);
    if (const char *e = getenv("CHISEL_HIP_CULL_WAVES")) m->tune.force_cull_waves = atoi(e) == 4 ? 4 : (atoi(e) == 1 ? 1 : 16);
    if (const char *e = getenv("CHISEL_HIP_DEFER_TOTALS")) m->tune.defer_totals = atoi(e);
    if (const char *e = getenv("CHISEL_HIP_CULL_CONTIG")) m->tune.force_cull_contig = atoi(e) ? 1 : 0;
    if (const char *e = getenv("CHISEL_HIP_PERSISTENT")) m->tune.persistent_grid = atoi(e) > 0 ? atoi(e) : 0;
    if (const char *e = getenv("CHISEL_HIP_TAIL_PERCENT")) m->tune.tail_percent = atoi(e);
    if (const char *e = getenv("CHISEL_HIP_FINE_BELOW")) m->tune.fine_below = atoi(e);
    m->tune.no_zero_copy = getenv("CHISEL_HIP_NO_ZERO_COPY") != nullptr;
    m->tune.always_wait_packet = getenv("CHISEL_HIP_ALWAYS_WAIT_PACKET") != nullptr;
    // (two runtime calls and two barrier packets fewer per launch set.  Round 4: one rank of eight 404 -> 419 k and 259 -> 265 k frames/s,
    // nothing on the two headline windows -- the shards of a map only; round 5, with the host out of the device's loop: one frame per call
    // 31 -> 37 k, 640x480 @ 2 cm depth only 208 -> 227 k, 16 frames per call 247 -> 254 k, driver's window + 1 %, nothing lost anywhere:
    // every map.  Not for a launch set whose host frames are copied with hipMemcpyAsync: a caller that waits after every frame got them
    // 11 us later, 97 -> 108 us per page-locked frame whose colour image is staged -- BatchSet::staged)
    m->tune.ext_events = true;
    if (const char *e = getenv("CHISEL_HIP_EXT_EVENTS")) m->tune.ext_events = atoi(e) != 0;
    if (const char *e = getenv("CHISEL_HIP_FRONT_POLL_US")) m->tune.front_poll_after_publish_us = atoi(e);
    if (const char *e = getenv("CHISEL_HIP_BRICKS_K1")) m->tune.bricks_for_one_frame = atoi(e) != 0;
    if (const char *e = getenv("CHISEL_HIP_MESH_STAGES")) m->tune.mesh_stage_mask = atoi(e) & 3;
    m->force_pipeline = m->force_uncertain || getenv("CHISEL_HIP_FORCE_PIPELINE") != nullptr;
    {
        const int one = 1;
        HIP_TRY_C(hipMemcpyAsync(m->sets[0].cand_count + COUNT_ONE, &one, sizeof(int), hipMemcpyHostToDevice, m->own_stream));
    }
    MapView &v = m->view;
    v.max_chunks = (int)C;
    v.committed = (int)C0;
    v.hash_mask = hc - 1;
    if (m->growable) {
        // the voxel arrays as reserved address ranges, the first C0 slots mapped (a device without the virtual-memory calls gets a fixed pool)
        hipMemAllocationProp prop;
        memset(&prop, 0, sizeof(prop));
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = dev;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) {
            (void)hipGetLastError();
            m->growable = false;
        } else {
            // (the device reports 4 KiB, but mappings of that size behaved erratically -- hipMemSetAccess on a range behind an earlier mapping
            // failed now and then, tools/micro/vmm_probe.hip --: the pool maps whole 2 MiB pages, as the driver's own allocator does)
            gran = std::max<size_t>(gran, (size_t)2 << 20);
            m->vmm_granularity = gran;
            const size_t per_chunk = (size_t)m->V * sizeof(float);
            const int64_t step = (int64_t)std::max<size_t>(1, gran / per_chunk);
            v.committed = (int)std::min<int64_t>(C, (C0 + step - 1) / step * step);
            const size_t reserve = ((size_t)C * per_chunk + gran - 1) / gran * gran;
            for (int a = 0; a < 3 && m->growable; a++) {
                if (a == 2 && !cfg->use_color) continue;
                void *base = nullptr;
                if (hipMemAddressReserve(&base, reserve, gran, nullptr, 0) != hipSuccess) {
                    (void)hipGetLastError();
                    m->growable = false;
                    break;
                }
                m->pool_mem[a].base = static_cast<char *>(base);
                m->pool_mem[a].reserved = reserve;
                if (pool_map_upto(m, m->pool_mem[a], (size_t)v.committed * per_chunk) != CHISEL_HIP_OK) m->growable = false;
            }
            if (!m->growable)
                for (auto &A : m->pool_mem) pool_release(A);
        }
        if (m->growable) {
            v.sdf = reinterpret_cast<float *>(m->pool_mem[0].base);
            v.wgt = reinterpret_cast<float *>(m->pool_mem[1].base);
            if (cfg->use_color) v.rgbw = reinterpret_cast<uchar4 *>(m->pool_mem[2].base);
        } else {
            // no growth on this device: the pool is its first size, as with max_chunks > 0 (hash and per-slot arrays keep the larger layout)
            v.committed = (int)C0;
        }
    }
    if (!m->growable) {
        HIP_TRY_C(hipMalloc(&v.sdf, (size_t)v.committed * m->V * sizeof(float)));
        HIP_TRY_C(hipMalloc(&v.wgt, (size_t)v.committed * m->V * sizeof(float)));
        if (cfg->use_color) HIP_TRY_C(hipMalloc(&v.rgbw, (size_t)v.committed * m->V * sizeof(uchar4)));
    }
    HIP_TRY_C(hipMalloc(&v.hash_keys, hc * sizeof(uint64_t)));
    HIP_TRY_C(hipMalloc(&v.hash_vals, hc * sizeof(int)));
    HIP_TRY_C(hipMalloc(&v.slot_key, (size_t)C * sizeof(uint64_t)));
    HIP_TRY_C(hipMemsetAsync(v.slot_key, 0xff, (size_t)C * sizeof(uint64_t), m->stream));  // KEY_EMPTY: also the slots a growable pool has not committed yet
    HIP_TRY_C(hipMalloc(&v.slot_dirty, (3 * (size_t)C + SLOT_SUMMARY_PAD) * sizeof(uint32_t)));  // flags, list of dirty slots, its length; sign summaries
    HIP_TRY_C(hipMemsetAsync(v.slot_dirty, 0, (3 * (size_t)C + SLOT_SUMMARY_PAD) * sizeof(uint32_t), m->stream));
    HIP_TRY_C(hipMalloc(&v.free_list, (size_t)C * sizeof(int)));
    HIP_TRY_C(hipMalloc(&v.free_top, sizeof(int)));
    HIP_TRY_C(hipMalloc(&v.counters, 32 * sizeof(unsigned long long)));
    HIP_TRY_C(hipMalloc(&v.block_counters, (size_t)INTEGRATE_MAX_GRID * 32 * sizeof(unsigned long long)));
    HIP_TRY_C(hipMemsetAsync(v.block_counters, 0, (size_t)INTEGRATE_MAX_GRID * 32 * sizeof(unsigned long long), m->stream));
    HIP_TRY_C(hipHostMalloc((void **)&m->error_flag_host, 16 * sizeof(int), hipHostMallocDefault));
    memset(m->error_flag_host, 0, 16 * sizeof(int));
    HIP_TRY_C(hipHostGetDevicePointer((void **)&v.error_flag, m->error_flag_host, 0));
    HIP_TRY_C(hipMemsetAsync(v.counters, 0, CHISEL_HIP_NUM_COUNTERS * sizeof(unsigned long long), m->stream));
    // the mesh recompute's job list, kept by the integration kernels (kernels_map.h: mesh_expand_dirty): a flag per slot, the ids of the
    // listed chunks (twice the pool: a removed chunk leaves its entry behind), the recompute totals + the list's length
    v.mesh_jobs_capacity = (int)std::min<size_t>(2 * C, (size_t)INT32_MAX / 4);
    // (a shard of a sharded map is meshed from a plan's job list, never from the kept one: without the flags mesh_expand_dirty returns at
    // once, and a newly dirtied chunk costs its wave no 27 hash probes)
    if (n_shards == 1) {
        HIP_TRY_C(hipMalloc(&v.mesh_flag, C * sizeof(unsigned)));
        HIP_TRY_C(hipMemsetAsync(v.mesh_flag, 0, C * sizeof(unsigned), m->stream));
    }
    HIP_TRY_C(hipMalloc(&v.mesh_jobs, (size_t)v.mesh_jobs_capacity * 3 * sizeof(int)));
    HIP_TRY_C(hipMalloc(&v.mesh_ctl, MC_INTS * sizeof(int)));
    HIP_TRY_C(hipMemsetAsync(v.mesh_ctl, 0, MC_INTS * sizeof(int), m->stream));
    m->mesh_buf.flags = v.mesh_flag;
    m->mesh_buf.totals = v.mesh_ctl;
    HIP_TRY_C(hipMalloc(&m->view_dev, sizeof(MapView)));
    HIP_TRY_C(hipMemcpyAsync(m->view_dev, &m->view, sizeof(MapView), hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(reset_map_kernel, dim3(2048), dim3(256), 0, m->stream, m->view, m->V, 1);
    HIP_TRY_C(hipGetLastError());
    HIP_TRY_C(hipStreamSynchronize(m->stream));
#undef HIP_TRY_C
    *out = m;
    return CHISEL_HIP_OK;
}

int chisel_hip_create_group(const chisel_hip_config *cfg, const int *device_ids, int n_devices, chisel_hip_map **out) {
    return group::create(cfg, device_ids, n_devices, out);
}

int chisel_hip_destroy(chisel_hip_map *m) {
    if (m && m->is_group) return group::destroy(m);
    if (!m) return CHISEL_HIP_OK;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)sync_all(m);
    MapView &v = m->view;
    if (m->pool_mem[0].base) {  // (a growable pool: mapped ranges, not allocations)
        for (auto &A : m->pool_mem) pool_release(A);
        v.sdf = nullptr; v.wgt = nullptr; v.rgbw = nullptr;
    }
    void *ptrs[] = {v.sdf, v.wgt, v.rgbw, v.hash_keys, v.hash_vals, v.slot_key, v.slot_dirty, v.free_list, v.free_top,
                    v.counters, v.block_counters, m->view_dev, m->scratch_i, m->shell_items_dev, m->shell_offs_dev, m->shell_first_dev, v.mesh_jobs};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (auto &bs : m->sets) {
        void *bp[] = {bs.pyr_data, bs.rec_data, bs.depth_stage, bs.color_stage, bs.boxes, bs.brick_masks, bs.cand_count, bs.items, bs.sync};
        for (void *p : bp)
            if (p) (void)hipFree(p);
        if (bs.front_done) (void)hipEventDestroy(bs.front_done);
        if (bs.cull_done) (void)hipEventDestroy(bs.cull_done);
    }
    for (auto &pr : m->pending_ring)
        if (pr) (void)hipFree(pr);
    for (hipEvent_t e : m->order_events)
        if (e) (void)hipEventDestroy(e);
    if (m->call_event) (void)hipEventDestroy(m->call_event);
    if (m->mutation_event) (void)hipEventDestroy(m->mutation_event);
    if (m->aux && m->aux != m->own_stream) (void)hipStreamDestroy(m->aux);
    if (m->aux2) (void)hipStreamDestroy(m->aux2);
    if (m->aux3) (void)hipStreamDestroy(m->aux3);
    if (m->copy_stream) (void)hipStreamDestroy(m->copy_stream);
    if (m->mesh_totals_host) (void)hipHostFree(m->mesh_totals_host);
    if (m->error_flag_host) (void)hipHostFree(m->error_flag_host);
    if (m->mesh_info_host) (void)hipHostFree(m->mesh_info_host);
    if (m->dirty_tail_host) (void)hipHostFree(m->dirty_tail_host);
    if (m->shell_plan_host) (void)hipHostFree(m->shell_plan_host);
    for (void *p : {(void *)m->shell_plan.jobset, (void *)m->shell_plan.my_jobs, (void *)m->shell_plan.ctl, (void *)m->shell_plan.send_items})
        if (p) (void)hipFree(p);
    clear_meshes(m);
    release_arena_pool(m);
    free_mesh_buffers(m->mesh_buf);
    free_cloud_buffers(m->cloud);
    for (const ProfEvent &p : m->prof_live) {
        (void)hipEventDestroy(p.start);
        (void)hipEventDestroy(p.stop);
    }
    for (hipEvent_t e : m->event_pool) (void)hipEventDestroy(e);
    if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
    delete m;
    return CHISEL_HIP_OK;
}

int chisel_hip_reset(chisel_hip_map *m) {
    if (m && m->is_group) {
        group::forget_recompute(m);  // (a wait-free recompute in flight is void with the maps, and so are the sizes the next one would have gone by)
        return group::for_all(m, [](chisel_hip_map *s) { return chisel_hip_reset(s); });
    }
    if (!m) return fail(CHISEL_HIP_ERR_INVALID, "null map");
    HIP_TRY(hipSetDevice(m->device));
    { m->topology_epoch++; m->dirty_tail_queued = false; }
    hipLaunchKernelGGL(reset_map_kernel, dim3(2048), dim3(256), 0, m->stream, m->view, m->V, 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(note_map_mutation(m));
    clear_meshes(m);
    m->pending_mesh_ids.clear();
    m->dirty_epoch++;
    return CHISEL_HIP_OK;
}

int chisel_hip_set_integrator(chisel_hip_map *m, const chisel_hip_integrator *in) {
    if (m && m->is_group) return group::for_all(m, [&](chisel_hip_map *s) { return chisel_hip_set_integrator(s, in); });
    if (!m || !in) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    if (in->truncator_kind < 0 || in->truncator_kind > 2) return fail(CHISEL_HIP_ERR_INVALID, "unknown truncator kind");
    m->integ = *in;
    return CHISEL_HIP_OK;
}

int chisel_hip_set_stream(chisel_hip_map *m, void *s) {
    SETTLE(m);
    if (m && m->is_group) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "a group runs on the streams of its shards (one set per GPU): order device frames with chisel_hip_wait_event / record_event");
    if (!m) return fail(CHISEL_HIP_ERR_INVALID, "null map");
    int rc = sync_all(m);
    if (rc) return rc;
    m->stream = s ? (hipStream_t)s : m->own_stream;
    return CHISEL_HIP_OK;
}

int chisel_hip_wait_event(chisel_hip_map *m, void *ev) {
    if (m && m->is_group) {  // kept on the group: every launch set of the next integrate call re-arms its shards with it (group::integrate)
        if (!ev) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
        m->input_event = (hipEvent_t)ev;
        return CHISEL_HIP_OK;
    }
    if (!m || !ev) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    m->input_event = (hipEvent_t)ev;
    return CHISEL_HIP_OK;
}

int chisel_hip_record_event(chisel_hip_map *m, void *ev) {
    SETTLE(m);
    if (m && m->is_group) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "one event cannot be recorded on the streams of several GPUs: chisel_hip_synchronize the group instead");
    if (!m || !ev) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipEventRecord((hipEvent_t)ev, m->stream));
    return CHISEL_HIP_OK;
}

// The two orderings a caller with a stream of its own needs around the map (collectives on the communication library's stream, cvids_amd/sharded.py),
// each as ONE call with events the map keeps: `stream` continues after what the map has queued so far / the map's next call starts after what
// `stream` has been given so far.  Nothing is waited for.
int chisel_hip_order_stream_after_map(chisel_hip_map *m, void *stream) {
    SETTLE(m);
    if (m && m->is_group) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "one event cannot be recorded on the streams of several GPUs: chisel_hip_synchronize the group instead");
    if (!m) return fail(CHISEL_HIP_ERR_INVALID, "null map");
    HIP_TRY(hipSetDevice(m->device));
    if (!m->order_events[0]) HIP_TRY(hipEventCreateWithFlags(&m->order_events[0], hipEventDisableTiming));
    HIP_TRY(hipEventRecord(m->order_events[0], m->stream));
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, m->order_events[0], 0));
    return CHISEL_HIP_OK;
}
int chisel_hip_order_map_after_stream(chisel_hip_map *m, void *stream) {
    if (m && m->is_group) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "a group takes an event of the caller's: chisel_hip_wait_event");
    if (!m) return fail(CHISEL_HIP_ERR_INVALID, "null map");
    HIP_TRY(hipSetDevice(m->device));
    if (!m->order_events[1]) HIP_TRY(hipEventCreateWithFlags(&m->order_events[1], hipEventDisableTiming));
    HIP_TRY(hipEventRecord(m->order_events[1], (hipStream_t)stream));
    m->input_event = m->order_events[1];  // (as chisel_hip_wait_event: the next call that queues work makes its streams wait)
    return CHISEL_HIP_OK;
}

int chisel_hip_synchronize(chisel_hip_map *m) {
    SETTLE(m);
    if (m && m->is_group) return group::for_all(m, [](chisel_hip_map *s) { return chisel_hip_synchronize(s); });
    if (!m) return fail(CHISEL_HIP_ERR_INVALID, "null map");
    HIP_TRY(hipSetDevice(m->device));
    {
        int rc_m = check_mesh_totals(m);
        if (rc_m) return rc_m;
    }
    return check_device_error(m);
}

int chisel_hip_integrate_depth(chisel_hip_map *m, const chisel_hip_depth_frame *f) {
    if (m && m->is_group) return group::integrate(m, 1, f, nullptr);
    if (!m) return fail(CHISEL_HIP_ERR_INVALID, "null map");
    return integrate_frames(m, 1, f, nullptr, 1);
}

int chisel_hip_integrate_depth_color(chisel_hip_map *m, const chisel_hip_depth_frame *f, const chisel_hip_color_frame *c) {
    if (m && m->is_group) return group::integrate(m, 1, f, c);
    if (!m || !c) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    return integrate_frames(m, 1, f, c, 1);
}

int chisel_hip_integrate_batch(chisel_hip_map *m, int n, const chisel_hip_depth_frame *frames,
                               const chisel_hip_color_frame *colors) {
    if (m && m->is_group) return group::integrate(m, n, frames, colors);
    if (!m || n < 0 || (n > 0 && !frames)) return fail(CHISEL_HIP_ERR_INVALID, "bad batch");
    return integrate_frames(m, n, frames, colors, m->batch_frames);
}

// ProjectionIntegrator::Integrate / IntegrateColor (ProjectionIntegrator.h:51-52, 101-102) are per-chunk calls: one frame into ONE
// chunk, the return value "some voxel changed".  The chunk must be resident (the reference's caller holds a Chunk object); the frame
// goes through the ordinary launch set with the candidate range pinned to that id.
int chisel_hip_integrate_chunk(chisel_hip_map *m, const int id[3], const chisel_hip_depth_frame *f, const chisel_hip_color_frame *c, int *updated) {
    SETTLE(m);
    if (m && m->is_group) return id ? chisel_hip_integrate_chunk(group::owner_map(m, id), id, f, c, updated) : fail(CHISEL_HIP_ERR_INVALID, "null argument");
    if (!m || !id || !f) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    if (chunk_owner(id[0], id[1], id[2], m->cfg.n_shards, m->cfg.shard_block) != m->cfg.shard_rank) return fail(CHISEL_HIP_ERR_INVALID, "this shard does not own the chunk");
    int has = 0;
    int rc = chisel_hip_has_chunk(m, id, &has);
    if (rc) return rc;
    if (!has) return fail(CHISEL_HIP_ERR_NOT_FOUND, "chunk not resident");
    uint64_t before[CHISEL_HIP_NUM_COUNTERS], after[CHISEL_HIP_NUM_COUNTERS];
    rc = chisel_hip_get_counters(m, before, 0);
    if (rc) return rc;
    m->single_chunk = true;
    for (int a = 0; a < 3; a++) m->single_id[a] = id[a];
    rc = integrate_frames(m, 1, f, c, 1);
    m->single_chunk = false;
    if (rc) return rc;
    rc = chisel_hip_synchronize(m);
    if (rc) return rc;
    rc = chisel_hip_get_counters(m, after, 0);
    if (rc) return rc;
    if (updated) *updated = after[CHISEL_HIP_CNT_UPDATED_CHUNKS] != before[CHISEL_HIP_CNT_UPDATED_CHUNKS] ? 1 : 0;
    return CHISEL_HIP_OK;
}

int chisel_hip_garbage_collect(chisel_hip_map *m, const int *ids, int n) {
    SETTLE(m);
    if (m && m->is_group) return group::garbage_collect(m, ids, n);
    if (!m || n < 0 || (n > 0 && !ids)) return fail(CHISEL_HIP_ERR_INVALID, "bad id list");
    if (n == 0) return CHISEL_HIP_OK;
    HIP_TRY(hipSetDevice(m->device));
    { m->topology_epoch++; m->dirty_tail_queued = false; }
    {
        int rc_m = check_mesh_totals(m);  // a recompute in flight reads the voxels as they are
        if (rc_m) return rc_m;
    }
    // keep the reference's meshesToUpdate entries of chunks that disappear (Chisel.h:228 lives on the host)
    std::vector<int> dirty;
    int rc = fetch_listed(m, true, dirty, nullptr);
    if (rc) return rc;
    if (!dirty.empty()) {
        std::unordered_set<uint64_t, IdHash> doomed;
        for (int i = 0; i < n; i++) doomed.insert(pack_id(ids[3 * i], ids[3 * i + 1], ids[3 * i + 2]));
        std::vector<int> gone;
        for (size_t i = 0; i + 2 < dirty.size(); i += 3)
            if (doomed.count(pack_id(dirty[i], dirty[i + 1], dirty[i + 2]))) {
                gone.push_back(dirty[i]); gone.push_back(dirty[i + 1]); gone.push_back(dirty[i + 2]);
            }
        expand27(gone, m->pending_mesh_ids);
        if (!gone.empty()) m->pending_version++;
    }
    rc = ensure_scratch(m, (size_t)n * 3 + 16);
    if (rc) return rc;
    int *d_cnt = m->scratch_i, *d_ids = m->scratch_i + 16;
    HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(int), m->stream));
    HIP_TRY(hipMemcpyAsync(d_ids, ids, (size_t)n * 3 * sizeof(int), hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(remove_chunks_kernel, dim3(n), dim3(256), 0, m->stream, m->view, d_ids, n, d_cnt, m->V);
    HIP_TRY(hipGetLastError());
    // every removed chunk may leave a dead entry in the job list the integration kernels keep (its slot can be listed again): long
    // before the list could run full it is dropped and rebuilt from the dirty flags at the next recompute
    m->removed_since_recompute += n;
    if (m->removed_since_recompute >= (int64_t)m->view.max_chunks / 2) give_up_job_list(m);
    HIP_TRY(hipStreamSynchronize(m->stream));
    return CHISEL_HIP_OK;
}

int chisel_hip_num_chunks(chisel_hip_map *m, int64_t *out) {
    SETTLE(m);
    if (m && m->is_group) {
        if (!out) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
        *out = 0;
        return group::for_all(m, [&](chisel_hip_map *s) {
            int64_t n = 0;
            const int rc = chisel_hip_num_chunks(s, &n);
            *out += n;
            return rc;
        });
    }
    if (!m || !out) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(m->device));
    int rc = check_device_error(m);
    if (rc) return rc;
    int top = 0;
    HIP_TRY(hipMemcpy(&top, m->view.free_top, sizeof(int), hipMemcpyDeviceToHost));
    *out = (int64_t)m->view.committed - top;
    return CHISEL_HIP_OK;
}

int chisel_hip_list_chunks(chisel_hip_map *m, int *ids, int64_t max_ids, int64_t *count) {
    SETTLE(m);
    if (m && m->is_group) {
        std::vector<int> all;
        const int rc = group::gather_ids(m, chisel_hip_list_chunks, false, all);
        return rc ? rc : group::emit_ids(all, ids, max_ids, count);
    }
    if (!m || !count) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(m->device));
    int rc = check_device_error(m);
    if (rc) return rc;
    std::vector<int> all;
    rc = fetch_listed(m, false, all, nullptr);
    if (rc) return rc;
    *count = (int64_t)all.size() / 3;
    if (ids && max_ids > 0) {
        // ascending (x, then y, then z), like a group's listing and the mesh listings: the device lists in the order its atomics came in,
        // and a caller that acts on "every n-th chunk" (tools/soak.py) should get the same chunks every time.  Keys with x in the high
        // bits, least-significant-digit radix sort in 11-bit digits, digits that are the same in every key skipped (ids span a few
        // hundred per axis: three or four passes; 10 000 chunks in about 0.1 ms, a comparison sort of the triples takes ten times that)
        const size_t n = (size_t)*count;
        std::vector<uint64_t> a(n), b(n);
        uint64_t all_or = 0, all_and = ~0ull;
        for (size_t j = 0; j < n; j++) {
            a[j] = ((uint64_t)(uint32_t)(all[3 * j] + ID_BIAS) << 42) | ((uint64_t)(uint32_t)(all[3 * j + 1] + ID_BIAS) << 21) | (uint64_t)(uint32_t)(all[3 * j + 2] + ID_BIAS);
            all_or |= a[j];
            all_and &= a[j];
        }
        const uint64_t varying = all_or ^ all_and;
        for (int sh = 0; sh < 63; sh += 11) {
            if (((varying >> sh) & 2047ull) == 0) continue;
            unsigned cnt[2049] = {0};
            for (uint64_t x : a) cnt[((x >> sh) & 2047ull) + 1]++;
            for (int i = 0; i < 2048; i++) cnt[i + 1] += cnt[i];
            for (uint64_t x : a) b[cnt[(x >> sh) & 2047ull]++] = x;
            a.swap(b);
        }
        for (size_t j = 0; j < n && (int64_t)j < max_ids; j++) {
            ids[3 * j] = (int)((a[j] >> 42) & 0x1FFFFF) - ID_BIAS;
            ids[3 * j + 1] = (int)((a[j] >> 21) & 0x1FFFFF) - ID_BIAS;
            ids[3 * j + 2] = (int)(a[j] & 0x1FFFFF) - ID_BIAS;
        }
    }
    return CHISEL_HIP_OK;
}

int chisel_hip_has_chunk(chisel_hip_map *m, const int id[3], int *out) {
    SETTLE(m);
    if (m && m->is_group) return id ? chisel_hip_has_chunk(group::owner_map(m, id), id, out) : fail(CHISEL_HIP_ERR_INVALID, "null argument");
    if (!m || !id || !out) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(m->device));
    std::vector<int> slots;
    int rc = lookup_slots(m, id, 1, slots);
    if (rc) return rc;
    *out = slots[0] >= 0 ? 1 : 0;
    return CHISEL_HIP_OK;
}

int chisel_hip_download_chunk(chisel_hip_map *m, const int id[3], float *sdf, float *weight, uint8_t *rgbw) {
    SETTLE(m);
    if (m && m->is_group) return id ? chisel_hip_download_chunk(group::owner_map(m, id), id, sdf, weight, rgbw) : fail(CHISEL_HIP_ERR_INVALID, "null argument");
    if (!m || !id) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(m->device));
    int rc = check_device_error(m);
    if (rc) return rc;
    std::vector<int> slots;
    rc = lookup_slots(m, id, 1, slots);
    if (rc) return rc;
    if (slots[0] < 0) return fail(CHISEL_HIP_ERR_NOT_FOUND, "chunk not resident (ChunkManager::GetChunk would throw std::out_of_range)");
    const size_t off = (size_t)slots[0] * m->V;
    if (sdf) HIP_TRY(hipMemcpy(sdf, m->view.sdf + off, (size_t)m->V * sizeof(float), hipMemcpyDeviceToHost));
    if (weight) HIP_TRY(hipMemcpy(weight, m->view.wgt + off, (size_t)m->V * sizeof(float), hipMemcpyDeviceToHost));
    if (rgbw) {
        if (!m->view.rgbw) return fail(CHISEL_HIP_ERR_INVALID, "map has no colour voxels");
        HIP_TRY(hipMemcpy(rgbw, m->view.rgbw + off, (size_t)m->V * 4, hipMemcpyDeviceToHost));
    }
    return CHISEL_HIP_OK;
}

namespace {
// device staging for the batched chunk transfers: ids, flags and (host callers) the voxel rows
struct ChunkStage {
    int *ids = nullptr, *flags = nullptr;
    float *sdf = nullptr, *wgt = nullptr;
    uchar4 *col = nullptr;
    ~ChunkStage() {
        for (void *p : {(void *)ids, (void *)flags, (void *)sdf, (void *)wgt, (void *)col})
            if (p) (void)hipFree(p);
    }
};
}  // namespace

int chisel_hip_condition_depth(const double *src, int w0, int h0, int src_on_device, float *dst, int w, int h, int dst_on_device, double K[4],
                               void *hip_stream) {
    if (!src || !dst || w0 <= 0 || h0 <= 0 || w <= 0 || h <= 0) return fail(CHISEL_HIP_ERR_INVALID, "bad argument");
    hipStream_t st = (hipStream_t)hip_stream;
    const size_t n0 = (size_t)w0 * h0, n1 = (size_t)w * h;
    double *d_src = nullptr;
    float *d_dst = nullptr;
    const double *in = src;
    float *out = dst;
    if (!src_on_device) {
        HIP_TRY(hipMalloc(&d_src, n0 * sizeof(double)));
        HIP_TRY(hipMemcpyAsync(d_src, src, n0 * sizeof(double), hipMemcpyHostToDevice, st));
        in = d_src;
    }
    if (!dst_on_device) {
        HIP_TRY(hipMalloc(&d_dst, n1 * sizeof(float)));
        out = d_dst;
    }
    hipLaunchKernelGGL(condition_depth_kernel, dim3((w + 255) / 256, h), dim3(256), 0, st, in, w0, h0, out, w, h);
    HIP_TRY(hipGetLastError());
    if (!dst_on_device) HIP_TRY(hipMemcpyAsync(dst, d_dst, n1 * sizeof(float), hipMemcpyDeviceToHost, st));
    if (!src_on_device || !dst_on_device) {
        HIP_TRY(hipStreamSynchronize(st));
        if (d_src) (void)hipFree(d_src);
        if (d_dst) (void)hipFree(d_dst);
    }
    if (K) {  // collaborative_server_system.cpp:216-219
        K[0] = K[0] / (double)w0 * (double)w;
        K[2] = K[2] / (double)w0 * (double)w;
        K[1] = K[1] / (double)h0 * (double)h;
        K[3] = K[3] / (double)h0 * (double)h;
    }
    return CHISEL_HIP_OK;
}

int chisel_hip_condition_color(const uint8_t *src, int w0, int h0, int channels, int src_on_device, uint8_t *dst, int w, int h,
                               int dst_on_device, void *hip_stream) {
    if (!src || !dst || w0 <= 0 || h0 <= 0 || w <= 0 || h <= 0 || (channels != 1 && channels != 3 && channels != 4))
        return fail(CHISEL_HIP_ERR_INVALID, "bad argument");
    if ((size_t)w0 * h0 * channels > ((size_t)1 << 31) || (size_t)w * h * channels > ((size_t)1 << 31))
        return fail(CHISEL_HIP_ERR_INVALID, "image too large");
    hipStream_t st = (hipStream_t)hip_stream;
    const size_t n0 = (size_t)w0 * h0 * channels, n1 = (size_t)w * h * channels;
    uint8_t *d_src = nullptr, *d_dst = nullptr;
    const uint8_t *in = src;
    uint8_t *out = dst;
    if (!src_on_device) {
        HIP_TRY(hipMalloc(&d_src, n0));
        HIP_TRY(hipMemcpyAsync(d_src, src, n0, hipMemcpyHostToDevice, st));
        in = d_src;
    }
    if (!dst_on_device) {
        HIP_TRY(hipMalloc(&d_dst, n1));
        out = d_dst;
    }
    hipLaunchKernelGGL(condition_color_kernel, dim3((w * channels + 255) / 256, h), dim3(256), 0, st, in, w0, h0, channels, out, w, h);
    HIP_TRY(hipGetLastError());
    if (!dst_on_device) HIP_TRY(hipMemcpyAsync(dst, d_dst, n1, hipMemcpyDeviceToHost, st));
    if (!src_on_device || !dst_on_device) {
        HIP_TRY(hipStreamSynchronize(st));
        if (d_src) (void)hipFree(d_src);
        if (d_dst) (void)hipFree(d_dst);
    }
    return CHISEL_HIP_OK;
}

int chisel_hip_publish_cloud(const double *depth, const uint8_t *color, int w, int h, int color_step, int src_on_device, void *points,
                             int dst_on_device, void *hip_stream) {
    if (!depth || !color || !points || w <= 0 || h <= 0 || color_step < w) return fail(CHISEL_HIP_ERR_INVALID, "bad argument");
    hipStream_t st = (hipStream_t)hip_stream;
    const size_t npx = (size_t)w * h, cbytes = (size_t)color_step * h;
    double *d_depth = nullptr;
    uint8_t *d_color = nullptr;
    uint4 *d_pts = nullptr;
    const double *in_d = depth;
    const uint8_t *in_c = color;
    uint4 *out = static_cast<uint4 *>(points);
    if (!src_on_device) {
        HIP_TRY(hipMalloc(&d_depth, npx * sizeof(double)));
        HIP_TRY(hipMalloc(&d_color, cbytes));
        HIP_TRY(hipMemcpyAsync(d_depth, depth, npx * sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_color, color, cbytes, hipMemcpyHostToDevice, st));
        in_d = d_depth;
        in_c = d_color;
    }
    if (!dst_on_device) {
        HIP_TRY(hipMalloc(&d_pts, npx * sizeof(uint4)));
        out = d_pts;
    }
    hipLaunchKernelGGL(publish_cloud_kernel, dim3((w + 255) / 256, h), dim3(256), 0, st, in_d, in_c, w, h, color_step, out);
    HIP_TRY(hipGetLastError());
    if (!dst_on_device) HIP_TRY(hipMemcpyAsync(points, d_pts, npx * sizeof(uint4), hipMemcpyDeviceToHost, st));
    if (!src_on_device || !dst_on_device) {
        HIP_TRY(hipStreamSynchronize(st));
        if (d_depth) (void)hipFree(d_depth);
        if (d_color) (void)hipFree(d_color);
        if (d_pts) (void)hipFree(d_pts);
    }
    return CHISEL_HIP_OK;
}

// ---- DepthFilter (depth_filter.cpp) ----------------------------------------------------------------------------------------
struct chisel_hip_depth_filter {
    int device = 0;
    int height = 0, width = 0;
    FilterView view{};
    double *stage_mu = nullptr, *stage_cov = nullptr, *stage_out = nullptr;  // host arrays pass through these
};
int chisel_hip_depth_filter_create(int height, int width, int device_id, chisel_hip_depth_filter **out) {
    if (!out || height <= 0 || width <= 0 || (int64_t)height * width > (1 << 28)) return fail(CHISEL_HIP_ERR_INVALID, "bad filter size");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(CHISEL_HIP_ERR_HIP, "no HIP device (there is no CPU path)");
    if (device_id < 0) (void)hipGetDevice(&device_id);
    if (device_id >= n_dev) return fail(CHISEL_HIP_ERR_INVALID, "bad device id");
    HIP_TRY(hipSetDevice(device_id));
    chisel_hip_depth_filter *f = new chisel_hip_depth_filter();
    f->device = device_id; f->height = height; f->width = width;
    const size_t n = (size_t)height * width;
    f->view.n = (int)n;
    f->view.inv_depth_range = 100 - 0.01;  // m_nMaxInvDepth - m_nMinInvDepth, depth_filter.cpp:138-141
    double **arrays[] = {&f->view.a, &f->view.b, &f->view.mu, &f->view.cov, &f->stage_mu, &f->stage_cov, &f->stage_out};
    for (double **p : arrays)
        if (hipMalloc(p, n * sizeof(double)) != hipSuccess) {
            chisel_hip_depth_filter_destroy(f);
            return fail(CHISEL_HIP_ERR_HIP, "hipMalloc failed");
        }
    hipLaunchKernelGGL(filter_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, f->view);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    *out = f;
    return CHISEL_HIP_OK;
}
int chisel_hip_depth_filter_destroy(chisel_hip_depth_filter *f) {
    if (!f) return CHISEL_HIP_OK;
    (void)hipSetDevice(f->device);
    (void)hipDeviceSynchronize();
    void *ptrs[] = {f->view.a, f->view.b, f->view.mu, f->view.cov, f->stage_mu, f->stage_cov, f->stage_out};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete f;
    return CHISEL_HIP_OK;
}
int chisel_hip_depth_filter_update(chisel_hip_depth_filter *f, const double *mu, const double *cov, double cov_all, int reciprocal,
                                   int on_device) {
    if (!f || !mu) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(f->device));
    const size_t n = (size_t)f->view.n;
    const double *d_mu = mu, *d_cov = cov;
    if (!on_device) {
        HIP_TRY(hipMemcpyAsync(f->stage_mu, mu, n * sizeof(double), hipMemcpyHostToDevice, 0));
        d_mu = f->stage_mu;
        if (cov) {
            HIP_TRY(hipMemcpyAsync(f->stage_cov, cov, n * sizeof(double), hipMemcpyHostToDevice, 0));
            d_cov = f->stage_cov;
        }
    }
    hipLaunchKernelGGL(filter_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, f->view, d_mu, d_cov, cov_all, reciprocal);
    HIP_TRY(hipGetLastError());
    return CHISEL_HIP_OK;  // stream 0: ordered against the next call; reads wait
}
int chisel_hip_depth_filter_read(chisel_hip_depth_filter *f, int which, double *dst, int dst_on_device) {
    if (!f || !dst || which < 0 || which > 6) return fail(CHISEL_HIP_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(f->device));
    const size_t n = (size_t)f->view.n;
    double *d_out = dst_on_device ? dst : f->stage_out;
    hipLaunchKernelGGL(filter_read_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, f->view, which, d_out);
    HIP_TRY(hipGetLastError());
    if (!dst_on_device) HIP_TRY(hipMemcpyAsync(dst, d_out, n * sizeof(double), hipMemcpyDeviceToHost, 0));
    HIP_TRY(hipStreamSynchronize(0));
    return CHISEL_HIP_OK;
}

int chisel_hip_export_chunks(chisel_hip_map *m, const int *ids, int n, float *sdf, float *weight, uint8_t *rgbw, int *found, int on_device) {
    SETTLE(m);
    if (m && m->is_group) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "chisel_hip_export_chunks is a call between the shards of a map: a group makes it itself (chisel_hip_update_meshes)");
    if (!m || n < 0 || (n > 0 && (!ids || !sdf || !weight || !found))) return fail(CHISEL_HIP_ERR_INVALID, "bad argument");
    if (n == 0) return CHISEL_HIP_OK;
    HIP_TRY(hipSetDevice(m->device));
    const size_t V = (size_t)m->V, rows = (size_t)n * V;
    const bool color = m->view.rgbw && rgbw;
    ChunkStage st;
    HIP_TRY(hipMalloc(&st.ids, (size_t)n * 3 * sizeof(int)));
    HIP_TRY(hipMalloc(&st.flags, (size_t)n * sizeof(int)));
    HIP_TRY(hipMemcpyAsync(st.ids, ids, (size_t)n * 3 * sizeof(int), hipMemcpyHostToDevice, m->stream));
    float *d_s = sdf, *d_w = weight;
    uchar4 *d_c = reinterpret_cast<uchar4 *>(rgbw);
    if (!on_device) {
        HIP_TRY(hipMalloc(&st.sdf, rows * sizeof(float)));
        HIP_TRY(hipMalloc(&st.wgt, rows * sizeof(float)));
        if (color) HIP_TRY(hipMalloc(&st.col, rows * sizeof(uchar4)));
        d_s = st.sdf; d_w = st.wgt; d_c = st.col;
    }
    hipLaunchKernelGGL(export_chunks_kernel, dim3(n), dim3(256), 0, m->stream, m->view, st.ids, m->V, d_s, d_w, color ? d_c : nullptr, st.flags);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(found, st.flags, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, m->stream));
    if (!on_device) {
        HIP_TRY(hipMemcpyAsync(sdf, st.sdf, rows * sizeof(float), hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipMemcpyAsync(weight, st.wgt, rows * sizeof(float), hipMemcpyDeviceToHost, m->stream));
        if (color) HIP_TRY(hipMemcpyAsync(rgbw, st.col, rows * sizeof(uchar4), hipMemcpyDeviceToHost, m->stream));
    }
    HIP_TRY(hipStreamSynchronize(m->stream));
    return CHISEL_HIP_OK;
}

int chisel_hip_import_ghost_chunks(chisel_hip_map *m, const int *ids, int n, const float *sdf, const float *weight, const uint8_t *rgbw,
                                   const int *found, int on_device) {
    SETTLE(m);
    if (m && m->is_group) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "chisel_hip_import_ghost_chunks is a call between the shards of a map: a group makes it itself (chisel_hip_update_meshes)");
    if (!m || n < 0 || (n > 0 && (!ids || !sdf || !weight))) return fail(CHISEL_HIP_ERR_INVALID, "bad argument");
    if (n == 0) return CHISEL_HIP_OK;
    HIP_TRY(hipSetDevice(m->device));
    { m->topology_epoch++; m->dirty_tail_queued = false; }
    int rc = check_mesh_totals(m);
    if (rc) return rc;
    rc = maybe_grow(m, n);  // (ghosts take slots of this shard's pool until they are dropped again)
    if (rc) return rc;
    for (int j = 0; j < n; j++)
        if ((!found || found[j]) && chunk_owner(ids[3 * j], ids[3 * j + 1], ids[3 * j + 2], m->cfg.n_shards, m->cfg.shard_block) == m->cfg.shard_rank)
            return fail(CHISEL_HIP_ERR_INVALID, "a ghost chunk must belong to another shard");
    const size_t V = (size_t)m->V, rows = (size_t)n * V;
    ChunkStage st;
    HIP_TRY(hipMalloc(&st.ids, (size_t)n * 3 * sizeof(int)));
    HIP_TRY(hipMemcpyAsync(st.ids, ids, (size_t)n * 3 * sizeof(int), hipMemcpyHostToDevice, m->stream));
    if (found) {
        HIP_TRY(hipMalloc(&st.flags, (size_t)n * sizeof(int)));
        HIP_TRY(hipMemcpyAsync(st.flags, found, (size_t)n * sizeof(int), hipMemcpyHostToDevice, m->stream));
    }
    const float *d_s = sdf, *d_w = weight;
    const uchar4 *d_c = reinterpret_cast<const uchar4 *>(rgbw);
    if (!on_device) {
        HIP_TRY(hipMalloc(&st.sdf, rows * sizeof(float)));
        HIP_TRY(hipMalloc(&st.wgt, rows * sizeof(float)));
        HIP_TRY(hipMemcpyAsync(st.sdf, sdf, rows * sizeof(float), hipMemcpyHostToDevice, m->stream));
        HIP_TRY(hipMemcpyAsync(st.wgt, weight, rows * sizeof(float), hipMemcpyHostToDevice, m->stream));
        if (rgbw) {
            HIP_TRY(hipMalloc(&st.col, rows * sizeof(uchar4)));
            HIP_TRY(hipMemcpyAsync(st.col, rgbw, rows * sizeof(uchar4), hipMemcpyHostToDevice, m->stream));
        }
        d_s = st.sdf; d_w = st.wgt; d_c = st.col;
    }
    hipLaunchKernelGGL(import_chunks_kernel, dim3(n), dim3(256), 0, m->stream, m->view, st.ids, found ? st.flags : nullptr, m->V, d_s, d_w, d_c);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(m->stream));
    for (int j = 0; j < n; j++)
        if (!found || found[j]) m->ghost_ids.insert(m->ghost_ids.end(), ids + 3 * j, ids + 3 * j + 3);
    return check_device_error(m);
}

int chisel_hip_drop_ghost_chunks(chisel_hip_map *m) {
    SETTLE(m);
    if (m && m->is_group) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "chisel_hip_drop_ghost_chunks is a call between the shards of a map: a group makes it itself (chisel_hip_update_meshes)");
    if (!m) return fail(CHISEL_HIP_ERR_INVALID, "null map");
    if (m->ghost_packed && m->shell_fixed_ghosts) {
        // the wait-free form: nothing is waited for.  The recompute in flight may have to be emitted again (its totals are not known yet):
        // then the kernel leaves the ghosts where they are (MC_LATCH) and check_mesh_totals launches it again behind the second emission.
        HIP_TRY(hipSetDevice(m->device));
        { m->topology_epoch++; m->dirty_tail_queued = false; }
        launch_fixed_drop(m, m->view.mesh_ctl ? m->view.mesh_ctl + MC_LATCH : nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(note_map_mutation(m));
        m->shell_redrop = m->pending_meshes.unchecked;  // (ghost_packed stays: the buffer is the caller's until its next recompute)
        m->shell_fixed_ghosts = false;
        if (!m->shell_redrop) m->ghost_packed = nullptr;
    } else if (m->ghost_packed && !m->shell_redrop) {
        // the ghosts of chisel_hip_import_shells_packed: named by the items of the received segments, which the caller still holds
        HIP_TRY(hipSetDevice(m->device));
        { m->topology_epoch++; m->dirty_tail_queued = false; }
        int rc_p = check_mesh_totals(m);  // a recompute in flight may still read them
        if (rc_p) return rc_p;
        if (m->ghost_packed_items > 0) {  // (the two passes of kernels_map.h, over segments that lie back to back)
            hipLaunchKernelGGL(shell_reset_boxes_kernel, dim3((unsigned)m->ghost_packed_items), dim3(256), 0, m->stream, m->view, m->ghost_packed, 0ll, m->ghost_segments, m->cfg.n_shards,
                               m->N, (const int *)nullptr, (const int *)nullptr);
            hipLaunchKernelGGL(shell_remove_ghosts_kernel, dim3((unsigned)(m->ghost_packed_items + 255) / 256), dim3(256), 0, m->stream, m->view, m->ghost_packed, 0ll, m->ghost_segments,
                               m->cfg.n_shards, (const int *)nullptr, (const int *)nullptr);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(note_map_mutation(m));
        m->ghost_packed = nullptr;
        m->ghost_packed_items = 0;
    }
    if (m->ghost_ids.empty()) return CHISEL_HIP_OK;
    HIP_TRY(hipSetDevice(m->device));
    { m->topology_epoch++; m->dirty_tail_queued = false; }
    int rc = check_mesh_totals(m);  // a recompute in flight may still read them
    if (rc) return rc;
    const int n = (int)(m->ghost_ids.size() / 3);
    rc = ensure_scratch(m, (size_t)n * 3 + 16);
    if (rc) return rc;
    int *d_cnt = m->scratch_i, *d_ids = m->scratch_i + 16;
    HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(int), m->stream));
    HIP_TRY(hipMemcpyAsync(d_ids, m->ghost_ids.data(), (size_t)n * 3 * sizeof(int), hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(remove_chunks_kernel, dim3(n), dim3(256), 0, m->stream, m->view, d_ids, n, d_cnt, m->V);
    HIP_TRY(hipGetLastError());
    HIP_TRY(note_map_mutation(m));  // (stream-ordered; the next batch's front half waits for it -- nothing is waited for here)
    m->ghost_ids.clear();
    return CHISEL_HIP_OK;
}


// ---- meshing a sharded map: shells (kernels_map.h) -------------------------------------------------------------------------------
namespace {
// items (x, y, z, box) and their payload offsets onto the device (the map's stream; staging buffers kept by the map)
int stage_shell_items(chisel_hip_map *m, const int *items, int n, long long *total) {
    if (n > m->shell_capacity) {
        HIP_TRY(hipStreamSynchronize(m->stream));
        if (m->shell_items_dev) HIP_TRY(hipFree(m->shell_items_dev));
        if (m->shell_offs_dev) HIP_TRY(hipFree(m->shell_offs_dev));
        m->shell_items_dev = nullptr;
        m->shell_offs_dev = nullptr;
        const int cap = std::max(4096, 2 * n);
        HIP_TRY(hipMalloc(&m->shell_items_dev, (size_t)cap * 4 * sizeof(int)));
        HIP_TRY(hipMalloc(&m->shell_offs_dev, (size_t)cap * sizeof(long long)));
        m->shell_capacity = cap;
    }
    std::vector<long long> offs((size_t)n);
    long long t = 0;
    for (int j = 0; j < n; j++) {
        if (items[4 * j + 3] < 0 || items[4 * j + 3] > 63) return fail(CHISEL_HIP_ERR_INVALID, "bad shell box code");
        offs[(size_t)j] = t;
        t += shell_volume(items[4 * j + 3], m->N);
    }
    *total = t;
    HIP_TRY(hipMemcpyAsync(m->shell_items_dev, items, (size_t)n * 4 * sizeof(int), hipMemcpyHostToDevice, m->stream));
    HIP_TRY(hipMemcpyAsync(m->shell_offs_dev, offs.data(), (size_t)n * sizeof(long long), hipMemcpyHostToDevice, m->stream));
    return CHISEL_HIP_OK;
}
}  // namespace

int chisel_hip_dirty_ids_device(chisel_hip_map *m, int *out_dev, int capacity) {
    SETTLE(m);
    if (m && m->is_group) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "a call between the shards of a map");
    if (!m || !out_dev || capacity < 0) return fail(CHISEL_HIP_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(m->device));
    // meshesToUpdate entries kept on the host (27-neighbourhoods of chunks removed while dirty) come first, as entries that are not
    // expanded again (flag 1); the kernel appends the dirty chunks (flag 0) behind them
    if (m->input_event) {  // chisel_hip_wait_event: the caller's buffer is ready behind this event (its allocator's stream, not the map's)
        HIP_TRY(hipStreamWaitEvent(m->stream, m->input_event, 0));
        m->input_event = nullptr;
    }
    // out[0] is the TRUE number of entries -- host-held ones and the kernel's appends -- also when it exceeds `capacity` (only the first
    // `capacity` are stored): a caller that reads out[0] > capacity grows its buffer and asks again; a count capped at the capacity
    // would have looked like a list that just fits, and the entries beyond it would have been dropped for good
    std::vector<int> head(1, (int)std::min<size_t>(m->pending_mesh_ids.size(), (size_t)INT32_MAX / 8));
    int stored = 0;
    for (uint64_t key : m->pending_mesh_ids) {
        if (stored >= capacity) break;
        int x, y, z;
        unpack_id(key, x, y, z);
        head.push_back(x); head.push_back(y); head.push_back(z); head.push_back(1);
        stored++;
    }
    if (head.size() == 1 && head[0] == 0) HIP_TRY(hipMemsetAsync(out_dev, 0, sizeof(int), m->stream));  // (the common case: no copy from pageable memory)
    else HIP_TRY(hipMemcpyAsync(out_dev, head.data(), head.size() * sizeof(int), hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(list_dirty_ids_kernel, dim3(256), dim3(256), 0, m->stream, m->view, out_dev, capacity);
    HIP_TRY(hipGetLastError());
    return CHISEL_HIP_OK;
}

int chisel_hip_export_shells(chisel_hip_map *m, const int *items, int n, float *sdf, float *weight, uint8_t *rgbw, int *found, int on_device) {
    SETTLE(m);
    if (m && m->is_group) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "a call between the shards of a map");
    if (!m || n < 0 || (n > 0 && (!items || !sdf || !weight || !found))) return fail(CHISEL_HIP_ERR_INVALID, "bad argument");
    if (n == 0) return CHISEL_HIP_OK;
    HIP_TRY(hipSetDevice(m->device));
    {
        int rc_m = check_mesh_totals(m);
        if (rc_m) return rc_m;
    }
    if (m->input_event) {  // chisel_hip_wait_event: the output buffers may be used from here on
        HIP_TRY(hipStreamWaitEvent(m->stream, m->input_event, 0));
        m->input_event = nullptr;
    }
    long long total = 0;
    int rc = stage_shell_items(m, items, n, &total);
    if (rc) return rc;
    const bool color = m->view.rgbw && rgbw;
    ChunkStage st;
    float *d_s = sdf, *d_w = weight;
    uchar4 *d_c = reinterpret_cast<uchar4 *>(rgbw);
    int *d_f = found;
    if (!on_device) {
        HIP_TRY(hipMalloc(&st.sdf, (size_t)total * sizeof(float)));
        HIP_TRY(hipMalloc(&st.wgt, (size_t)total * sizeof(float)));
        if (color) HIP_TRY(hipMalloc(&st.col, (size_t)total * sizeof(uchar4)));
        HIP_TRY(hipMalloc(&st.flags, (size_t)n * sizeof(int)));
        d_s = st.sdf; d_w = st.wgt; d_c = st.col; d_f = st.flags;
    }
    hipLaunchKernelGGL(export_shells_kernel, dim3(n), dim3(256), 0, m->stream, m->view, m->shell_items_dev, m->shell_offs_dev, m->N, d_s, d_w, color ? d_c : nullptr, d_f);
    HIP_TRY(hipGetLastError());
    if (!on_device) {
        HIP_TRY(hipMemcpyAsync(sdf, st.sdf, (size_t)total * sizeof(float), hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipMemcpyAsync(weight, st.wgt, (size_t)total * sizeof(float), hipMemcpyDeviceToHost, m->stream));
        if (color) HIP_TRY(hipMemcpyAsync(rgbw, st.col, (size_t)total * sizeof(uchar4), hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipMemcpyAsync(found, st.flags, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipStreamSynchronize(m->stream));
    }
    return CHISEL_HIP_OK;  // on_device: nothing has been waited for (chisel_hip_record_event orders the consumer)
}

int chisel_hip_import_ghost_shells(chisel_hip_map *m, const int *items, int n, const float *sdf, const float *weight, const uint8_t *rgbw,
                                   const int *found, int on_device) {
    SETTLE(m);
    if (m && m->is_group) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "a call between the shards of a map");
    if (!m || n < 0 || (n > 0 && (!items || !sdf || !weight || !found))) return fail(CHISEL_HIP_ERR_INVALID, "bad argument");
    if (n == 0) return CHISEL_HIP_OK;
    HIP_TRY(hipSetDevice(m->device));
    { m->topology_epoch++; m->dirty_tail_queued = false; }
    int rc = check_mesh_totals(m);
    if (rc) return rc;
    rc = maybe_grow(m, n);  // (ghosts take slots of this shard's pool until they are dropped again)
    if (rc) return rc;
    // the distinct ghosts (several boxes may belong to one), each with the item whose `found` decides whether it is created
    std::unordered_set<uint64_t, IdHash> seen;
    std::vector<int> first;
    for (int j = 0; j < n; j++) {
        if (chunk_owner(items[4 * j], items[4 * j + 1], items[4 * j + 2], m->cfg.n_shards, m->cfg.shard_block) == m->cfg.shard_rank)
            return fail(CHISEL_HIP_ERR_INVALID, "a ghost chunk must belong to another shard");
        if (seen.insert(pack_id(items[4 * j], items[4 * j + 1], items[4 * j + 2])).second) first.push_back(j);
    }
    if (m->input_event) {  // chisel_hip_wait_event: the payload arrives on another stream (the collective's)
        HIP_TRY(hipStreamWaitEvent(m->stream, m->input_event, 0));
        m->input_event = nullptr;
    }
    long long total = 0;
    rc = stage_shell_items(m, items, n, &total);
    if (rc) return rc;
    ChunkStage st;
    const float *d_s = sdf, *d_w = weight;
    const uchar4 *d_c = reinterpret_cast<const uchar4 *>(rgbw);
    const int *d_f = found;
    if (!on_device) {
        HIP_TRY(hipMalloc(&st.sdf, (size_t)total * sizeof(float)));
        HIP_TRY(hipMalloc(&st.wgt, (size_t)total * sizeof(float)));
        HIP_TRY(hipMalloc(&st.flags, (size_t)n * sizeof(int)));
        HIP_TRY(hipMemcpyAsync(st.sdf, sdf, (size_t)total * sizeof(float), hipMemcpyHostToDevice, m->stream));
        HIP_TRY(hipMemcpyAsync(st.wgt, weight, (size_t)total * sizeof(float), hipMemcpyHostToDevice, m->stream));
        HIP_TRY(hipMemcpyAsync(st.flags, found, (size_t)n * sizeof(int), hipMemcpyHostToDevice, m->stream));
        if (rgbw) {
            HIP_TRY(hipMalloc(&st.col, (size_t)total * sizeof(uchar4)));
            HIP_TRY(hipMemcpyAsync(st.col, rgbw, (size_t)total * sizeof(uchar4), hipMemcpyHostToDevice, m->stream));
        }
        d_s = st.sdf; d_w = st.wgt; d_c = st.col; d_f = st.flags;
    }
    // phase 1: one thread block per distinct ghost creates the chunk (distinct ids never collide); phase 2: every box is written.
    // The list of first items travels through a staging buffer the map keeps (stream-ordered like the item list): nothing is
    // allocated, freed or waited for here when the payload is already on the device.
    if ((int)first.size() > m->shell_first_capacity) {
        HIP_TRY(hipStreamSynchronize(m->stream));
        if (m->shell_first_dev) HIP_TRY(hipFree(m->shell_first_dev));
        m->shell_first_dev = nullptr;
        const int cap = std::max(4096, 2 * (int)first.size());
        HIP_TRY(hipMalloc(&m->shell_first_dev, (size_t)cap * sizeof(int)));
        m->shell_first_capacity = cap;
    }
    HIP_TRY(hipMemcpyAsync(m->shell_first_dev, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(ensure_ghosts_kernel, dim3((unsigned)first.size()), dim3(64), 0, m->stream, m->view, m->shell_items_dev, m->shell_first_dev, d_f);
    hipLaunchKernelGGL(import_shells_kernel, dim3(n), dim3(256), 0, m->stream, m->view, m->shell_items_dev, m->shell_offs_dev, d_f, m->N, d_s, d_w, d_c);
    HIP_TRY(hipGetLastError());
    // every item may have become a ghost (which of them were resident at their owner is known on the device only): all are dropped
    // again by chisel_hip_drop_ghost_chunks (removing an absent id does nothing)
    for (int j = 0; j < n; j++) m->ghost_ids.insert(m->ghost_ids.end(), items + 4 * j, items + 4 * j + 3);
    if (!on_device) HIP_TRY(hipStreamSynchronize(m->stream));  // (the staging buffers above are freed on return)
    HIP_TRY(note_map_mutation(m));
    return CHISEL_HIP_OK;
}

// ---- the sharded recompute without host planning (kernels_map.h: ShellPlan) ------------------------------------------------------
namespace {
int ensure_shell_plan(chisel_hip_map *m, int jobset_capacity, int send_capacity) {
    ShellPlan &S = m->shell_plan;
    if (!S.ctl) {
        HIP_TRY(hipMalloc(&S.ctl, (16 + 4 * SHELL_MAX_SHARDS + 4) * sizeof(int)));  // ctl[16] | send_cur[64] (64-bit) | recv_cnt[64] (64-bit) | ghost chunks created so far (64-bit, never zeroed)
        HIP_TRY(hipMemsetAsync(S.ctl, 0, (16 + 4 * SHELL_MAX_SHARDS + 4) * sizeof(int), m->stream));
        S.send_cur = reinterpret_cast<unsigned long long *>(S.ctl + 16);
        S.recv_cnt = S.send_cur + SHELL_MAX_SHARDS;
        HIP_TRY(hipHostMalloc((void **)&m->shell_plan_host, (16 + 4 * SHELL_MAX_SHARDS + 4) * sizeof(int), hipHostMallocDefault));
        HIP_TRY(hipHostGetDevicePointer((void **)&m->shell_plan_host_dev, m->shell_plan_host, 0));
    }
    if (jobset_capacity > S.jobset_capacity) {
        HIP_TRY(hipStreamSynchronize(m->stream));
        if (S.jobset) HIP_TRY(hipFree(S.jobset));
        S.jobset = nullptr;
        HIP_TRY(hipMalloc(&S.jobset, 2 * (size_t)jobset_capacity * sizeof(unsigned long long)));  // (the set, then the list)
        S.jobset_capacity = jobset_capacity;
    }
    if (send_capacity > S.send_capacity) {
        HIP_TRY(hipStreamSynchronize(m->stream));
        if (S.send_items) HIP_TRY(hipFree(S.send_items));
        S.send_items = nullptr;
        HIP_TRY(hipMalloc(&S.send_items, (size_t)send_capacity * 8 * sizeof(int)));
        S.send_capacity = send_capacity;
    }
    return CHISEL_HIP_OK;
}
}  // namespace

// Step 2 of a sharded recompute, on the device: from the all-gathered list of updated chunks (`gathered_dev`: per rank 1 + 4 * cap
// ints -- count, then (x, y, z, flag) entries: chisel_hip_dirty_ids_device) this shard's jobs, the shells it sends and how much it
// receives (kernels_map.h).  out[0] = its jobs, out[1] = ghost chunks this shard's EARLIER recomputes created (for the record), out[2] = the largest per-rank count of the gathered list (> cap: entries
// were cut off -- the caller gathers again with more room; nothing else of `out` counts then), out[3] = items it sends; then per peer p
// out[4 + 2 p], out[5 + 2 p] = (items, voxels) sent to p, out[4 + 2 W + 2 p], ... = received from p.  The call waits for these figures:
// the one host wait of a sharded recompute.  A buffer that was ready behind an event: chisel_hip_wait_event first.
int chisel_hip_shell_plan_device(chisel_hip_map *m, const int *gathered_dev, int world, int cap, int64_t *out) {
    SETTLE(m);
    if (m && m->is_group) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "a call between the shards of a map");
    if (!m || !gathered_dev || !out || cap < 1 || world < 1 || world != m->cfg.n_shards || world > SHELL_MAX_SHARDS) return fail(CHISEL_HIP_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(m->device));
    int rc = ensure_mesh_jobs(m, m->view.committed);
    if (rc) return rc;
    if (m->input_event) {
        HIP_TRY(hipStreamWaitEvent(m->stream, m->input_event, 0));
        m->input_event = nullptr;
    }
    int jobset_capacity = std::max(m->shell_plan.jobset_capacity, 1 << 15), send_capacity = std::max(m->shell_plan.send_capacity, 1 << 15);
    for (int attempt = 0;; attempt++) {
        rc = ensure_shell_plan(m, jobset_capacity, send_capacity);
        if (rc) return rc;
        ShellPlan &S = m->shell_plan;
        if (!S.my_jobs || S.max_jobs < m->mesh_buf.capacity) {
            HIP_TRY(hipStreamSynchronize(m->stream));
            if (S.my_jobs) HIP_TRY(hipFree(S.my_jobs));
            S.my_jobs = nullptr;
            HIP_TRY(hipMalloc(&S.my_jobs, (size_t)m->mesh_buf.capacity * 3 * sizeof(int)));
            S.max_jobs = m->mesh_buf.capacity;
        }
        HIP_TRY(hipMemsetAsync(S.jobset, 0xff, (size_t)S.jobset_capacity * sizeof(unsigned long long), m->stream));
        HIP_TRY(hipMemsetAsync(S.ctl, 0, (16 + 4 * SHELL_MAX_SHARDS) * sizeof(int), m->stream));
        const long long threads = 27ll * cap * world;
        hipLaunchKernelGGL(shell_jobs_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, m->stream, gathered_dev, world, cap, S, m->cfg.n_shards, m->cfg.shard_rank, m->cfg.shard_block);
        hipLaunchKernelGGL(shell_items_kernel, dim3((unsigned)std::min(S.jobset_capacity / 8, 1024)), dim3(256), 0, m->stream, S, m->N, m->cfg.n_shards, m->cfg.shard_rank, m->cfg.shard_block);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(m->shell_plan_host_dev, S.ctl, (16 + 4 * SHELL_MAX_SHARDS + 4) * sizeof(int), hipMemcpyDeviceToDevice, m->stream));
        HIP_TRY(wait_stream_spinning(m->stream));
        std::atomic_thread_fence(std::memory_order_acquire);
        const int *h = m->shell_plan_host;
        if (h[2] > cap) {  // the gathered list itself was cut off: the caller's turn
            out[0] = out[1] = out[3] = 0;
            out[2] = h[2];
            return CHISEL_HIP_OK;
        }
        if (h[1] == 0 && h[3] <= S.send_capacity) break;
        if (attempt == 8) return fail(CHISEL_HIP_ERR_POOL_FULL, "sharded mesh plan: job set / job list / item list overflow after growing them (raise chisel_hip_config.max_chunks)");
        jobset_capacity = 2 * S.jobset_capacity;
        send_capacity = std::max(2 * S.send_capacity, 2 * h[3]);
    }
    const int *h = m->shell_plan_host;
    const unsigned long long *cur = reinterpret_cast<const unsigned long long *>(h + 16);
    m->shell_jobs = h[0];
    m->shell_send_items = h[3];
    out[0] = h[0]; out[2] = h[2]; out[3] = h[3];
    out[1] = (int64_t)*reinterpret_cast<const unsigned long long *>(h + 16 + 4 * SHELL_MAX_SHARDS);  // ghost chunks the earlier recomputes created (for the record)
    for (int p = 0; p < world; p++) {
        m->shell_send[p][0] = out[4 + 2 * p] = (int64_t)(cur[p] & 0xffffffffull);
        m->shell_send[p][1] = out[5 + 2 * p] = (int64_t)(cur[p] >> 32);
        m->shell_recv[p][0] = out[4 + 2 * world + 2 * p] = (int64_t)(cur[SHELL_MAX_SHARDS + p] & 0xffffffffull);
        m->shell_recv[p][1] = out[5 + 2 * world + 2 * p] = (int64_t)(cur[SHELL_MAX_SHARDS + p] >> 32);
    }
    return CHISEL_HIP_OK;
}
// bytes of the segment that carries `items` shell items with `voxels` voxels between two shards of this map (kernels_map.h)
int64_t chisel_hip_shell_segment_bytes(chisel_hip_map *m, int64_t items, int64_t voxels) {
    if (!m) return -1;
    const chisel_hip_map *s = m->is_group ? m->shards[0] : m;
    return (int64_t)shell_segment_bytes(items, voxels, s->view.rgbw != nullptr);
}
// Step 3, owner side: the segments of the latest plan -- one per peer, in rank order, back to back -- into `out_dev` (`bytes` = their
// sum: checked).  Nothing is waited for (chisel_hip_record_event orders the collective behind it).
int chisel_hip_export_shells_packed(chisel_hip_map *m, void *out_dev, int64_t bytes) {
    SETTLE(m);
    if (m && m->is_group) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "a call between the shards of a map");
    if (!m || (bytes > 0 && !out_dev)) return fail(CHISEL_HIP_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(m->device));
    int64_t want = 0;
    for (int p = 0; p < m->cfg.n_shards; p++) want += (int64_t)shell_segment_bytes(m->shell_send[p][0], m->shell_send[p][1], m->view.rgbw != nullptr);
    if (want != bytes) return fail(CHISEL_HIP_ERR_INVALID, "chisel_hip_export_shells_packed: the buffer is not the size the plan gives");
    {
        int rc_m = check_mesh_totals(m);
        if (rc_m) return rc_m;
    }
    if (m->input_event) {
        HIP_TRY(hipStreamWaitEvent(m->stream, m->input_event, 0));
        m->input_event = nullptr;
    }
    hipLaunchKernelGGL(shell_export_kernel, dim3((unsigned)std::max(1, m->shell_send_items)), dim3(256), 0, m->stream, m->view, m->shell_plan, m->N, m->cfg.n_shards,
                       static_cast<unsigned char *>(out_dev), 0ll, 0, (int *)nullptr);
    HIP_TRY(hipGetLastError());
    return CHISEL_HIP_OK;
}
// Step 4, requester side: the received segments (one per owner, in rank order, back to back: what the all-to-all of the exported
// buffers leaves) become ghost chunks.  The buffer must stay as it is until chisel_hip_drop_ghost_chunks has been queued (the ghosts are
// dropped by the ids it holds).  A buffer that is ready behind an event: chisel_hip_wait_event first.
int chisel_hip_import_shells_packed(chisel_hip_map *m, const void *in_dev, int64_t bytes) {
    SETTLE(m);
    if (m && m->is_group) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "a call between the shards of a map");
    if (!m || (bytes > 0 && !in_dev)) return fail(CHISEL_HIP_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(m->device));
    ShellSegments G;
    memset(&G, 0, sizeof(G));
    int64_t off = 0;
    int items = 0;
    for (int p = 0; p < m->cfg.n_shards; p++) {
        G.off[p] = off;
        G.first_item[p] = items;
        off += (int64_t)shell_segment_bytes(m->shell_recv[p][0], m->shell_recv[p][1], m->view.rgbw != nullptr);
        items += (int)m->shell_recv[p][0];
    }
    G.off[m->cfg.n_shards] = off;
    G.first_item[m->cfg.n_shards] = items;
    if (off != bytes) return fail(CHISEL_HIP_ERR_INVALID, "chisel_hip_import_shells_packed: the buffer is not the size the plan gives");
    { m->topology_epoch++; m->dirty_tail_queued = false; }
    int rc = check_mesh_totals(m);
    if (rc) return rc;
    if (m->input_event) {
        HIP_TRY(hipStreamWaitEvent(m->stream, m->input_event, 0));
        m->input_event = nullptr;
    }
    if (items > 0) {
        rc = maybe_grow(m, items);  // (ghosts take slots of this shard's pool until they are dropped again)
        if (rc) return rc;
        const unsigned char *in = static_cast<const unsigned char *>(in_dev);
        hipLaunchKernelGGL(shell_ensure_ghosts_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, m->stream, m->view, in, G, m->cfg.n_shards, items,
                           reinterpret_cast<unsigned long long *>(m->shell_plan.ctl + 16 + 4 * SHELL_MAX_SHARDS));
        hipLaunchKernelGGL(shell_import_kernel, dim3((unsigned)items), dim3(256), 0, m->stream, m->view, in, G, m->cfg.n_shards, m->N);
        HIP_TRY(hipGetLastError());
        HIP_TRY(note_map_mutation(m));
        m->ghost_packed = in;
        m->ghost_segments = G;
        m->ghost_packed_items = items;
        m->shell_fixed_ghosts = m->shell_redrop = false;
    }
    return CHISEL_HIP_OK;
}

// ---- the wait-free form of the same recompute (kernels_map.h: "the wait-free form"; include/chisel_hip.h) ----------------------------------
// Steps 2 and 3 without the host: the plan's kernels and, behind them, the export into `world` segments of `seg_stride` bytes each (out_dev), whose
// first workgroup also writes this rank's status vector into `status_dev` (SHELL_STATUS_INTS ints, the caller's: it all-reduces them with MAX
// in front of the exchange).  A segment that does not fit carries a head that says so; the status says it too.  The plan's buffers keep the
// sizes the last chisel_hip_shell_plan_device grew them to; a plan that outgrows them is one of the things the status reports.
// send_items_hint: what the previous recompute sent (grid size); 0 = unknown.
int chisel_hip_shell_plan_queue(chisel_hip_map *m, const int *gathered_dev, int world, int cap, int64_t seg_stride, int *status_dev, void *out_dev, int send_items_hint) {
    SETTLE(m);
    if (m && m->is_group) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "a call between the shards of a map");
    if (!m || !gathered_dev || !status_dev || !out_dev || cap < 1 || world < 1 || world != m->cfg.n_shards || world > SHELL_MAX_SHARDS || seg_stride < 16 || (seg_stride & 15))
        return fail(CHISEL_HIP_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(m->device));
    int rc = check_mesh_totals(m);  // (the recompute before this one: its second emission, if any, and the drop behind it come first)
    if (rc) return rc;
    rc = ensure_mesh_jobs(m, m->view.committed);
    if (rc) return rc;
    if (m->input_event) {
        HIP_TRY(hipStreamWaitEvent(m->stream, m->input_event, 0));
        m->input_event = nullptr;
    }
    rc = ensure_shell_plan(m, std::max(m->shell_plan.jobset_capacity, 1 << 15), std::max(m->shell_plan.send_capacity, 1 << 15));
    if (rc) return rc;
    ShellPlan &S = m->shell_plan;
    if (!S.my_jobs || S.max_jobs < m->mesh_buf.capacity) {
        HIP_TRY(hipStreamSynchronize(m->stream));
        if (S.my_jobs) HIP_TRY(hipFree(S.my_jobs));
        S.my_jobs = nullptr;
        HIP_TRY(hipMalloc(&S.my_jobs, (size_t)m->mesh_buf.capacity * 3 * sizeof(int)));
        S.max_jobs = m->mesh_buf.capacity;
    }
    HIP_TRY(hipMemsetAsync(S.jobset, 0xff, (size_t)S.jobset_capacity * sizeof(unsigned long long), m->stream));
    HIP_TRY(hipMemsetAsync(S.ctl, 0, (16 + 4 * SHELL_MAX_SHARDS) * sizeof(int), m->stream));
    const long long threads = 27ll * cap * world;
    hipLaunchKernelGGL(shell_jobs_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, m->stream, gathered_dev, world, cap, S, m->cfg.n_shards, m->cfg.shard_rank, m->cfg.shard_block);
    hipLaunchKernelGGL(shell_items_kernel, dim3((unsigned)std::min(S.jobset_capacity / 8, 1024)), dim3(256), 0, m->stream, S, m->N, m->cfg.n_shards, m->cfg.shard_rank, m->cfg.shard_block);
    // (a workgroup per item when the hint holds, several items per workgroup when there are more)
    const unsigned egrid = (unsigned)std::min<long long>(16384, std::max<long long>(256, (long long)send_items_hint + send_items_hint / 4 + 64));
    hipLaunchKernelGGL(shell_export_kernel, dim3(egrid), dim3(256), 0, m->stream, m->view, S, m->N, m->cfg.n_shards, static_cast<unsigned char *>(out_dev), (long long)seg_stride, cap, status_dev);
    HIP_TRY(hipGetLastError());
    m->shell_stride = seg_stride;
    return CHISEL_HIP_OK;
}
// Steps 4-6 behind the exchange: ghosts from the received segments, the plan's jobs meshed, the ghosts dropped -- all of it queued, none of
// it done if word 0 of the all-reduced status (`status_dev`, which stays the caller's until chisel_hip_shell_commit) is not zero.
// jobs_hint / items_hint: what the previous recompute had (grid sizes, pool growth); 0 = unknown.
int chisel_hip_import_shells_fixed(chisel_hip_map *m, const void *in_dev, int64_t seg_stride, const int *status_dev, int jobs_hint, int items_hint) {
    SETTLE(m);
    if (m && m->is_group) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "a call between the shards of a map");
    if (!m || !in_dev || !status_dev || seg_stride != m->shell_stride || !m->shell_plan.my_jobs) return fail(CHISEL_HIP_ERR_INVALID, "bad argument (chisel_hip_shell_plan_queue first)");
    HIP_TRY(hipSetDevice(m->device));
    { m->topology_epoch++; m->dirty_tail_queued = false; }
    if (m->input_event) {
        HIP_TRY(hipStreamWaitEvent(m->stream, m->input_event, 0));
        m->input_event = nullptr;
    }
    int rc = maybe_grow(m, std::max(256, 2 * items_hint));  // (ghosts take slots of this shard's pool until they are dropped again)
    if (rc) return rc;
    const unsigned char *in = static_cast<const unsigned char *>(in_dev);
    m->shell_items_grid = (unsigned)std::min<long long>(16384, std::max<long long>(256, (long long)items_hint + items_hint / 4 + 64));  // (a workgroup per item when the hint holds)
    hipLaunchKernelGGL(shell_ensure_ghosts_fixed_kernel, dim3((m->shell_items_grid + 255) / 256), dim3(256), 0, m->stream, m->view, in, (long long)seg_stride, m->cfg.n_shards, status_dev,
                       reinterpret_cast<unsigned long long *>(m->shell_plan.ctl + 16 + 4 * SHELL_MAX_SHARDS), m->shell_plan.ctl);
    hipLaunchKernelGGL(shell_import_fixed_kernel, dim3(m->shell_items_grid), dim3(256), 0, m->stream, m->view, in, (long long)seg_stride, m->cfg.n_shards, m->N, status_dev);
    HIP_TRY(hipGetLastError());
    HIP_TRY(note_map_mutation(m));
    m->ghost_packed = in;
    m->shell_fixed_ghosts = true;
    m->shell_abort_dev = status_dev;
    m->shell_jobs = jobs_hint;
    m->shell_uncommitted = true;
    return CHISEL_HIP_OK;
}
// The host's look at a wait-free recompute, once it has read the all-reduced status: the totals of its mesh step are settled (a second
// emission, and the drop behind it, if it did not fit) and, unless the recompute was called off, the bookkeeping that chisel_hip_update_meshes_planned
// left open is closed.  aborted != 0: nothing happened on the device; the caller makes the recompute again with chisel_hip_shell_plan_device.
int chisel_hip_shell_commit(chisel_hip_map *m, int aborted) {
    SETTLE(m);
    if (m && m->is_group) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "a call between the shards of a map");
    if (!m) return fail(CHISEL_HIP_ERR_INVALID, "null map");
    HIP_TRY(hipSetDevice(m->device));
    const int rc = check_mesh_totals(m);  // (the mesh step's totals: a second emission and the drop behind it, if it did not fit)
    if (m->shell_uncommitted && !aborted) m->pending_mesh_ids.clear();
    m->shell_uncommitted = false;
    m->shell_abort_dev = nullptr;
    return rc;
}

// The plan of one rank for a recompute of a sharded map (pure host arithmetic, the same on every rank for the same entries -- so
// every rank can also work out what the others will ask of it, and the request lists need no exchange of their own):
//   entries: (x, y, z, flag) x n: flag 0 = a chunk updated since the last recompute (its 27-neighbourhood is to be meshed,
//            Chisel.h:175-189), flag 1 = an id to be meshed as it is
//   jobs:    the ids of that set `rank` owns, ascending (x, then y, then z)
//   items:   (owner, x, y, z, box) x n_items, ascending by owner, then id, then box: the ghosts `rank` needs -- the 26 neighbours of
//            its jobs that other shards own -- with the boxes of each (what its jobs read of it; a box another one contains is
//            dropped, the two ends of one axis become one box)
int chisel_hip_mesh_shell_plan(const int *entries, int64_t n_entries, int n_shards, int rank, int shard_block, int *jobs, int64_t max_jobs,
                               int64_t *n_jobs, int *items, int64_t max_items, int64_t *n_items) {
    if ((n_entries > 0 && !entries) || n_shards < 1 || rank < 0 || rank >= n_shards || !n_jobs || !n_items) return fail(CHISEL_HIP_ERR_INVALID, "bad argument");
    const int sb = shard_block < 1 ? 2 : shard_block;
    std::set<std::array<int, 3>> mine;
    for (int64_t i = 0; i < n_entries; i++) {
        const int *e = entries + 4 * i;
        const int r = e[3] ? 0 : 1;
        for (int dx = -r; dx <= r; dx++)
            for (int dy = -r; dy <= r; dy++)
                for (int dz = -r; dz <= r; dz++)
                    if (chunk_owner(e[0] + dx, e[1] + dy, e[2] + dz, n_shards, sb) == rank) mine.insert({e[0] + dx, e[1] + dy, e[2] + dz});
    }
    // (owner, x, y, z) -> the boxes asked of that ghost: one per job and direction, minus those another one contains
    auto axis_within = [](int a, int b) { return b == 0 || a == b || (b == 3 && a != 0); };  // coordinates of code a within those of code b
    auto box_within = [&](int a, int b) { return axis_within(a & 3, b & 3) && axis_within((a >> 2) & 3, (b >> 2) & 3) && axis_within((a >> 4) & 3, (b >> 4) & 3); };
    std::map<std::array<int, 4>, std::vector<int>> ghosts;
    for (const auto &j : mine)
        for (int dx = -1; dx <= 1; dx++)
            for (int dy = -1; dy <= 1; dy++)
                for (int dz = -1; dz <= 1; dz++) {
                    if (!dx && !dy && !dz) continue;
                    const int g[3] = {j[0] + dx, j[1] + dy, j[2] + dz};
                    const int o = chunk_owner(g[0], g[1], g[2], n_shards, sb);
                    if (o == rank) continue;
                    const int d[3] = {dx, dy, dz};
                    int box = 0;
                    for (int a = 0; a < 3; a++) box |= (d[a] > 0 ? 1 : (d[a] < 0 ? 2 : 0)) << (2 * a);
                    std::vector<int> &v = ghosts[{o, g[0], g[1], g[2]}];
                    bool covered = false;
                    for (int b : v) covered = covered || box_within(box, b);
                    if (covered) continue;
                    v.erase(std::remove_if(v.begin(), v.end(), [&](int b) { return box_within(b, box); }), v.end());
                    v.push_back(box);
                }
    // two boxes that differ on one axis only, one end each: one box with both ends
    int64_t total_items = 0;
    for (auto &g : ghosts) {
        std::vector<int> &v = g.second;
        bool merged = true;
        while (merged) {
            merged = false;
            for (size_t i = 0; i < v.size() && !merged; i++)
                for (size_t k = i + 1; k < v.size() && !merged; k++)
                    for (int a = 0; a < 3 && !merged; a++) {
                        const int m = 3 << (2 * a), ca = (v[i] >> (2 * a)) & 3, cb = (v[k] >> (2 * a)) & 3;
                        if ((v[i] & ~m) == (v[k] & ~m) && ca != 0 && cb != 0 && ca != cb) {
                            v[i] = (v[i] & ~m) | (3 << (2 * a));
                            v.erase(v.begin() + (long)k);
                            merged = true;
                        }
                    }
        }
        std::sort(v.begin(), v.end());
        total_items += (int64_t)v.size();
    }
    *n_jobs = (int64_t)mine.size();
    *n_items = total_items;
    int64_t k = 0;
    if (jobs)
        for (const auto &j : mine) {
            if (k >= max_jobs) break;
            jobs[3 * k] = j[0]; jobs[3 * k + 1] = j[1]; jobs[3 * k + 2] = j[2];
            k++;
        }
    k = 0;
    if (items)
        for (const auto &g : ghosts)
            for (int box : g.second) {
                if (k >= max_items) break;
                items[5 * k] = g.first[0]; items[5 * k + 1] = g.first[1]; items[5 * k + 2] = g.first[2]; items[5 * k + 3] = g.first[3]; items[5 * k + 4] = box;
                k++;
            }
    return CHISEL_HIP_OK;
}
// voxels in the payload of a box of a chunk of edge n
int64_t chisel_hip_shell_volume(int box, int chunk_edge) { return (int64_t)shell_volume(box, chunk_edge); }

int chisel_hip_save_map(chisel_hip_map *m, const char *path) {
    SETTLE(m);
    if (m && m->is_group) return group::save_map(m, path);
    if (!m || !path) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(m->device));
    int rc = check_device_error(m);  // waits for the queued batches
    if (rc) return rc;
    std::vector<int> ids, slots;
    rc = fetch_listed(m, false, ids, &slots);
    if (rc) return rc;
    const size_t n = slots.size();
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) {
        for (int k = 0; k < 3; k++)
            if (ids[3 * a + k] != ids[3 * b + k]) return ids[3 * a + k] < ids[3 * b + k];
        return false;
    });
    std::ofstream out(path, std::ios::binary);
    if (!out) return fail(CHISEL_HIP_ERR_IO, std::string("cannot open ") + path);
    MapFileHeader h;
    memcpy(h.magic, "CHSLHIP1", 8);
    h.chunk_edge = m->N;
    h.resolution = m->cfg.voxel_resolution;
    h.has_color = m->view.rgbw ? 1 : 0;
    h.spare = 0;
    h.n_chunks = (int64_t)n;
    out.write(reinterpret_cast<const char *>(&h), sizeof(h));
    const size_t V = (size_t)m->V;
    std::vector<float> sdf(V), wgt(V);
    std::vector<uint8_t> col(h.has_color ? 4 * V : 0);
    for (size_t i : order) {
        const size_t off = (size_t)slots[i] * V;
        HIP_TRY(hipMemcpy(sdf.data(), m->view.sdf + off, V * sizeof(float), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(wgt.data(), m->view.wgt + off, V * sizeof(float), hipMemcpyDeviceToHost));
        if (h.has_color) HIP_TRY(hipMemcpy(col.data(), m->view.rgbw + off, 4 * V, hipMemcpyDeviceToHost));
        out.write(reinterpret_cast<const char *>(&ids[3 * i]), 3 * sizeof(int));
        out.write(reinterpret_cast<const char *>(sdf.data()), (std::streamsize)(V * sizeof(float)));
        out.write(reinterpret_cast<const char *>(wgt.data()), (std::streamsize)(V * sizeof(float)));
        if (h.has_color) out.write(reinterpret_cast<const char *>(col.data()), (std::streamsize)(4 * V));
    }
    if (!out) return fail(CHISEL_HIP_ERR_IO, std::string("write failed: ") + path);
    return CHISEL_HIP_OK;
}

int chisel_hip_load_map(chisel_hip_map *m, const char *path) {
    if (m && m->is_group) return group::for_all(m, [&](chisel_hip_map *s) { return chisel_hip_load_map(s, path); });  // every shard takes the chunks it owns
    if (!m || !path) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    std::ifstream in(path, std::ios::binary);
    if (!in) return fail(CHISEL_HIP_ERR_IO, std::string("cannot open ") + path);
    MapFileHeader h;
    in.read(reinterpret_cast<char *>(&h), sizeof(h));
    if (!in || memcmp(h.magic, "CHSLHIP1", 8) != 0) return fail(CHISEL_HIP_ERR_IO, "not a chisel-hip map file");
    if (h.chunk_edge != m->N || h.resolution != m->cfg.voxel_resolution || (h.has_color != 0) != (m->view.rgbw != nullptr))
        return fail(CHISEL_HIP_ERR_INVALID, "map file was written with another chunk size, resolution or colour setting");
    if (h.n_chunks < 0 || h.n_chunks > m->view.max_chunks) return fail(CHISEL_HIP_ERR_POOL_FULL, "map file holds more chunks than max_chunks");
    int rc = chisel_hip_reset(m);
    if (rc) return rc;
    rc = ensure_free_exact(m, h.n_chunks);  // (a growable pool: room for the file's chunks)
    if (rc) return rc;
    const size_t V = (size_t)m->V;
    std::vector<float> sdf(V), wgt(V);
    std::vector<uint8_t> col(h.has_color ? 4 * V : 0);
    for (int64_t i = 0; i < h.n_chunks; i++) {
        int id[3];
        in.read(reinterpret_cast<char *>(id), sizeof(id));
        in.read(reinterpret_cast<char *>(sdf.data()), (std::streamsize)(V * sizeof(float)));
        in.read(reinterpret_cast<char *>(wgt.data()), (std::streamsize)(V * sizeof(float)));
        if (h.has_color) in.read(reinterpret_cast<char *>(col.data()), (std::streamsize)(4 * V));
        if (!in) return fail(CHISEL_HIP_ERR_IO, std::string("truncated map file: ") + path);
        if (chunk_owner(id[0], id[1], id[2], m->cfg.n_shards, m->cfg.shard_block) != m->cfg.shard_rank) continue;  // another shard's chunk
        rc = chisel_hip_upload_chunk(m, id, sdf.data(), wgt.data(), h.has_color ? col.data() : nullptr);
        if (rc) return rc;
    }
    return CHISEL_HIP_OK;
}

int chisel_hip_upload_chunk(chisel_hip_map *m, const int id[3], const float *sdf, const float *weight, const uint8_t *rgbw) {
    SETTLE(m);
    if (m && m->is_group) return id ? chisel_hip_upload_chunk(group::owner_map(m, id), id, sdf, weight, rgbw) : fail(CHISEL_HIP_ERR_INVALID, "null argument");
    if (!m || !id || !sdf || !weight) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    if (chunk_owner(id[0], id[1], id[2], m->cfg.n_shards, m->cfg.shard_block) != m->cfg.shard_rank)
        return fail(CHISEL_HIP_ERR_INVALID, "chunk belongs to another shard");
    HIP_TRY(hipSetDevice(m->device));
    { m->topology_epoch++; m->dirty_tail_queued = false; }
    int rc = check_mesh_totals(m);  // a recompute in flight reads the voxels as they are
    if (rc) return rc;
    rc = ensure_scratch(m, 16);
    if (rc) return rc;
    rc = ensure_free_exact(m, 1);
    if (rc) return rc;
    hipLaunchKernelGGL(ensure_chunk_kernel, dim3(1), dim3(1), 0, m->stream, m->view, id[0], id[1], id[2], m->scratch_i);
    int slot = -1;
    HIP_TRY(hipMemcpyAsync(&slot, m->scratch_i, sizeof(int), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (slot < 0) return check_device_error(m) ? CHISEL_HIP_ERR_POOL_FULL : fail(CHISEL_HIP_ERR_POOL_FULL, "no slot");
    const size_t off = (size_t)slot * m->V;
    {
        const uint32_t any = SUM_ANY;  // (voxels from outside: the mesher looks at the chunk)
        HIP_TRY(hipMemcpy(m->view.slot_dirty + 2 * (size_t)m->view.max_chunks + SLOT_SUMMARY_PAD + slot, &any, sizeof(any), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(m->view.sdf + off, sdf, (size_t)m->V * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m->view.wgt + off, weight, (size_t)m->V * sizeof(float), hipMemcpyHostToDevice));
    if (m->view.rgbw) {
        if (rgbw) HIP_TRY(hipMemcpy(m->view.rgbw + off, rgbw, (size_t)m->V * 4, hipMemcpyHostToDevice));
        else HIP_TRY(hipMemset(m->view.rgbw + off, 0, (size_t)m->V * 4));
    }
    return CHISEL_HIP_OK;
}

int chisel_hip_meshes_to_update(chisel_hip_map *m, int *ids, int64_t max_ids, int64_t *count) {
    SETTLE(m);
    if (m && m->is_group) {
        std::vector<int> all;
        const int rc = group::gather_ids(m, chisel_hip_meshes_to_update, true, all);
        return rc ? rc : group::emit_ids(all, ids, max_ids, count);
    }
    if (!m || !count) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(m->device));
    std::vector<int> dirty;
    int rc = fetch_listed(m, true, dirty, nullptr);
    if (rc) return rc;
    std::unordered_set<uint64_t, IdHash> all(m->pending_mesh_ids);
    expand27(dirty, all);
    *count = (int64_t)all.size();
    if (ids) {
        int64_t k = 0;
        for (uint64_t key : all) {
            if (k >= max_ids) break;
            unpack_id(key, ids[3 * k], ids[3 * k + 1], ids[3 * k + 2]);
            k++;
        }
    }
    return CHISEL_HIP_OK;
}

// Chisel::GetMeshesToUpdate for a caller that keeps its copy of the set (see chisel_hip.h).  The device lists the slots whose dirty flag
// went 0 -> 1 in order of appearance (mark_slot_dirty): what joined since the caller's cursor is the tail of that list, expanded to the
// 27-neighbourhoods (Chisel.h:175-189) here, plus the entries kept on the host for chunks that were removed while dirty.
constexpr int DIRTY_TAIL_CAP = 16384;
int chisel_hip_meshes_to_update_since(chisel_hip_map *m, uint64_t cursor[2], int *ids, int64_t max_ids, int64_t *count, int *cleared) {
    SETTLE(m);
    if (!m || !cursor || !count || !cleared || max_ids < 0 || (max_ids > 0 && !ids)) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    if (m->is_group) {  // (a group's shards keep a list each: the whole set every time)
        *cleared = 1;
        return chisel_hip_meshes_to_update(m, ids, max_ids, count);
    }
    HIP_TRY(hipSetDevice(m->device));
    const uint64_t epoch_tag = (uint64_t)m->dirty_epoch + 1u;  // (a zero cursor matches no epoch)
    const bool restart = (cursor[0] >> 32) != epoch_tag;
    unsigned from = restart ? 0u : (unsigned)(cursor[0] & 0xffffffffull);
    if (!m->dirty_tail_host) {
        HIP_TRY(hipHostMalloc((void **)&m->dirty_tail_host, (2 + 3 * (size_t)DIRTY_TAIL_CAP) * sizeof(int), hipHostMallocDefault));
        HIP_TRY(hipHostGetDevicePointer((void **)&m->dirty_tail_dev, m->dirty_tail_host, 0));
    }
    // (the kernel may have been queued behind the integration already -- chisel_hip_meshes_to_update_prefetch -- and the caller's wait for the
    // integration then covered it: no launch and no second wait here)
    const bool prefetched = m->dirty_tail_queued && !restart && m->dirty_tail_cursor == cursor[0] && m->dirty_tail_batch == m->batch_seq;
    m->dirty_tail_queued = false;
    if (!prefetched) {
        hipLaunchKernelGGL(list_dirty_tail_kernel, dim3(1), dim3(256), 0, m->stream, m->view, from, m->dirty_tail_dev, DIRTY_TAIL_CAP);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(wait_stream_spinning(m->stream));
    std::atomic_thread_fence(std::memory_order_acquire);
    const unsigned listed = (unsigned)m->dirty_tail_host[0];
    const int n_new = m->dirty_tail_host[1];
    std::unordered_set<uint64_t, IdHash> joined;
    bool whole = restart;
    if (listed > (unsigned)m->view.max_chunks || n_new > DIRTY_TAIL_CAP) {
        // the list overflowed (or the tail does not fit the staging buffer): the whole set from the flags
        std::vector<int> dirty;
        int rc = fetch_listed(m, true, dirty, nullptr);
        if (rc) return rc;
        expand27(dirty, joined);
        whole = true;
    } else {
        std::vector<int> fresh(m->dirty_tail_host + 2, m->dirty_tail_host + 2 + 3 * (size_t)n_new);
        expand27(fresh, joined);
    }
    if (whole || cursor[1] != m->pending_version) joined.insert(m->pending_mesh_ids.begin(), m->pending_mesh_ids.end());
    *count = (int64_t)joined.size();
    *cleared = whole ? 1 : 0;
    if (*count > max_ids) return CHISEL_HIP_OK;  // nothing consumed: the caller comes back with room for *count ids
    int64_t k = 0;
    for (uint64_t key : joined) {
        unpack_id(key, ids[3 * k], ids[3 * k + 1], ids[3 * k + 2]);
        k++;
    }
    cursor[0] = (epoch_tag << 32) | (uint64_t)std::min(listed, (unsigned)m->view.max_chunks);
    cursor[1] = m->pending_version;
    return CHISEL_HIP_OK;
}

// Queues the listing of chisel_hip_meshes_to_update_since behind what the map has queued so far (the integration of the frame just
// handed over) without waiting: the caller's own wait for that integration (chisel_hip_synchronize: the reference's calls are synchronous)
// then covers the listing too, and the _since call that follows finds its result.  Nothing happens for a cursor of another epoch.
int chisel_hip_meshes_to_update_prefetch(chisel_hip_map *m, const uint64_t cursor[2]) {
    SETTLE(m);
    if (!m || !cursor) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    if (m->is_group) return CHISEL_HIP_OK;
    if ((cursor[0] >> 32) != (uint64_t)m->dirty_epoch + 1u) return CHISEL_HIP_OK;  // (the set was emptied since: the _since call starts over)
    HIP_TRY(hipSetDevice(m->device));
    if (!m->dirty_tail_host) {
        HIP_TRY(hipHostMalloc((void **)&m->dirty_tail_host, (2 + 3 * (size_t)DIRTY_TAIL_CAP) * sizeof(int), hipHostMallocDefault));
        HIP_TRY(hipHostGetDevicePointer((void **)&m->dirty_tail_dev, m->dirty_tail_host, 0));
    }
    hipLaunchKernelGGL(list_dirty_tail_kernel, dim3(1), dim3(256), 0, m->stream, m->view, (unsigned)(cursor[0] & 0xffffffffull), m->dirty_tail_dev, DIRTY_TAIL_CAP);
    HIP_TRY(hipGetLastError());
    m->dirty_tail_queued = true;
    m->dirty_tail_cursor = cursor[0];
    m->dirty_tail_batch = m->batch_seq;
    return CHISEL_HIP_OK;
}

int chisel_hip_get_counters(chisel_hip_map *m, uint64_t *out, int reset_counters) {
    SETTLE(m);
    if (m && m->is_group) return out ? group::get_counters(m, out, reset_counters) : fail(CHISEL_HIP_ERR_INVALID, "null argument");
    if (!m || !out) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(m->device));
    hipLaunchKernelGGL(reduce_counters_kernel, dim3(1), dim3(256), 0, m->stream, m->view, INTEGRATE_MAX_GRID);
    HIP_TRY(hipMemcpyAsync(out, m->view.counters, CHISEL_HIP_NUM_COUNTERS * sizeof(uint64_t), hipMemcpyDeviceToHost, m->stream));
#ifdef CHISEL_PHASES
    uint64_t ph[25];
    HIP_TRY(hipMemcpyAsync(ph, m->view.counters, sizeof(ph), hipMemcpyDeviceToHost, m->stream));
    std::vector<uint64_t> rows((size_t)INTEGRATE_MAX_GRID * 16);
    HIP_TRY(hipMemcpyAsync(rows.data(), m->view.block_counters + (size_t)INTEGRATE_MAX_GRID * 16, rows.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, m->stream));
#endif
    if (reset_counters)
        HIP_TRY(hipMemsetAsync(m->view.block_counters, 0, (size_t)INTEGRATE_MAX_GRID * 32 * sizeof(unsigned long long), m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
#ifdef CHISEL_PHASES
    if (ph[15])  // s_memrealtime ticks of 10 ns, summed over the waves of all launches since the last reset
        fprintf(stderr, "integrate_kernel phases, us per wave: item %.2f frames %.2f deposit %.2f store+or %.2f dequeue %.2f | alive %.2f | waves %llu units/wave %.2f\n",
                ph[9] * 0.01 / ph[15], ph[10] * 0.01 / ph[15], ph[11] * 0.01 / ph[15], ph[12] * 0.01 / ph[15], ph[13] * 0.01 / ph[15],
                ph[14] * 0.01 / ph[15], (unsigned long long)ph[15], (double)out[5] * 16.0 / ph[15]);
    if (ph[15])
        fprintf(stderr, "   units %llu (%.2f per wave)  frames visited %.2f per unit, executed %.2f per unit (%.2f us each), with a band update %.2f per unit\n",
                (unsigned long long)ph[19], (double)ph[19] / ph[15], (double)ph[16] / ph[19], (double)ph[17] / ph[19], ph[17] ? ph[18] * 0.01 / ph[17] : 0.0, (double)ph[20] / ph[19]);
    if (ph[15]) {  // lane-level utilisation of the executed wave-frames
        uint64_t u[6] = {0, 0, 0, 0, 0, 0};
        for (int b = 0; b < 64; b++)
            for (int c = 0; c < 3; c++) { u[c] += rows[(size_t)b * 16 + 13 + c]; u[3 + c] += rows[(size_t)(64 + b) * 16 + 13 + c]; }
        const double ex = (double)ph[17];
        fprintf(stderr, "   per executed wave-frame: lanes needed %.1f / 64, voxels with a record %.1f / 256, voxels in band or carve test %.1f / 256; band code run by %.2f of them with %.1f lanes, carve code by %.2f with %.1f lanes\n",
                u[0] / ex, u[1] / ex, u[2] / ex, ph[20] / ex, ph[20] ? (double)u[3] / ph[20] : 0.0, u[5] / ex, u[5] ? (double)u[4] / u[5] : 0.0);
    }
    {
        unsigned long long mp[8];
        if (hipMemcpyFromSymbol(mp, HIP_SYMBOL(g_mesh_phase), sizeof(mp)) == hipSuccess && mp[7]) {
            fprintf(stderr, "mesh_count_kernel, us per job (thread 0): lookups %.2f | corners staged %.2f | cubes classified %.2f | scan %.2f | reserve %.2f | records %.2f | jobs %llu\n",
                    mp[0] * 0.01 / mp[7], mp[1] * 0.01 / mp[7], mp[2] * 0.01 / mp[7], mp[3] * 0.01 / mp[7], mp[4] * 0.01 / mp[7], mp[5] * 0.01 / mp[7], mp[7]);
            unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mesh_phase), z, sizeof(z));
        }
        if (hipMemcpyFromSymbol(mp, HIP_SYMBOL(g_tri_phase), sizeof(mp)) == hipSuccess && mp[7]) {
            fprintf(stderr, "mesh_triangle_kernel, us per wave (lane 0): prefix + search %.2f | record, job row, position %.2f | corners, three vertices %.2f | gradient %.2f | colour %.2f | waves sampled %llu\n",
                    mp[0] * 0.01 / mp[7], mp[1] * 0.01 / mp[7], mp[2] * 0.01 / mp[7], mp[3] * 0.01 / mp[7], mp[4] * 0.01 / mp[7], mp[7]);
            unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tri_phase), z, sizeof(z));
        }
    }
    if (ph[15] && ph[17]) {  // where an executed frame's time goes (10 ns ticks summed over waves)
        uint64_t f[6] = {0, 0, 0, 0, 0, 0};
        for (int b = 0; b < 64; b++)
            for (int c = 0; c < 3; c++) { f[c] += rows[(size_t)(128 + b) * 16 + 13 + c]; f[3 + c] += rows[(size_t)(192 + b) * 16 + 13 + c]; }
        const double ex = (double)ph[17] * 100.0;
        fprintf(stderr, "   an executed frame, us: constants + z test %.2f | state loads + projection, gathers issued %.2f | records arrived, band / carve verdicts %.2f | band update (colour) %.2f | carve %.2f\n",
                f[0] / ex, f[1] / ex, f[2] / ex, f[3] / ex, f[4] / ex);
    }
    ph[21] = ~ph[21];
    if (ph[15] && getenv("CHISEL_HIP_PHASE_TABLE")) {  // wave 0 of every block, grouped by dispatch generation (256 blocks each)
        uint64_t t0 = ~0ull;
        for (int b = 0; b < INTEGRATE_MAX_GRID; b++)
            if (rows[(size_t)b * 16 + 9] && rows[(size_t)b * 16 + 9] < t0) t0 = rows[(size_t)b * 16 + 9];
        for (int g = 0; g < INTEGRATE_MAX_GRID / 256; g++) {
            double s_start = 0, s_end = 0, s_units = 0, s_frames = 0, mx_end = 0, mn_end = 1e30;
            int n = 0;
            for (int b = g * 256; b < (g + 1) * 256; b++) {
                const uint64_t *r = &rows[(size_t)b * 16];
                if (!r[9]) continue;
                n++;
                s_start += (r[9] - t0) * 0.01; s_end += (r[10] - t0) * 0.01; s_units += (double)r[11]; s_frames += (double)r[12];
                mx_end = std::max(mx_end, (r[10] - t0) * 0.01); mn_end = std::min(mn_end, (r[10] - t0) * 0.01);
            }
            if (n) fprintf(stderr, "   blocks %4d-%4d: start %.1f us, end %.1f us (min %.1f max %.1f), units %.2f, frames executed %.1f\n", g * 256, g * 256 + 255,
                           s_start / n, s_end / n, mn_end, mx_end, s_units / n, s_frames / n);
        }
    }
    if (ph[15])
        fprintf(stderr, "   (meaningful for ONE launch) first wave start -> last wave end %.1f us; last unit started at %.1f us; longest unit %.1f us (%d frames executed, wid %d = item %d unit %d)\n",
                (ph[22] - ph[21]) * 0.01, (ph[23] - ph[21]) * 0.01, (ph[24] >> 32) * 0.01, (int)(ph[24] & 255), (int)((ph[24] >> 8) & 0xffffff), (int)((ph[24] >> 8) & 0xffffff) / 16, (int)((ph[24] >> 8) & 0xffffff) % 16);
#endif
    return CHISEL_HIP_OK;
}

int chisel_hip_mc_tables(int *triangle_table, int *edge_index_pairs) {
    static const unsigned long long cases[256] = CHISEL_MC_PACKED_CASES;
    static const unsigned char edges[12] = CHISEL_MC_EDGE_CORNERS;
    if (triangle_table)
        for (int i = 0; i < 256; i++)
            for (int k = 0; k < 16; k++) {
                const int e = (int)((cases[i] >> (4 * k)) & 0xF);
                triangle_table[16 * i + k] = e == 0xF ? -1 : e;
            }
    if (edge_index_pairs)
        for (int e = 0; e < 12; e++) {
            edge_index_pairs[2 * e] = edges[e] & 0xF;
            edge_index_pairs[2 * e + 1] = edges[e] >> 4;
        }
    return CHISEL_HIP_OK;
}

int chisel_hip_mesh_cube_values(const float *vertex_coords, const float *vertex_sdf, float *edge_coords, int *configuration, float *vertices,
                                float *normals, int *n_vertices) {
    if (!vertex_coords || !vertex_sdf) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    float *d = nullptr;
    HIP_TRY(hipMalloc(&d, (32 + 2 + 36 + 90) * sizeof(float)));
    float in[32];
    memcpy(in, vertex_coords, 24 * sizeof(float));
    memcpy(in + 24, vertex_sdf, 8 * sizeof(float));
    hipError_t e = hipMemcpy(d, in, sizeof(in), hipMemcpyHostToDevice);
    float out[2 + 36 + 90];
    if (e == hipSuccess) {
        hipLaunchKernelGGL(mesh_cube_values_kernel, dim3(1), dim3(1), 0, 0, (const float *)d, (const float *)(d + 24), d + 32);
        e = hipMemcpy(out, d + 32, sizeof(out), hipMemcpyDeviceToHost);
    }
    (void)hipFree(d);
    if (e != hipSuccess) return fail(CHISEL_HIP_ERR_HIP, std::string("chisel_hip_mesh_cube_values: ") + hipGetErrorString(e));
    const int nv = (int)out[1];
    if (configuration) *configuration = (int)out[0];
    if (n_vertices) *n_vertices = nv;
    if (edge_coords) memcpy(edge_coords, out + 2, 36 * sizeof(float));
    if (vertices) memcpy(vertices, out + 2 + 36, (size_t)nv * 3 * sizeof(float));
    if (normals) memcpy(normals, out + 2 + 36 + 45, (size_t)nv * 3 * sizeof(float));
    return CHISEL_HIP_OK;
}

int chisel_hip_interpolate_vertex(const float v1[3], const float v2[3], float sdf1, float sdf2, float out[3]) {
    if (!v1 || !v2 || !out) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    float *d = nullptr;
    HIP_TRY(hipMalloc(&d, 12 * sizeof(float)));
    const float in[8] = {v1[0], v1[1], v1[2], v2[0], v2[1], v2[2], sdf1, sdf2};
    hipError_t e = hipMemcpy(d, in, sizeof(in), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(interpolate_vertex_kernel, dim3(1), dim3(1), 0, 0, (const float *)d, d + 8);
        e = hipMemcpy(out, d + 8, 3 * sizeof(float), hipMemcpyDeviceToHost);
    }
    (void)hipFree(d);
    if (e != hipSuccess) return fail(CHISEL_HIP_ERR_HIP, std::string("chisel_hip_interpolate_vertex: ") + hipGetErrorString(e));
    return CHISEL_HIP_OK;
}

int chisel_hip_raycast(const float start[3], const float end[3], const int min_xyz[3], const int max_xyz[3], int *cells, int64_t capacity,
                       int64_t *count) {
    if (!start || !end || !min_xyz || !max_xyz || !count || capacity < 0 || (capacity > 0 && !cells)) return fail(CHISEL_HIP_ERR_INVALID, "bad argument");
    const int cap = (int)std::min<int64_t>(capacity, 1 << 24);
    float *d_in = nullptr;
    int *d_cells = nullptr, *d_count = nullptr;
    HIP_TRY(hipMalloc(&d_in, 6 * sizeof(float)));
    HIP_TRY(hipMalloc(&d_cells, ((size_t)cap * 3 + 4) * sizeof(int)));
    HIP_TRY(hipMalloc(&d_count, sizeof(int)));
    const float in[6] = {start[0], start[1], start[2], end[0], end[1], end[2]};
    hipError_t e = hipMemcpy(d_in, in, sizeof(in), hipMemcpyHostToDevice);
    int c = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(kat_raycast_kernel, dim3(1), dim3(64), 0, 0, (const float *)d_in, 1, make_int3(min_xyz[0], min_xyz[1], min_xyz[2]),
                           make_int3(max_xyz[0], max_xyz[1], max_xyz[2]), d_cells, cap, d_count);
        e = hipMemcpy(&c, d_count, sizeof(int), hipMemcpyDeviceToHost);
        if (e == hipSuccess && cap > 0 && c > 0) e = hipMemcpy(cells, d_cells, (size_t)std::min(c, cap) * 3 * sizeof(int), hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_in); (void)hipFree(d_cells); (void)hipFree(d_count);
    if (e != hipSuccess) return fail(CHISEL_HIP_ERR_HIP, std::string("chisel_hip_raycast: ") + hipGetErrorString(e));
    *count = c;
    return CHISEL_HIP_OK;
}

int chisel_hip_pool_info(chisel_hip_map *m, int64_t out[4]) {
    if (!m || !out) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    out[0] = out[1] = out[2] = out[3] = 0;
    if (m->is_group)
        return group::for_all(m, [&](chisel_hip_map *s) {
            int64_t v[4];
            const int rc = chisel_hip_pool_info(s, v);
            for (int k = 0; k < 3; k++) out[k] += v[k];
            out[3] = std::max(out[3], v[3]);
            return rc;
        });
    out[0] = m->view.committed;
    out[1] = m->growable ? m->view.max_chunks : m->view.committed;
    out[2] = m->grow_events;
    out[3] = m->growable ? 1 : 0;
    return CHISEL_HIP_OK;
}

int chisel_hip_get_launch_stats(chisel_hip_map *m, int64_t *out, int reset_stats) {
    if (!m || !out) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    for (int k = 0; k < CHISEL_HIP_NUM_LAUNCH_STATS; k++) out[k] = 0;
    auto take = [&](chisel_hip_map *s) {
        for (int k = 0; k < CHISEL_HIP_NUM_LAUNCH_STATS; k++) {
            out[k] += s->launch_stats[k];
            if (reset_stats) s->launch_stats[k] = 0;
        }
    };
    if (m->is_group)
        for (chisel_hip_map *s : m->shards) take(s);
    else
        take(m);
    return CHISEL_HIP_OK;
}

int chisel_hip_memory_statistics(chisel_hip_map *m, chisel_hip_statistics *out) {
    SETTLE(m);
    if (!m || !out) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    memset(out, 0, sizeof(*out));
    for (int a = 0; a < 3; a++) {
        out->id_min[a] = INT32_MAX;
        out->id_max[a] = INT32_MIN;
    }
    if (m->is_group) {  // the shards' chunks are disjoint: sums, and extrema of extrema
        return group::for_all(m, [&](chisel_hip_map *s) -> int {
            chisel_hip_statistics t;
            const int rc = chisel_hip_memory_statistics(s, &t);
            if (rc) return rc;
            out->n_unknown += t.n_unknown; out->n_known_inside += t.n_known_inside; out->n_known_outside += t.n_known_outside;
            out->total_weight += t.total_weight; out->n_chunks += t.n_chunks;
            for (int a = 0; a < 3 && t.n_chunks; a++) {
                out->id_min[a] = std::min(out->id_min[a], t.id_min[a]);
                out->id_max[a] = std::max(out->id_max[a], t.id_max[a]);
            }
            return CHISEL_HIP_OK;
        });
    }
    HIP_TRY(hipSetDevice(m->device));
    int rc = ensure_scratch(m, sizeof(CensusOut) / sizeof(int) + 4);
    if (rc) return rc;
    CensusOut init;
    memset(&init, 0, sizeof(init));
    for (int a = 0; a < 3; a++) {
        init.id_min[a] = INT32_MAX;
        init.id_max[a] = INT32_MIN;
    }
    CensusOut *d = reinterpret_cast<CensusOut *>(m->scratch_i);
    HIP_TRY(hipMemcpyAsync(d, &init, sizeof(init), hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(census_kernel, dim3(std::min(m->view.max_chunks, 8192)), dim3(256), 0, m->stream, m->view, m->V, d);
    HIP_TRY(hipGetLastError());
    CensusOut h;
    HIP_TRY(hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    out->n_unknown = (int64_t)h.unknown; out->n_known_inside = (int64_t)h.inside; out->n_known_outside = (int64_t)h.outside;
    out->total_weight = h.weight; out->n_chunks = (int64_t)h.chunks;
    for (int a = 0; a < 3; a++) {
        out->id_min[a] = h.id_min[a];
        out->id_max[a] = h.id_max[a];
    }
    return CHISEL_HIP_OK;
}

int chisel_hip_topology_epoch(chisel_hip_map *m, uint64_t *out) {
    SETTLE(m);
    if (!m || !out) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    *out = m->topology_epoch;
    if (m->is_group)
        for (chisel_hip_map *s : m->shards) *out += s->topology_epoch;
    return CHISEL_HIP_OK;
}

// ChunkManager::GetChunkIDsIntersecting(const Frustum &, ChunkIDList *) (ChunkManager.cpp:182-212) for a frustum given by its corners
// and planes: the AABB of the corners (Frustum::ComputeBoundingBox Frustum.cpp:101-122), the id range [GetIDAt(min) - 1,
// GetIDAt(max) + 2] per axis walked x outer, z inner, every box kept for which Frustum::Intersects holds (Frustum.cpp:41-79).
// Host arithmetic, as in the reference (the integration path does not call it: its candidates come from cull_kernel).
int chisel_hip_candidates(const float corners[24], const float planes[24], const int chunk_size[3], float voxel_resolution, int *ids, int64_t max_ids,
                          int64_t *count) {
    if (!corners || !planes || !chunk_size || !count || !(voxel_resolution > 0.0f)) return fail(CHISEL_HIP_ERR_INVALID, "bad argument");
    float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    for (int i = 0; i < 8; i++)
        for (int k = 0; k < 3; k++) {
            mn[k] = std::min(mn[k], corners[3 * i + k]);
            mx[k] = std::max(mx[k], corners[3 * i + k]);
        }
    int lo[3], hi[3];
    for (int k = 0; k < 3; k++) {
        const float rf = 1.0f / (chunk_size[k] * voxel_resolution);  // ChunkManager::GetIDAt ChunkManager.h:136-145
        if (!(std::fabs(mn[k] * rf) < 1e6f) || !(std::fabs(mx[k] * rf) < 1e6f)) return fail(CHISEL_HIP_ERR_INVALID, "frustum outside the addressable chunk-id range");
        lo[k] = (int)std::floor(mn[k] * rf) - 1;
        hi[k] = (int)std::floor(mx[k] * rf) + 1 + 1;
    }
    int64_t n = 0;
    for (int x = lo[0]; x <= hi[0]; x++)
        for (int y = lo[1]; y <= hi[1]; y++)
            for (int z = lo[2]; z <= hi[2]; z++) {
                const float bmin[3] = {(float)(x * chunk_size[0]) * voxel_resolution, (float)(y * chunk_size[1]) * voxel_resolution,
                                       (float)(z * chunk_size[2]) * voxel_resolution};
                const float bmax[3] = {bmin[0] + (float)chunk_size[0] * voxel_resolution, bmin[1] + (float)chunk_size[1] * voxel_resolution,
                                       bmin[2] + (float)chunk_size[2] * voxel_resolution};
                bool hit = false;
                for (int p = 0; p < 6 && !hit; p++) {
                    const float *pl = planes + 4 * p;
                    const float vx = pl[0] < 0.0f ? bmin[0] : bmax[0], vy = pl[1] < 0.0f ? bmin[1] : bmax[1], vz = pl[2] < 0.0f ? bmin[2] : bmax[2];
                    hit = (vx * pl[0] + (vy * pl[1] + vz * pl[2])) + pl[3] > 0.0f;  // axisVert.dot(normal) + distance, a0 + (a1 + a2)
                }
                if (!hit) continue;
                if (ids && n < max_ids) {
                    ids[3 * n] = x; ids[3 * n + 1] = y; ids[3 * n + 2] = z;
                }
                n++;
            }
    *count = n;
    return CHISEL_HIP_OK;
}

int chisel_hip_shade_vertices(chisel_hip_map *m, const float *vertices, int64_t n, float *normals, float *colors, int stages) {
    SETTLE(m);
    if (m && m->is_group) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "chisel_hip_shade_vertices reads the voxels around every vertex: ask the shard that owns them (a group's meshes are shaded by chisel_hip_update_meshes)");
    if (!m || n < 0 || (n > 0 && !vertices)) return fail(CHISEL_HIP_ERR_INVALID, "bad argument");
    if (n == 0) return CHISEL_HIP_OK;
    HIP_TRY(hipSetDevice(m->device));
    {
        int rc_m = check_mesh_totals(m);
        if (rc_m) return rc_m;
    }
    float *d = nullptr;
    const size_t f = (size_t)n * 3;
    HIP_TRY(hipMalloc(&d, 3 * f * sizeof(float)));
    float *dv = d, *dn = d + f, *dc = d + 2 * f;
    HIP_TRY(hipMemcpyAsync(dv, vertices, f * sizeof(float), hipMemcpyHostToDevice, m->stream));
    if (normals && (stages & 1)) HIP_TRY(hipMemcpyAsync(dn, normals, f * sizeof(float), hipMemcpyHostToDevice, m->stream));
    const MeshParams P = mesh_params(m);
    const dim3 grid((unsigned)((n + 255) / 256));
    switch (m->N) {
        case 8: hipLaunchKernelGGL(shade_vertices_kernel<8>, grid, dim3(256), 0, m->stream, m->view, P, dv, (long long)n, dn, dc, stages); break;
        case 16: hipLaunchKernelGGL(shade_vertices_kernel<16>, grid, dim3(256), 0, m->stream, m->view, P, dv, (long long)n, dn, dc, stages); break;
        case 32: hipLaunchKernelGGL(shade_vertices_kernel<32>, grid, dim3(256), 0, m->stream, m->view, P, dv, (long long)n, dn, dc, stages); break;
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && normals && (stages & 1)) e = hipMemcpyAsync(normals, dn, f * sizeof(float), hipMemcpyDeviceToHost, m->stream);
    if (e == hipSuccess && colors && (stages & 2) && m->view.rgbw) e = hipMemcpyAsync(colors, dc, f * sizeof(float), hipMemcpyDeviceToHost, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(CHISEL_HIP_ERR_HIP, std::string("chisel_hip_shade_vertices: ") + hipGetErrorString(e));
    return CHISEL_HIP_OK;
}

int chisel_hip_set_profiling(chisel_hip_map *m, int enable) {
    if (m && m->is_group) return group::for_all(m, [&](chisel_hip_map *s) { return chisel_hip_set_profiling(s, enable); });
    if (!m) return fail(CHISEL_HIP_ERR_INVALID, "null map");
    int rc = drain_profile(m);
    m->profiling = enable != 0;
    return rc;
}

int chisel_hip_get_profile(chisel_hip_map *m, double *ms_total, int64_t *launches, int reset_profile) {
    if (m && m->is_group) return (ms_total && launches) ? group::get_profile(m, ms_total, launches, reset_profile) : fail(CHISEL_HIP_ERR_INVALID, "null argument");
    if (!m) return fail(CHISEL_HIP_ERR_INVALID, "null map");
    int rc = drain_profile(m);
    if (rc) return rc;
    for (int k = 0; k < CHISEL_HIP_NUM_KERNELS; k++) {
        if (ms_total) ms_total[k] = m->prof_ms[k];
        if (launches) launches[k] = m->prof_launches[k];
        if (reset_profile) {
            m->prof_ms[k] = 0;
            m->prof_launches[k] = 0;
        }
    }
    return CHISEL_HIP_OK;
}

// ---- known-answer entry points (tests only; not part of the reference surface) -------------------------
int chisel_hip_kat_truncation(int kind, float param, const float *depths, int n, float *trunc, float *weight1) {
    float *d_in = nullptr, *d_t = nullptr, *d_w = nullptr;
    HIP_TRY(hipMalloc(&d_in, n * sizeof(float)));
    HIP_TRY(hipMalloc(&d_t, n * sizeof(float)));
    HIP_TRY(hipMalloc(&d_w, n * sizeof(float)));
    HIP_TRY(hipMemcpy(d_in, depths, n * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kat_truncation_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, kind, param, d_in, n, d_t, d_w);
    HIP_TRY(hipMemcpy(trunc, d_t, n * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(weight1, d_w, n * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(d_in); (void)hipFree(d_t); (void)hipFree(d_w);
    return CHISEL_HIP_OK;
}
int chisel_hip_kat_dist(const float *ops, int n, float *out) {
    float *d_in = nullptr, *d_out = nullptr;
    HIP_TRY(hipMalloc(&d_in, n * 3 * sizeof(float)));
    HIP_TRY(hipMalloc(&d_out, n * 2 * sizeof(float)));
    HIP_TRY(hipMemcpy(d_in, ops, n * 3 * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kat_dist_kernel, dim3(1), dim3(64), 0, 0, d_in, n, d_out);
    HIP_TRY(hipMemcpy(out, d_out, n * 2 * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(d_in); (void)hipFree(d_out);
    return CHISEL_HIP_OK;
}
int chisel_hip_kat_raycast(const float *rays, int n, const int lo[3], const int hi[3], int *cells, int cap, int *count) {
    float *d_in = nullptr;
    int *d_cells = nullptr, *d_count = nullptr;
    HIP_TRY(hipMalloc(&d_in, (size_t)n * 6 * sizeof(float)));
    HIP_TRY(hipMalloc(&d_cells, (size_t)n * cap * 3 * sizeof(int)));
    HIP_TRY(hipMalloc(&d_count, (size_t)n * sizeof(int)));
    HIP_TRY(hipMemcpy(d_in, rays, (size_t)n * 6 * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kat_raycast_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, d_in, n, make_int3(lo[0], lo[1], lo[2]),
                       make_int3(hi[0], hi[1], hi[2]), d_cells, cap, d_count);
    HIP_TRY(hipMemcpy(cells, d_cells, (size_t)n * cap * 3 * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(count, d_count, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(d_in); (void)hipFree(d_cells); (void)hipFree(d_count);
    return CHISEL_HIP_OK;
}
int chisel_hip_kat_color(const uint8_t *ops, int n, uint8_t *out) {
    uint8_t *d_in = nullptr, *d_out = nullptr;
    HIP_TRY(hipMalloc(&d_in, n * 4));
    HIP_TRY(hipMalloc(&d_out, n * 4));
    HIP_TRY(hipMemcpy(d_in, ops, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kat_color_kernel, dim3(1), dim3(64), 0, 0, d_in, n, d_out);
    HIP_TRY(hipMemcpy(out, d_out, n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d_in); (void)hipFree(d_out);
    return CHISEL_HIP_OK;
}
int chisel_hip_kat_color_fresh(unsigned *mismatches) {
    unsigned *d = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof(unsigned)));
    HIP_TRY(hipMemset(d, 0, sizeof(unsigned)));
    hipLaunchKernelGGL(kat_color_fresh_kernel, dim3(8 * 256 * 256 / 256), dim3(256), 0, 0, d);
    HIP_TRY(hipMemcpy(mismatches, d, sizeof(unsigned), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return CHISEL_HIP_OK;
}
// diagnostics of the last cloud: listed chunks, (unit, point) pairs, rays of the largest unit, units with rays
int chisel_hip_debug_cloud_stats(chisel_hip_map *m, int64_t out[4]) {
    SETTLE(m);
    if (m && m->is_group) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "per-shard read-out");
    if (!m || !m->cloud.view.ctl) return fail(CHISEL_HIP_ERR_INVALID, "no cloud yet");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipStreamSynchronize(m->stream));
    int ctl[2] = {0, 0};
    HIP_TRY(hipMemcpy(ctl, m->cloud.view.ctl, sizeof(ctl), hipMemcpyDeviceToHost));
    const int units = std::min(ctl[0], CLOUD_MAX_LISTED) * CloudUnits(m->N, 0, cloud_unit_depth(m->N)).count;
    std::vector<int> off((size_t)units + 1);
    HIP_TRY(hipMemcpy(off.data(), m->cloud.view.offsets, off.size() * sizeof(int), hipMemcpyDeviceToHost));
    int64_t mx = 0, used = 0;
    for (int i = 0; i < units; i++) {
        mx = std::max<int64_t>(mx, off[i + 1] - off[i]);
        used += off[i + 1] > off[i];
    }
    out[0] = ctl[0]; out[1] = ctl[1]; out[2] = mx; out[3] = used;
    return CHISEL_HIP_OK;
}
int chisel_hip_kat_color_any(unsigned *mismatches) {
    unsigned *d = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof(unsigned)));
    HIP_TRY(hipMemset(d, 0, sizeof(unsigned)));
    hipLaunchKernelGGL(kat_color_any_kernel, dim3(256 * 256 * 256 / 256), dim3(256), 0, 0, d);
    HIP_TRY(hipMemcpy(mismatches, d, sizeof(unsigned), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return CHISEL_HIP_OK;
}
int chisel_hip_kat_reciprocal(unsigned long long *mismatches, unsigned *example_bits) {
    unsigned long long *d = nullptr;
    unsigned *e = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof(unsigned long long)));
    HIP_TRY(hipMalloc(&e, sizeof(unsigned)));
    HIP_TRY(hipMemset(d, 0, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(e, 0, sizeof(unsigned)));
    unsigned lo, hi;
    const float fmin = FASTZ_MIN, fmax = FASTZ_MAX;
    memcpy(&lo, &fmin, 4);
    memcpy(&hi, &fmax, 4);
    hipLaunchKernelGGL(kat_reciprocal_kernel, dim3(4096), dim3(256), 0, 0, lo, (unsigned long long)(hi - lo) + 1ull, d, e);
    HIP_TRY(hipMemcpy(mismatches, d, sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(example_bits, e, sizeof(unsigned), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    (void)hipFree(e);
    return CHISEL_HIP_OK;
}
int chisel_hip_kat_floor(unsigned long long *mismatches, unsigned *example_bits) {
    unsigned long long *d = nullptr;
    unsigned *e = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof(unsigned long long)));
    HIP_TRY(hipMalloc(&e, sizeof(unsigned)));
    HIP_TRY(hipMemset(d, 0, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(e, 0, sizeof(unsigned)));
    hipLaunchKernelGGL(kat_floor_kernel, dim3(4096), dim3(256), 0, 0, d, e);
    HIP_TRY(hipMemcpy(mismatches, d, sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(example_bits, e, sizeof(unsigned), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    (void)hipFree(e);
    return CHISEL_HIP_OK;
}
// host-side frustum arithmetic of the product (host_frustum.h), for CPU-only tests against the oracle
// the candidate ids cull_kernel hands to shard `rank` of `n_shards` for an id range (host evaluation of CullSpace; no GPU needed):
// returns the number of slots, writes the ids of the slots that hold one (at most `capacity`), *count = how many do
int chisel_hip_debug_cull_space(const int range_min[3], const int range_dim[3], int n_shards, int shard_rank, int shard_block, int *ids,
                                int capacity, int *count) {
    CullParams P;
    memset(&P, 0, sizeof(P));
    for (int a = 0; a < 3; a++) {
        P.range_min[a] = range_min[a];
        P.range_dim[a] = range_dim[a];
    }
    P.ip.n_shards = n_shards;
    P.ip.shard_rank = shard_rank;
    P.ip.shard_block = shard_block;
    const CullSpace space(P);
    int n = 0;
    for (int c = 0; c < space.total; c++) {
        int x, y, z;
        if (!space.id(P, c, x, y, z)) continue;
        if (n < capacity) {
            ids[3 * n] = x; ids[3 * n + 1] = y; ids[3 * n + 2] = z;
        }
        n++;
    }
    *count = n;
    return space.total;
}
int chisel_hip_frustum(const float pose[12], float fy, float cy, int width, int height, float near_plane, float far_plane, float *corners,
                       float *lines, float *planes) {
    if (!pose || width <= 0 || height <= 0) return fail(CHISEL_HIP_ERR_INVALID, "bad frustum arguments");
    const hostmath::FrustumRange fr = hostmath::frustum_range(pose, near_plane, far_plane, fy, cy, width, height, 16, 1.0f);
    if (corners) memcpy(corners, fr.corners, sizeof(fr.corners));
    if (planes) memcpy(planes, fr.planes, sizeof(fr.planes));
    if (lines) {
        // far face, near face, connecting edges (Frustum.cpp:190-217)
        static const int idx[24] = {0, 1, 3, 2, 1, 3, 2, 0, 4, 7, 6, 5, 5, 7, 6, 4, 0, 5, 1, 6, 2, 7, 3, 4};
        for (int i = 0; i < 24; i++) memcpy(lines + 3 * i, fr.corners + 3 * idx[i], 3 * sizeof(float));
    }
    return CHISEL_HIP_OK;
}
int chisel_hip_frustum_from_vectors(const float forward[3], const float pos[3], const float right[3], const float up[3], float near_plane, float far_plane,
                                    float fov, float aspect, float *corners, float *lines, float *planes) {
    if (!forward || !pos || !right || !up) return fail(CHISEL_HIP_ERR_INVALID, "bad frustum arguments");
    float pl[24], co[24];
    hostmath::frustum_from_vectors(hostmath::mk(forward[0], forward[1], forward[2]), hostmath::mk(pos[0], pos[1], pos[2]), hostmath::mk(right[0], right[1], right[2]),
                                   hostmath::mk(up[0], up[1], up[2]), near_plane, far_plane, fov, aspect, pl, co);
    if (corners) memcpy(corners, co, sizeof(co));
    if (planes) memcpy(planes, pl, sizeof(pl));
    if (lines) {
        static const int idx[24] = {0, 1, 3, 2, 1, 3, 2, 0, 4, 7, 6, 5, 5, 7, 6, 4, 0, 5, 1, 6, 2, 7, 3, 4};  // Frustum.cpp:190-217
        for (int i = 0; i < 24; i++) memcpy(lines + 3 * i, co + 3 * idx[i], 3 * sizeof(float));
    }
    return CHISEL_HIP_OK;
}
int chisel_hip_debug_frustum_range(const float *pose, float near_plane, float far_plane, float fy, float cy, int W, int H,
                                   int chunk_n, float res, int *range_min3, int *range_dim3, float *planes24, float *corners24) {
    hostmath::FrustumRange fr = hostmath::frustum_range(pose, near_plane, far_plane, fy, cy, W, H, chunk_n, res);
    memcpy(range_min3, fr.range_min, sizeof(fr.range_min));
    memcpy(range_dim3, fr.range_dim, sizeof(fr.range_dim));
    if (planes24) memcpy(planes24, fr.planes, sizeof(fr.planes));
    if (corners24) memcpy(corners24, fr.corners, sizeof(fr.corners));
    return CHISEL_HIP_OK;
}

}  // extern "C"
