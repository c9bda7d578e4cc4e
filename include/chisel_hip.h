/*
 * chisel_hip.h -- C ABI of libchisel_hip.so, the MI355X (gfx950) dense-TSDF backend that replaces the
 * CPU hot path of the OpenChisel library vendored in z619850002/CVIDS.
 *
 * Plain C: opaque handle, plain pointers and sizes, no Eigen / torch / STL types.  Every entry point
 * names the reference interface it replaces (paths relative to OpenChisel/open_chisel/ in the
 * reference tree).  The C++ facade in cvids_amd/open_chisel/ re-creates the reference's class
 * surface (chisel::Chisel, ProjectionIntegrator, ChunkManager, Chunk, Mesh) on top of this ABI;
 * INTEGRATION.md shows the binding a CVIDS maintainer adds.
 *
 * Conventions
 *   - all functions return a chisel_hip_status (0 = OK) unless documented otherwise;
 *     chisel_hip_last_error() gives the message of the calling thread's last failure.
 *   - poses are camera->world rigid transforms, row-major 3x4 [R|t] (the `Transform` /
 *     Eigen::Affine3f the reference passes; optical frame: z forward, x right, y down).
 *   - chunk ids are int[3] (chisel::ChunkID = Eigen::Vector3i); voxel i of a chunk is
 *     (z*N + y)*N + x (Chunk.h:81-84).
 *   - integrate calls are asynchronous on the map's HIP stream(s); every query / download
 *     synchronises first, so the observable behaviour is the reference's synchronous one.
 *   - image pointers may be host or device memory (`on_device`); device images must be complete
 *     when the call is made (or ordered with chisel_hip_set_stream / chisel_hip_wait_event) and stay
 *     valid until the map has consumed them (chisel_hip_synchronize / chisel_hip_record_event).
 *     Host images in pageable memory are copied during the call.  Host images in page-locked memory
 *     (hipHostMalloc / hipHostRegister) are read by the device when the batch runs -- depth straight
 *     over the bus by the first kernel, colour through an asynchronous copy -- so they too must stay
 *     untouched until the map has consumed them.
 *   - a map is driven from one thread at a time, as the reference is (chisel_ros: one ros::spin thread).
 */
#ifndef CHISEL_HIP_H_
#define CHISEL_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an export is removed or re-typed, a struct of this header changes its layout, or one of the array-length macros below
 * (CHISEL_HIP_NUM_*) grows: a client built against another value must not call into the library (chisel_hip_abi_version() tells; the
 * facade's chisel::Chisel constructor and cvids_amd/capi.py refuse a mismatch).
 *   1  rounds 1-4
 *   2  round 5-6: chisel_hip_mesh_shell_plan_all removed, CHISEL_HIP_NUM_LAUNCH_STATS 8 -> 10, the device-plan and incremental
 *      meshesToUpdate entries added; later additions within 2 (nothing removed or re-typed): chisel_hip_pool_info,
 *      chisel_hip_frustum_from_vectors, chisel_hip_order_stream_after_map / _map_after_stream, the wait-free sharded recompute
 *      (chisel_hip_shell_plan_queue, _import_shells_fixed, _shell_commit) */
#define CHISEL_HIP_ABI_VERSION 2

typedef struct chisel_hip_map chisel_hip_map; /* opaque: one TSDF map (or one shard of it) on one GPU */

typedef enum {
    CHISEL_HIP_OK = 0,
    CHISEL_HIP_ERR_INVALID = 1,     /* bad argument (null, non-cubic chunk, bad channel count ...)        */
    CHISEL_HIP_ERR_HIP = 2,         /* a HIP runtime call failed / no gfx950 device                        */
    CHISEL_HIP_ERR_POOL_FULL = 3,   /* chunk pool or hash exhausted (a fixed pool, or a growing one at its limit) */
    CHISEL_HIP_ERR_NOT_FOUND = 4,   /* chunk / mesh absent (the reference throws std::out_of_range)        */
    CHISEL_HIP_ERR_UNSUPPORTED = 5, /* feature outside the supported envelope                             */
    CHISEL_HIP_ERR_IO = 6           /* file could not be written                                          */
} chisel_hip_status;

/* truncation/{ConstantTruncator.h:48-51, InverseTruncator.h:42-53, QuadraticTruncator.h:42-67} */
typedef enum {
    CHISEL_HIP_TRUNC_CONSTANT = 0,  /* param = truncation distance [m]  */
    CHISEL_HIP_TRUNC_INVERSE = 1,   /* param = scalingFactor (the one ChiselNode.cpp:98 instantiates) */
    CHISEL_HIP_TRUNC_QUADRATIC = 2  /* param = scalingFactor */
} chisel_hip_truncator_kind;

/* Chisel::Chisel(const Eigen::Vector3i& chunkSize, float voxelResolution, bool useColor) Chisel.h:41,
 * ChunkManager::ChunkManager(...) ChunkManager.cpp:44-48 */
typedef struct {
    int chunk_size[3];       /* voxels per chunk edge; must be cubic: 8, 16 or 32                      */
    float voxel_resolution;  /* metres                                                                  */
    int use_color;           /* allocate RGBW voxels (ColorVoxel.h:95-98)                               */
    int device_id;           /* HIP device ordinal, -1 = current device                                 */
    int64_t max_chunks;      /* chunk pool: > 0 = exactly that many chunks (more: CHISEL_HIP_ERR_POOL_FULL); 0 = about 6 GiB of
                              * voxel payload to begin with, GROWING with the scene as the reference's unordered_map of heap chunks
                              * does (ChunkManager.h:40-55; up to 16 times that or three quarters of the device's memory); < 0 =
                              * -max_chunks chunks to begin with, growing likewise.  See chisel_hip_pool_info.  */
    int n_shards;            /* spatial sharding of the chunk hash over GPUs: number of shards (>=1)    */
    int shard_rank;          /* this map's shard, 0 <= shard_rank < n_shards                            */
    int shard_block;         /* ownership super-block edge in chunks, 0 = default (2)                   */
} chisel_hip_config;

/* ProjectionIntegrator(truncator, weighter, carvingDist, enableCarving, centroids) ProjectionIntegrator.h:43-44;
 * ChiselServer::SetupProjectionIntegrator chisel_ros/src/ChiselServer.cpp:480-487 */
typedef struct {
    int truncator_kind;      /* chisel_hip_truncator_kind                                               */
    float truncator_param;
    float weight;            /* ConstantWeighter(weight) weighting/ConstantWeighter.h:35-46            */
    int carving_enabled;     /* ProjectionIntegrator::SetCarvingEnabled                                 */
    float carving_dist;      /* ProjectionIntegrator::SetCarvingDist                                    */
} chisel_hip_integrator;

/* DepthImage<float> (camera/DepthImage.h:33-103) + PinholeCamera (camera/PinholeCamera.h:35-69) + Transform */
typedef struct {
    const float *depth;      /* width*height row-major fp32 metres, NaN = invalid                       */
    int width, height;
    int on_device;           /* 0: host pointer (copied in), 1: device pointer (used in place)          */
    float pose[12];
    float fx, fy, cx, cy;    /* Intrinsics.h:40-47                                                      */
    float near_plane, far_plane;
} chisel_hip_depth_frame;

/* ColorImage<uint8_t> (camera/ColorImage.h:38-134): 1 = mono, 2, 3 = BGR, 4 = BGRA */
typedef struct {
    const uint8_t *color;
    int width, height, channels;
    int on_device;
    float pose[12];
    float fx, fy, cx, cy;
} chisel_hip_color_frame;

/* PointCloud (pointcloud/PointCloud.h:33-82) + the arguments of Chisel::IntegratePointCloud (Chisel.h:57) */
typedef struct {
    const float *points;     /* n_points x (x, y, z) in the sensor frame (PointCloud::GetPoints)              */
    const float *colors;     /* n_points x (r, g, b) in [0, 1] (PointCloud::GetColors), or NULL = no colours  */
    int64_t n_points;
    int on_device;           /* both arrays in device memory (complete, or ordered by stream / event)         */
    float pose[12];          /* extrinsic: sensor -> world, row-major 3x4                                      */
    float truncation;        /* segment half-length of the chunk enumeration (ChiselServer.cpp:523: 0.1)      */
    float max_dist;          /* points farther from the sensor list no chunk (ChiselServer.cpp:523: far plane) */
} chisel_hip_pointcloud;

/* per-map accumulated voxel counters; index into the array of chisel_hip_get_counters */
enum {
    CHISEL_HIP_CNT_SDF = 0,       /* DistVoxel::Integrate executions                                    */
    CHISEL_HIP_CNT_COL = 1,       /* ColorVoxel::Integrate executions                                   */
    CHISEL_HIP_CNT_COL_SAT = 2,   /* in band, on the colour image, colour weight already >= 8           */
    CHISEL_HIP_CNT_PROBE = 3,     /* voxels of resident chunks that took the carve test                 */
    CHISEL_HIP_CNT_CARVED = 4,    /* Carve() / weight decay executions                                  */
    CHISEL_HIP_CNT_WORK_CHUNKS = 5, /* chunks the integration kernel visited                            */
    CHISEL_HIP_CNT_NEW_CHUNKS = 6,  /* chunks allocated                                                 */
    CHISEL_HIP_CNT_UPDATED_CHUNKS = 7, /* chunks whose integration reported "updated"                   */
    CHISEL_HIP_CNT_FRAMES = 8,
    CHISEL_HIP_NUM_COUNTERS = 9
};

/* kernels timed by the built-in hipEvent profiler (chisel_hip_set_profiling) */
enum {
    CHISEL_HIP_KERNEL_PYRAMID = 0,   /* depth min/max pyramid                                           */
    CHISEL_HIP_KERNEL_CULL = 1,      /* candidate enumeration + conservative culling + compaction        */
    CHISEL_HIP_KERNEL_INTEGRATE = 2, /* projective SDF/weight/colour integration (+ allocation)          */
    CHISEL_HIP_KERNEL_MESH = 3,      /* marching cubes (count + emit)                                   */
    CHISEL_HIP_KERNEL_RESOLVE = 4,   /* hash lookup of the candidates -> work-list                       */
    CHISEL_HIP_KERNEL_CLOUD = 5,     /* point-cloud fusion mode (all of its kernels)                     */
    CHISEL_HIP_NUM_KERNELS = 6
};

/* ---- life cycle ------------------------------------------------------------------------------------- */
int chisel_hip_abi_version(void);
const char *chisel_hip_last_error(void);
/* number of visible gfx950 devices (0 when there is none) */
int chisel_hip_device_count(void);
/* Image buffers of the caller (DepthImage.h:42-52 / ColorImage.h:44-58: `new DataType[...]`, allocated once by chisel_ros and refilled
 * every frame, ChiselServer.cpp:268-273,287-292): page-locked host memory, which the integrate calls read without the runtime's staged
 * copy of pageable memory (depth straight over the bus, colour as one asynchronous copy).  The facade's DepthImage / ColorImage allocate
 * through these; plain malloc / free when no HIP device is present (the buffer is then ordinary memory -- nothing computes on the CPU). */
void *chisel_hip_host_alloc(size_t bytes);
void chisel_hip_host_free(void *ptr);
/* Chisel::Chisel Chisel.h:41 */
int chisel_hip_create(const chisel_hip_config *config, chisel_hip_map **out);
int chisel_hip_destroy(chisel_hip_map *map);
/* Chisel::Reset Chisel.cpp:44-48 (+ ChunkManager::Reset ChunkManager.cpp:176-180) */
int chisel_hip_reset(chisel_hip_map *map);
/* ProjectionIntegrator setters ProjectionIntegrator.h:185-217 */
int chisel_hip_set_integrator(chisel_hip_map *map, const chisel_hip_integrator *integrator);
/* run the map's kernels on a caller-owned hipStream_t (NULL = the map's own stream) */
int chisel_hip_set_stream(chisel_hip_map *map, void *hip_stream);
int chisel_hip_synchronize(chisel_hip_map *map);
/* Ordering against work the caller queued on OTHER streams, without blocking the host (no reference counterpart: the
 * reference's images are host buffers).  wait_event: the device frames of the next integrate call are complete once
 * `hip_event` (a hipEvent_t the caller recorded behind their producer, e.g. an RCCL all-gather) has completed.
 * record_event: records `hip_event` behind everything queued on the map so far -- once it has completed the frames of
 * all earlier integrate calls have been read and their buffers may be overwritten. */
int chisel_hip_wait_event(chisel_hip_map *map, void *hip_event);
int chisel_hip_record_event(chisel_hip_map *map, void *hip_event);
/* The same two orderings for a caller that has a STREAM rather than events (hipStream_t as void*), one call each, with events the map keeps:
 * order_stream_after_map: whatever `stream` is given next starts after what the map has queued so far; order_map_after_stream: the map's
 * next call starts after what `stream` has been given so far.  Nothing is waited for. */
int chisel_hip_order_stream_after_map(chisel_hip_map *map, void *stream);
int chisel_hip_order_map_after_stream(chisel_hip_map *map, void *stream);

/* ---- the hot path ------------------------------------------------------------------------------------ */
/* Chisel::IntegrateDepthScan<float> Chisel.h:59-112 -> ProjectionIntegrator::Integrate ProjectionIntegrator.h:51-99
 * (candidate enumeration ChunkManager.cpp:182-212, allocation :171-174, GarbageCollect Chisel.cpp:61-67) */
int chisel_hip_integrate_depth(chisel_hip_map *map, const chisel_hip_depth_frame *frame);
/* Chisel::IntegrateDepthScanColor<float,uint8_t> Chisel.h:114-213 -> IntegrateColor ProjectionIntegrator.h:101-183 */
int chisel_hip_integrate_depth_color(chisel_hip_map *map, const chisel_hip_depth_frame *frame,
                                     const chisel_hip_color_frame *color);
/* Chisel::IntegratePointCloud Chisel.cpp:107-157 -> ChunkManager::GetChunkIDsIntersecting(cloud) ChunkManager.cpp:214-257,
 * ProjectionIntegrator::Integrate(cloud) ProjectionIntegrator.cpp:38-173, Raycast geometry/Raycast.cpp:35-128.  Every voxel
 * receives the updates of the rays that meet it in cloud order, as the reference's loop applies them.  Counters afterwards:
 * SDF, COL, PROBE (= ray cells inside listed chunks), CARVED, WORK_CHUNKS (= listed chunks), NEW_CHUNKS, UPDATED_CHUNKS.
 * Rays whose cell walk the reference would never finish (an axis steps past its end cell) stop at that point; rays with a
 * coordinate that is not finite meet no voxel.  CHISEL_HIP_ERR_UNSUPPORTED (at the next call that waits): more than 65536
 * chunks or 16 (chunk, point) pairs per point in one cloud, chunk ids beyond +-2^20. */
int chisel_hip_integrate_pointcloud(chisel_hip_map *map, const chisel_hip_pointcloud *cloud);
/* n frames in order (frame k+1 sees frame k's result, as n successive calls would); colors may be NULL.  Consecutive
 * frames of one image size share launch sets of up to 16 frames: the voxels of a chunk stay in registers across them. */
int chisel_hip_integrate_batch(chisel_hip_map *map, int n, const chisel_hip_depth_frame *frames,
                               const chisel_hip_color_frame *colors);
/* Chisel::GarbageCollect(const ChunkIDList&) Chisel.cpp:61-67 / ChunkManager::RemoveChunk(ChunkID) ChunkManager.h:99-108 */
int chisel_hip_garbage_collect(chisel_hip_map *map, const int *chunk_ids_xyz, int n);
/* Chisel::UpdateMeshes Chisel.cpp:50-59 (recompute on every 10th call unless force) ->
 * ChunkManager::RecomputeMeshes ChunkManager.cpp:130-169 */
int chisel_hip_update_meshes(chisel_hip_map *map, int force);

/* ---- queries (each synchronises) --------------------------------------------------------------------- */
/* ChunkManager::GetChunks().size() ChunkManager.h:67-70 */
int chisel_hip_num_chunks(chisel_hip_map *map, int64_t *out);
/* ids of all resident chunks, 3 ints each, ascending (x, then y, then z; the reference: the order of an unordered_map); writes
 * min(count, max_ids) ids, *count = total */
int chisel_hip_list_chunks(chisel_hip_map *map, int *ids_xyz, int64_t max_ids, int64_t *count);
/* ChunkManager::HasChunk ChunkManager.h:79-82 */
int chisel_hip_has_chunk(chisel_hip_map *map, const int id_xyz[3], int *out);
/* Chunk::GetVoxels / GetColorVoxels Chunk.h:66,117-120: N^3 sdf, N^3 weight, 4*N^3 rgbw (may be NULL) */
int chisel_hip_download_chunk(chisel_hip_map *map, const int id_xyz[3], float *sdf, float *weight, uint8_t *rgbw);
/* inverse of download (ChunkManager::AddChunk ChunkManager.h:89-92 with caller-filled voxels) */
int chisel_hip_upload_chunk(chisel_hip_map *map, const int id_xyz[3], const float *sdf, const float *weight,
                            const uint8_t *rgbw);
/* Chisel::GetMeshesToUpdate Chisel.h:220-223: ids flagged since the last recompute (27-neighbourhoods) */
int chisel_hip_meshes_to_update(chisel_hip_map *map, int *ids_xyz, int64_t max_ids, int64_t *count);
/* The same set for a caller that keeps its own copy between calls (the facade's Chisel::GetMeshesToUpdate, read after every frame:
 * ChiselServer.cpp:346): the ids that joined Chisel::meshesToUpdate (Chisel.h:175-189, :228) since the state `cursor` stands for -- two
 * words, zero before the first call, updated by the call.  *cleared != 0: the set was emptied in between (Chisel::UpdateMeshes' recompute,
 * Chisel.cpp:57, or Reset): the caller empties its copy first.  When more than max_ids ids are due nothing is consumed: *count says how
 * many, call again with room for them.  Cost: proportional to what changed since the cursor, not to the map. */
int chisel_hip_meshes_to_update_since(chisel_hip_map *map, uint64_t cursor[2], int *ids_xyz, int64_t max_ids, int64_t *count, int *cleared);
/* queues that listing behind the integration just handed over, without waiting: the caller's chisel_hip_synchronize then covers it and the
 * _since call that follows (same cursor, nothing integrated in between) neither launches nor waits */
int chisel_hip_meshes_to_update_prefetch(chisel_hip_map *map, const uint64_t cursor[2]);
/* ChunkManager::GetAllMeshes ChunkManager.h:163-166 */
int chisel_hip_num_meshes(chisel_hip_map *map, int64_t *out);
int chisel_hip_list_meshes(chisel_hip_map *map, int *ids_xyz, int64_t max_ids, int64_t *count);
/* Mesh (mesh/Mesh.h:54-58) sizes of one chunk's mesh: vertices (= normals = colors = indices), grids */
int chisel_hip_mesh_size(chisel_hip_map *map, const int id_xyz[3], int64_t *n_vertices, int64_t *n_grids);
/* 3 floats per vertex / grid; any pointer may be NULL; indices are 0..n_vertices-1 (MarchingCubes.h:92-94) */
int chisel_hip_download_mesh(chisel_hip_map *map, const int id_xyz[3], float *vertices, float *normals,
                             float *colors, float *grids);
/* ChunkManager::GetSDF ChunkManager.cpp:476-499 ; *found = 0 when the reference returns false */
int chisel_hip_get_sdf(chisel_hip_map *map, const float pos[3], double *dist, int *found);
/* ChunkManager::GetSDFAndGradient ChunkManager.cpp:449-474 */
int chisel_hip_get_sdf_and_gradient(chisel_hip_map *map, const float pos[3], double *dist, float grad[3], int *found);
/* Chisel::SaveAllMeshesToPLY Chisel.cpp:69-105 + SaveMeshPLYASCII io/PLY.cpp:29-88 */
int chisel_hip_save_ply(chisel_hip_map *map, const char *path);
/* SaveMeshPLYASCII(fileName, mesh) (src/io/PLY.cpp:29-88) for ONE mesh the caller holds: the same text format; colors (3 floats in
 * [0, 1] per vertex) may be NULL; indices: three per face, as Mesh::indices.  Host I/O only. */
int chisel_hip_write_mesh_ply(const char *path, const float *vertices, const float *colors, int64_t n_vertices, const int64_t *indices,
                              int64_t n_indices);

/* ---- meshing a sharded map (SURVEY.md 8e "meshing across shards") -------------------------------------------------
 * A chunk's mesh reads its 26 neighbours (cube corners: ChunkManager.cpp:316-357; gradient normals and colours of border
 * vertices: :449-499, :588-607), most of which belong to other shards.  The owners export those chunks, the mesher
 * imports them as "ghost" chunks -- resident, never integrated (cull_kernel only takes chunks this shard owns) --
 * recomputes the meshes of its own chunks and drops the ghosts again.  cvids_amd/sharded.py orchestrates the exchange
 * (ShardedChisel.UpdateMeshes).  ids must not repeat within a call.  Buffers: [n][V] floats / [n][V][4] bytes, on the
 * host or (on_device != 0) in HBM on the map's device.
 *   export_chunks        voxels of n chunks (found[j] = 0 and default voxels for a chunk that is not resident)
 *   import_ghost_chunks  install the chunks with found[j] != 0 (found may be NULL = all) as ghosts
 *   drop_ghost_chunks    remove every ghost imported since the last drop
 *   update_meshes_of     RecomputeMeshes (ChunkManager.cpp:130-169) for exactly these chunk ids -- the shard's share of
 *                        the union of all shards' meshesToUpdate -- then meshesToUpdate.clear() (Chisel.cpp:57) */
int chisel_hip_export_chunks(chisel_hip_map *map, const int *ids_xyz, int n, float *sdf, float *weight, uint8_t *rgbw,
                             int *found, int on_device);
int chisel_hip_import_ghost_chunks(chisel_hip_map *map, const int *ids_xyz, int n, const float *sdf, const float *weight,
                                   const uint8_t *rgbw, const int *found, int on_device);
int chisel_hip_drop_ghost_chunks(chisel_hip_map *map);
int chisel_hip_update_meshes_of(chisel_hip_map *map, const int *ids_xyz, int n);

/* ---- the step before the path ------------------------------------------------------------------------
 * CollaborativeServer::PublishDenseInfo's image conditioning (server_pose_graph/src/collaborative_server_system.cpp:
 * 199-276): cv::resize of the depth map (CV_64F) and the colour image (CV_8UC1 / CV_8UC3) to the publish size (640 x 480
 * there, :213-214), and for depth the narrowing to float (:255), NaN for readings < 0.1 or > 20 (:262-265) and the rescaled
 * intrinsics fx, fy, cx, cy (:216-219).  cv::resize with its default INTER_LINEAR is restated from OpenCV's published
 * algorithm (imgproc/resize.cpp without IPP): scale = 1. / (dst / (double)src); tap position (d + 0.5) * scale - 0.5 narrowed
 * to float; along x out-of-image taps clamp with weight 0 and the last column's horizontal pass is S[sx] * ONE; along y the rows
 * are clipped and the weights kept; 64-bit floats use float weights and double sums; 8-bit images use short weights
 * (round-to-even of weight * 2048), an int horizontal pass and uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2)
 * vertically; halving both axes exactly is INTER_AREA (mean of the 2 x 2 block: double sum * 0.25f, 8-bit (sum + 2) >> 2); an
 * image that already has the publish size is copied.  src: w0 x h0 (x channels, interleaved), dst: w x h, each on the host or
 * (flag) in HBM; the results feed chisel_hip_integrate_* directly.  Parity unpinned: OpenCV is not available here to generate
 * vectors; hand-computed cases of the rules above are in tests/test_publish_dense.py. */
int chisel_hip_condition_depth(const double *src, int w0, int h0, int src_on_device, float *dst, int w, int h, int dst_on_device,
                               double intrinsics_fx_fy_cx_cy[4], void *hip_stream);
int chisel_hip_condition_color(const uint8_t *src, int w0, int h0, int channels, int src_on_device, uint8_t *dst, int w, int h,
                               int dst_on_device, void *hip_stream);
/* CollaborativeServer::SendPointCloud (collaborative_server_system.cpp:318-381), the third thing PublishDenseInfo sends (:249): the
 * data array of its organised sensor_msgs::PointCloud2 -- w * h points of 16 bytes {float x = column, float y = row, float z =
 * (float)depth, int32 rgb = grey byte replicated}, all four words NaN unless 0.1 < z < 10.  depth: w x h doubles (the resized
 * map); color: the resized colour image, color_step bytes per row; the grey byte is the one at byte offset `column` of the row,
 * as mColorImage.at<uint8_t>(u, v) reads it whatever the channel count.  Both inputs on the host or both (flag) in HBM. */
int chisel_hip_publish_cloud(const double *depth, const uint8_t *color, int w, int h, int color_step, int src_on_device, void *points,
                             int dst_on_device, void *hip_stream);

/* ---- the step before that: the inverse-depth filter ----------------------------------------------
 * DepthFilter (server_pose_graph/src/dense_mapping/depth_filter.cpp): per-pixel Gaussian x uniform mixture filter of the
 * inverse depth a stereo matcher delivers; its state (a, b, mu, cov: four CV_64F maps) lives in HBM here.
 *   create   DepthFilter::DepthFilter(height, width)            depth_filter.cpp:130-142
 *   update   DepthFilter::Update(mUpdateMu, mUpdateCov)         depth_filter.cpp:177-259 (NormPdf :10-16)
 *            cov == NULL: cov_all for every pixel (depth_estimator.cpp:293); reciprocal != 0: the update is 1.0 / mu[i]
 *            (depth_estimator.cpp:286 fused in).  Arrays of width * height doubles, on the host or (flag) in HBM.
 *   read     which = 0 GetA, 1 GetB, 2 GetInvDepth, 3 GetCov, 4 GetRatio (depth_filter.h:66-86), 5 the inverse-depth map
 *            DepthEstimator keeps (1e-5 where the ratio is below 0.5, depth_estimator.cpp:387-398), 6 the depth map 1.0 / that
 *            (server_keyframe.cpp:1117) -- the input of chisel_hip_condition_depth, so depth can stay in HBM from the
 *            matcher to the TSDF.
 * All arithmetic in double in the reference's order; exp() is the device library's (parity unpinned, tolerance in
 * tests/test_gpu_filter.py).  PropogateDepth is not built: its only call site is commented out (server_pose_graph.cpp:891). */
typedef struct chisel_hip_depth_filter chisel_hip_depth_filter;
int chisel_hip_depth_filter_create(int height, int width, int device_id, chisel_hip_depth_filter **out);
int chisel_hip_depth_filter_destroy(chisel_hip_depth_filter *filter);
int chisel_hip_depth_filter_update(chisel_hip_depth_filter *filter, const double *mu, const double *cov, double cov_all, int reciprocal,
                                   int on_device);
int chisel_hip_depth_filter_read(chisel_hip_depth_filter *filter, int which, double *dst, int dst_on_device);

/* Binary dump / restore of the whole map (the correct counterpart of chisel_ros FillChunkMessage,
 * Serialization.h:31-84, whose bit packing loses data; also checkpoint / resume).  File: 32-byte header
 * {"CHSLHIP1", int32 chunk edge, float resolution, int32 has_colour, 4 spare bytes, int64 n_chunks}, then per chunk, in
 * ascending id order: int32 id[3], float sdf[V], float weight[V], (uint8 rgbw[4 V] if has_colour).  load replaces the
 * map's contents (Reset first); chunk size, resolution and colour must match the map's. */
int chisel_hip_save_map(chisel_hip_map *map, const char *path);
int chisel_hip_load_map(chisel_hip_map *map, const char *path);

/* ---- one map over several GPUs of the node, inside one process -------------------------------------------------------- */
/* The reference has one chisel::Chisel object per map (Chisel.h:38-230) and one calling thread; a C++ caller that links this
 * library (chisel_ros) cannot start one process per GPU.  chisel_hip_create_group returns a handle that every entry point of
 * this header accepts like any other map: behind it one shard map per entry of device_ids[] (n_shards = n_devices, shard i on
 * device_ids[i]; chunk ownership = chisel_hip_chunk_owner; a device may be named several times).  The library hands every
 * shard every frame (device frames are copied peer-to-peer to the shards on other devices), all shards integrate concurrently,
 * chisel_hip_update_meshes exchanges the neighbour chunks between the shards, queries go to the owner, listings / PLY / map
 * dumps are merged in ascending id order and equal a single map's.  cfg->device_id, n_shards and shard_rank are ignored.
 * Not available on a group: chisel_hip_set_stream, chisel_hip_record_event (one stream / event cannot span GPUs: use
 * chisel_hip_synchronize), and the shard-to-shard calls (export / import / drop ghost chunks, update_meshes_of). */
int chisel_hip_create_group(const chisel_hip_config *cfg, const int *device_ids, int n_devices, chisel_hip_map **out);

/* ---- view frustum (host arithmetic, no GPU needed) -------------------------------------------------------------- */
/* PinholeCamera::SetupFrustum (src/camera/PinholeCamera.cpp:55-59) -> Frustum::SetFromParams / SetFromVectors
 * (src/geometry/Frustum.cpp:143-219), with the reference's quirks (fy is used for both focal lengths, cx is ignored) and fp32
 * operation order: the frustum the integration enumerates its candidate chunks from, for the caller's use
 * (chisel_ros draws it: ChiselServer.cpp:97-134).
 *   corners[8][3]   farTopLeft, farTopRight, farBotLeft, farBotRight, nearBotRight, nearTopLeft, nearTopRight, nearBotLeft
 *                   (Frustum::GetCorners, Frustum.cpp:181-188)
 *   lines[24][3]    the 12 edges as point pairs in the order of Frustum::GetLines (Frustum.cpp:190-217)
 *   planes[6][4]    far, near, top, bottom, left, right: normalised normal xyz + offset as Plane(p1, p2, p3) leaves them
 *                   (src/geometry/Plane.cpp:44-52: the offset is NOT divided by the normal's length)
 * Any of the three outputs may be NULL. */
int chisel_hip_frustum(const float pose_c2w[12], float fy, float cy, int width, int height, float near_plane, float far_plane,
                       float *corners, float *lines, float *planes);
/* Frustum::SetFromVectors (src/geometry/Frustum.cpp:155-219) itself, for a caller that has its own view vectors (what
 * Frustum::SetFromOpenGLViewProjection, :124-141, ends in): same outputs, same fp32 operation order. */
int chisel_hip_frustum_from_vectors(const float forward[3], const float pos[3], const float right[3], const float up[3], float near_plane,
                                    float far_plane, float fov, float aspect, float *corners, float *lines, float *planes);

/* ---- meshing a sharded map with shells -----------------------------------------------------------------
 * What a chunk's mesh reads of a neighbour chunk is a shell one or two voxels thick (cube corners, gradients around the vertices, the
 * nearest voxel's colour: SURVEY.md 8e's "faces"), not the whole chunk.  A "box code" names the part of a ghost chunk that travels:
 * two bits per axis (x: bits 0-1, y: 2-3, z: 4-5), 0 = every coordinate, 1 = {0, 1}, 2 = {N - 1}, 3 = {0, 1, N - 1}; chisel_hip_shell_volume gives its
 * number of voxels; the payload of a list of items (x, y, z, box) is the concatenation of their boxes in z, y, x order.
 *   chisel_hip_dirty_ids_device  the chunks updated since the last recompute as a DEVICE int array: out[0] = n, then n entries
 *                                (x, y, z, flag) -- flag 0: the chunk was updated (its 27-neighbourhood is meshesToUpdate, Chisel.h:175-189),
 *                                flag 1: an entry of meshesToUpdate kept on the host; nothing is waited for (record_event orders the
 *                                collective that gathers the ranks' arrays)
 *   chisel_hip_mesh_shell_plan   host arithmetic, identical on every rank: from the gathered entries the ids `rank` meshes (jobs) and
 *                                the ghosts it needs as items (owner, x, y, z, box) -- one or more boxes per ghost, ascending by owner, id, box;
 *                                chisel_hip_import_ghost_shells creates a ghost once, from its first item --,
 *                                every rank can evaluate it for every other rank, so the request lists need no exchange
 *   chisel_hip_export_shells     the boxes of the listed chunks of this shard, packed (device pointers with on_device: no wait);
 *                                found[j] = 0 and default voxels for a chunk that is not resident
 *   chisel_hip_import_ghost_shells  installs them as ghost chunks (only the box is written; honours chisel_hip_wait_event; with
 *                                on_device nothing is allocated or waited for: queued on the map's stream);
 *                                chisel_hip_drop_ghost_chunks removes them again (queued as well) */
int chisel_hip_dirty_ids_device(chisel_hip_map *map, int *out_dev, int capacity);
int chisel_hip_mesh_shell_plan(const int *entries, int64_t n_entries, int n_shards, int rank, int shard_block, int *jobs, int64_t max_jobs,
                               int64_t *n_jobs, int *items, int64_t max_items, int64_t *n_items);
/* The sharded recompute without host planning (round 5; cvids_amd/sharded.py: ShardedChisel.UpdateMeshes): from the
 * all-gathered list of updated chunks (per rank 1 + 4 * cap ints: count, then (x, y, z, flag) entries -- chisel_hip_dirty_ids_device)
 * every shard derives ON THE DEVICE its own jobs, the shells it sends to every peer and how much it receives from each.
 * chisel_hip_shell_plan_device: out[0] = jobs of this shard, out[1] = ghost chunks its earlier recomputes created (a running total), out[2] = largest per-rank count of the list (> cap: gather again with more
 * room), out[3] = items sent, then (items, voxels) per peer sent, then per peer received -- the one host wait of a sharded recompute.
 * chisel_hip_export_shells_packed writes one byte segment per peer, back to back in rank order (chisel_hip_shell_segment_bytes each:
 * heads, items that say where their voxels are, sdf | weight | rgbw), the caller's all-to-all moves them, chisel_hip_import_shells_packed
 * turns what arrived into ghost chunks (the buffer stays untouched until chisel_hip_drop_ghost_chunks), chisel_hip_update_meshes_planned
 * recomputes the plan's jobs.  No reference counterpart: the reference has one process and one map (Chisel.h:150-195). */
int chisel_hip_shell_plan_device(chisel_hip_map *map, const int *gathered_dev, int world, int capacity, int64_t *out);
int64_t chisel_hip_shell_segment_bytes(chisel_hip_map *map, int64_t items, int64_t voxels);
int chisel_hip_export_shells_packed(chisel_hip_map *map, void *out_dev, int64_t bytes);
int chisel_hip_import_shells_packed(chisel_hip_map *map, const void *in_dev, int64_t bytes);
int chisel_hip_update_meshes_planned(chisel_hip_map *map);
/* The same recompute with NO host wait (round 6; ShardedChisel.UpdateMeshes(wait_free=True)).  The host reads nothing of the plan: every
 * (sender, receiver) segment has `seg_stride` bytes to itself (a multiple of 16, agreed between the ranks beforehand -- from what the previous
 * recompute needed), the exchange is an all-to-all of equal splits, and the steps behind it read the heads of the received segments.
 *   chisel_hip_shell_plan_queue    queues the plan and, behind it, the export of `world` segments of seg_stride bytes into out_dev, whose first
 *                                  workgroup also writes this rank's STATUS into status_dev (CHISEL_HIP_SHELL_STATUS_INTS ints on the device):
 *                                  [0] bits that call the recompute off (1: a rank's dirty list exceeds `capacity`, 2: the plan's
 *                                  tables overflowed, 4: a segment exceeds seg_stride), [1] largest per-rank dirty count, [2] bytes of this
 *                                  rank's largest segment, [3] its jobs, [4] items it receives, [5] items it sends, [6] voxels it receives,
 *                                  [7] ghost chunks its earlier recomputes created.  The caller all-reduces the vector with MAX (in place)
 *                                  in front of the exchange.  send_items_hint: [5] of the previous recompute (a grid size; 0 = unknown)
 *   chisel_hip_import_shells_fixed ghosts from the received segments; then chisel_hip_update_meshes_planned and chisel_hip_drop_ghost_chunks
 *                                  as before -- all of it queued, and all of it a no-op ON THE DEVICE if word 0 of the all-reduced status
 *                                  is not zero (no ghost, no mesh, no dirty flag cleared: on every rank alike)
 *   chisel_hip_shell_commit        once the host has read the all-reduced status (any time before it next changes the map): settles the mesh
 *                                  step; aborted != 0: the recompute did not happen -- make it again with chisel_hip_shell_plan_device */
#define CHISEL_HIP_SHELL_STATUS_INTS 8
int chisel_hip_shell_plan_queue(chisel_hip_map *map, const int *gathered_dev, int world, int capacity, int64_t seg_stride, int *status_dev, void *out_dev,
                                int send_items_hint);
int chisel_hip_import_shells_fixed(chisel_hip_map *map, const void *in_dev, int64_t seg_stride, const int *status_dev, int jobs_hint, int items_hint);
int chisel_hip_shell_commit(chisel_hip_map *map, int aborted);
int64_t chisel_hip_shell_volume(int box, int chunk_edge);
int chisel_hip_export_shells(chisel_hip_map *map, const int *items, int n, float *sdf, float *weight, uint8_t *rgbw, int *found, int on_device);
int chisel_hip_import_ghost_shells(chisel_hip_map *map, const int *items, int n, const float *sdf, const float *weight, const uint8_t *rgbw,
                                   const int *found, int on_device);

/* ---- measurement ------------------------------------------------------------------------------------ */
/* accumulated since creation / last reset_counters; out has CHISEL_HIP_NUM_COUNTERS entries */
int chisel_hip_get_counters(chisel_hip_map *map, uint64_t *out, int reset_counters);
/* ChunkManager::PrintMemoryStatistics (src/ChunkManager.cpp:641-678; the reference calls it after every depth-only frame,
 * Chisel.h:111): the census of Chunk::ComputeStatistics (src/Chunk.cpp:89-116) over every resident chunk -- voxels with weight > 0 and
 * sdf < 0 / >= 0, voxels with weight <= 0 -- as one reduction kernel, the sum of the weights (a double here; the reference adds floats
 * in hash-map order), and the smallest / largest resident chunk id per axis, from which the caller forms the bounds
 * (min = chunk_size * id_min * resolution, max = chunk_size * (id_max + 1) * resolution, Chunk.cpp:65-70).  n_chunks == 0: the ids are
 * undefined. */
typedef struct chisel_hip_statistics {
    int64_t n_unknown, n_known_inside, n_known_outside;
    double total_weight;
    int64_t n_chunks;
    int32_t id_min[3], id_max[3];
} chisel_hip_statistics;
int chisel_hip_memory_statistics(chisel_hip_map *map, chisel_hip_statistics *out);
/* A counter that moves whenever chunks appear or disappear by anything other than integration (garbage collection, reset, upload, map
 * load, ghost import / drop): together with chisel_hip_num_chunks -- integration only ever adds chunks -- it tells a caller that keeps
 * mirrors of the chunk map (ChunkManager::GetChunks, ChunkManager.h:65-73) whether they are still current. */
int chisel_hip_topology_epoch(chisel_hip_map *map, uint64_t *out);
/* ChunkManager::GetChunkIDsIntersecting(const Frustum &, ChunkIDList *) (src/ChunkManager.cpp:182-212) for a frustum given as
 * chisel_hip_frustum returns it (corners[8][3], planes[6][4] in the order far, near, top, bottom, left, right): the ids in the
 * reference's order (x outer, z inner).  ids may be NULL (count only); at most max_ids are written. */
int chisel_hip_candidates(const float corners[24], const float planes[24], const int chunk_size[3], float voxel_resolution, int *ids,
                          int64_t max_ids, int64_t *count);
/* ChunkManager::GetChunkIDsIntersecting(const PointCloud &, const Transform &, float truncation, float maxDist, ChunkIDList *)
 * (src/ChunkManager.cpp:214-257): the chunks the segments point -+ truncation along the viewing rays pass through (Raycast over the
 * chunk grid, points further than max_dist skipped) -- the listing step of chisel_hip_integrate_pointcloud on its own, same kernel.
 * The reference returns them in the order of an unordered_map; here ascending (x, then y, then z).  A shard lists the chunks it owns.
 * ids may be NULL (count only); at most max_ids are written. */
int chisel_hip_cloud_candidates(chisel_hip_map *map, const chisel_hip_pointcloud *cloud, int *ids_xyz, int64_t max_ids, int64_t *count);
/* ChunkManager::ComputeNormalsFromGradients (src/ChunkManager.cpp:609-626; stages bit 0: a normal is overwritten where the gradient
 * lookup of its vertex succeeds, kept otherwise) and ChunkManager::ColorizeMesh / InterpolateColor (:628-639, :501-573; stages bit 1)
 * for a caller's own vertex list (host arrays of 3 n floats). */
int chisel_hip_shade_vertices(chisel_hip_map *map, const float *vertices, int64_t n, float *normals, float *colors, int stages);
/* ProjectionIntegrator::Integrate<DataType>(depthImage, camera, cameraPose, chunk) / IntegrateColor (ProjectionIntegrator.h:51-52,
 * :101-102): ONE frame into ONE resident chunk -- whether or not the frustum's id range holds it, as the reference's per-chunk call
 * knows nothing of frusta --; color may be NULL (the depth-only update rule).  *updated = the call's return value there ("some voxel
 * changed").  CHISEL_HIP_ERR_NOT_FOUND when the chunk is not resident. */
int chisel_hip_integrate_chunk(chisel_hip_map *map, const int id_xyz[3], const chisel_hip_depth_frame *frame, const chisel_hip_color_frame *color,
                               int *updated);
/* ChunkManager::ExtractInsideVoxelMesh / ExtractBorderVoxelMesh(chunk, index, coordinates, nextMeshIndex, mesh) (src/ChunkManager.cpp:
 * 259-379): the marching-cubes triangles of ONE cube of a resident chunk -- corner voxels index + cubeIndexOffsets, read from the
 * neighbouring chunk where a coordinate leaves [0, N) (index components -1 .. N-1), none when a corner is unobserved (weight <= 0.5) or
 * its chunk absent --, MarchingCubes::MeshCube with the caller's cube coordinates: up to 15 vertices (3 floats each) and their face
 * normals; *occupied: MarchingCubes::IsOccupied (the caller pushes a grid entry).  vertices / normals may be NULL. */
int chisel_hip_mesh_cube(chisel_hip_map *map, const int id_xyz[3], const int voxel_index[3], const float coordinates[3], float *vertices, float *normals,
                         int *n_vertices, int *occupied);
/* ChunkManager::RecomputeMesh(chunkID, mutex) (src/ChunkManager.cpp:91-128): the mesh of one chunk into ChunkManager::allMeshes, leaving
 * meshesToUpdate as it is (chisel_hip_update_meshes_of ends with the meshesToUpdate.clear() of Chisel::UpdateMeshes, Chisel.cpp:57) */
int chisel_hip_recompute_mesh(chisel_hip_map *map, const int id_xyz[3]);
/* ChunkManager::GenerateMesh(chunk, mesh) (src/ChunkManager.cpp:381-447) for one resident chunk into the caller's arrays -- marching
 * cubes with face normals; stages bit 0 adds ComputeNormalsFromGradients, bit 1 ColorizeMesh (stages 3 = what RecomputeMesh stores) --
 * without touching ChunkManager::allMeshes or meshesToUpdate.  Arrays of 3 floats per vertex / grid entry; n_vertices / n_grids are
 * always set, the arrays only when both capacities suffice (a 16^3 chunk has at most 15 * 4096 vertices and 4096 grid entries). */
int chisel_hip_generate_mesh(chisel_hip_map *map, const int id_xyz[3], int stages, int64_t capacity_vertices, int64_t capacity_grids,
                             float *vertices, float *normals, float *colors, float *grids, int64_t *n_vertices, int64_t *n_grids);
/* hipEvent pairs around every kernel launch on the map's stream (off by default) */
int chisel_hip_set_profiling(chisel_hip_map *map, int enable);
/* total milliseconds and launch counts per CHISEL_HIP_KERNEL_* since enabled / last reset */
int chisel_hip_get_profile(chisel_hip_map *map, double *ms_total, int64_t *launches, int reset_profile);
/* The public statics of marching_cubes/MarchingCubes.h:41-146 for a caller's own cube, on the device (no map involved; the current
 * HIP device):
 *   chisel_hip_mc_tables          triangleTable[256][16] (-1 terminated rows) and edgeIndexPairs[12][2] (MarchingCubes.cpp:29-302)
 *   chisel_hip_mesh_cube_values   vertex_coords: 3 x 8 column-major, vertex_sdf: 8 -> CalculateVertexConfiguration (:108-118),
 *                                 InterpolateEdgeVertices (:120-132; edge_coords 3 x 12 column-major, zeros where the reference leaves a
 *                                 column unset; may be null) and MeshCube(.., Mesh*) (:73-106): up to 15 vertices in push order
 *                                 (t + 2, t + 1, t) with each triangle's face normal thrice (vertices / normals: 45 floats each; may be null)
 *   chisel_hip_interpolate_vertex InterpolateVertex (:135-146), "vertex1 + 0.5 * vertex2" included
 * and geometry/Raycast.h:9 / Raycast.cpp:35-128: the cells of [min, max) the segment start -> end meets, in order (cells: capacity x 3
 * ints; *count = cells met, which may exceed capacity). */
int chisel_hip_mc_tables(int *triangle_table, int *edge_index_pairs);
int chisel_hip_mesh_cube_values(const float *vertex_coords, const float *vertex_sdf, float *edge_coords, int *configuration, float *vertices,
                                float *normals, int *n_vertices);
int chisel_hip_interpolate_vertex(const float v1[3], const float v2[3], float sdf1, float sdf2, float out[3]);
int chisel_hip_raycast(const float start[3], const float end[3], const int min_xyz[3], const int max_xyz[3], int *cells, int64_t capacity,
                       int64_t *count);
/* Which shapes the launch heuristics picked since the map was created / last reset of these figures (no reference counterpart: a
 * diagnostic beside chisel_hip_get_profile): out[0..7] = integration launches at 2 voxels per lane, at 4, at 4 with a 2-voxel tail;
 * cull launches with four waves per workgroup, with one wave per frame; launch sets without an order kernel; launch sets in the
 * short single-stream form; launch sets in all; out[8..9] = launch sets queued behind a mesh recompute whose totals the host had not seen
 * yet, and the ones of them that had to be replayed because that recompute did not fit.  A group handle sums its shards. */
#define CHISEL_HIP_NUM_LAUNCH_STATS 10
int chisel_hip_get_launch_stats(chisel_hip_map *map, int64_t *out, int reset_stats);
/* The chunk pool (no reference counterpart: ChunkManager's map has no capacity): out[0] = chunks with voxel memory behind them now,
 * out[1] = the most the pool can grow to (== out[0] for a fixed pool), out[2] = times it has grown, out[3] = 1 if it can grow.  A growing
 * pool commits more memory when its free slots fall under a quarter (checked when a launch set is queued, from the figures the integration
 * kernels report: nothing is waited for, nothing is moved, ids and slots stay what they are); a group handle reports its shards' sums. */
int chisel_hip_pool_info(chisel_hip_map *map, int64_t out[4]);
/* owner shard of a chunk id under (n_shards, shard_block); pure function, same on every rank */
int chisel_hip_chunk_owner(const int id_xyz[3], int n_shards, int shard_block);

#ifdef __cplusplus
}
#endif
#endif /* CHISEL_HIP_H_ */
