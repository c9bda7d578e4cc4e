// chisel_device.h -- structures shared by the host API (chisel_hip.hip) and the gfx950 kernels.
//
// HBM layout of one map (all arrays are hipMalloc'ed once at create time, sized by max_chunks = C):
//   sdf   float [C][V]    V = N^3 voxels of a chunk, index (z*N + y)*N + x (Chunk.h:81-84): x-rows are
//   wgt   float [C][V]    contiguous, so a wave64 reading 4 voxels/lane covers 256 consecutive voxels
//   rgbw  uchar4[C][V]    = 1 KiB per plane per instruction (fully coalesced).  DistVoxel / ColorVoxel
//                         payloads (8 B + 4 B) without the reference's vptr padding (16 B + 16 B).
//   hash_keys uint64[Hc]  open-addressing table keyed by the packed chunk id, probe = linear,
//   hash_vals int32 [Hc]  home bucket = the reference's ChunkHasher (ChunkManager.h:40-52) & (Hc-1)
//   slot_key  uint64[C]   packed id of the chunk living in a slot (EMPTY when free)
//   slot_dirty uint32[C]  "updated since the last mesh recompute" (Chisel.h:175-189 marks 27 neighbours
//                         on the host; here the mark is per slot and the neighbourhood is expanded later)
//   free_list int32 [C], free_top: stack of free slots
// Invariant: every FREE slot holds default voxels (sdf 99999, weight 0, rgbw 0 -- DistVoxel.cpp:27-31,
// ColorVoxel.cpp:27-31).  reset_map_kernel and remove_chunks_kernel restore it, so allocating a chunk
// (ChunkManager::CreateChunk) writes nothing but the hash entry and integration writes only the
// voxels that change.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace chisel_hip {

constexpr uint64_t KEY_EMPTY = ~0ull;
constexpr uint64_t KEY_TOMB = ~0ull - 1ull;
constexpr int ID_BIAS = 1 << 20;  // chunk ids in [-2^20, 2^20)
constexpr int INTEGRATE_MAX_GRID = 4096;  // rows of per-workgroup counters (power of two; larger grids wrap around)
constexpr int INTEGRATE_GRID_CAP = 1 << 19;  // largest integration grid (workgroups)

__host__ __device__ inline uint64_t pack_id(int x, int y, int z) {
    return (uint64_t)(uint32_t)(x + ID_BIAS) | ((uint64_t)(uint32_t)(y + ID_BIAS) << 21) |
           ((uint64_t)(uint32_t)(z + ID_BIAS) << 42);
}
__host__ __device__ inline void unpack_id(uint64_t k, int &x, int &y, int &z) {
    x = (int)(k & 0x1FFFFF) - ID_BIAS;
    y = (int)((k >> 21) & 0x1FFFFF) - ID_BIAS;
    z = (int)((k >> 42) & 0x1FFFFF) - ID_BIAS;
}
// ChunkHasher (ChunkManager.h:40-52): size_t arithmetic on sign-extended ints; third prime 8349279 (sic)
__host__ __device__ inline uint64_t chunk_hash(int x, int y, int z) {
    return ((uint64_t)(int64_t)x * 73856093ull) ^ ((uint64_t)(int64_t)y * 19349663ull) ^
           ((uint64_t)(int64_t)z * 8349279ull);
}
__host__ __device__ inline int floor_div(int a, int b) {
    int q = a / b, r = a % b;
    return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q;
}
// spatial ownership for the multi-GPU shard: pure function of the chunk id
__host__ __device__ inline int chunk_owner(int x, int y, int z, int n_shards, int shard_block) {
    if (n_shards <= 1) return 0;
    int bx = floor_div(x, shard_block), by = floor_div(y, shard_block), bz = floor_div(z, shard_block);
    int h = (bx + 3 * by + 5 * bz) % n_shards;
    return h < 0 ? h + n_shards : h;
}

// The map's error flags live in pinned host memory (the device addresses them through MapView::error_flag): two words, so
// that neither kind of report can overwrite the other (a third word, [2], is not an error: the number of work items of the
// latest integration launch, from which the host sizes a later launch's grid).
//   [0] chunk pool (1) or chunk hash (2) exhausted: the map is incomplete from here on; stays set until chisel_hip_reset
//   [1] a property of ONE point cloud (kernels_cloud.h: 3 = too many chunks / pairs, 4 = ray out of range): reported once, cleared
// Plain stores: every writer of a word stores a nonzero code; the host reads them after a wait, without a copy.
// Words [2], [3]: the work items / (item, frame) pairs of the latest integration launch (grid sizing hints).  Words [4], [5]: progress of the
// map's stream -- [4] the number of the latest integration launch that has STARTED (its first thread), [5] the number of the latest launch
// known to be OVER (stored by the count kernel of the recompute queued behind it) -- which the host reads instead of querying events.
__device__ inline void raise_error(int *flag, int code) { reinterpret_cast<volatile int *>(flag)[code >= 3 ? 1 : 0] = code; }

struct MapView {
    float *sdf;
    float *wgt;
    uchar4 *rgbw;            // null without colour
    uint64_t *hash_keys;
    int *hash_vals;
    uint64_t hash_mask;      // capacity - 1 (power of two)
    uint64_t *slot_key;
    uint32_t *slot_dirty;    // [max_chunks] "updated since the last mesh recompute"; behind them [max_chunks] the slots whose flag went 0 -> 1
                             // since the flags were last cleared, in order of appearance, and [2 * max_chunks] their number (mark_slot_dirty);
                             // from [2 * max_chunks + 16] on [max_chunks] sign summaries (slot_summary)
    int *free_list;
    int *free_top;
    unsigned long long *counters;  // CHISEL_HIP_NUM_COUNTERS (filled by reduce_counters_kernel)
    unsigned long long *block_counters;  // [INTEGRATE_MAX_GRID][16] per-workgroup partial sums
    int *error_flag;         // two words in pinned host memory, see raise_error
    int max_chunks;          // slots the per-slot arrays and the hash are laid out for (fixed at creation: the pool's upper limit)
    int committed;           // slots whose voxel payload has memory behind it, <= max_chunks: only these are ever on the free list (a growable
                             // pool commits more as it fills up, chisel_hip.hip: grow_pool; a fixed one has committed == max_chunks)
    // meshesToUpdate as a job list kept on the device while integrating (kernels_mesh.h: mesh_expand_dirty): the wave that first dirties
    // a slot appends the resident chunks of its 27-neighbourhood (Chisel.h:175-189), so that a recompute starts with its count kernel
    unsigned *mesh_flag;     // [max_chunks] "this slot is in the job list"
    int *mesh_jobs;          // [mesh_jobs_capacity][3] chunk ids
    int *mesh_ctl;           // [0..3] totals of the recompute in flight (triangles, grids, overflow, jobs), [4] entries of the job list
    int mesh_jobs_capacity;
};

// The bounding box of the ids of every chunk created since the last reset (never shrinks: removals leave it as it is), kept in
// mesh_ctl[MC_BBOX .. +5] = min x, y, z, max x, y, z by everything that creates a chunk.  A superset of what is resident, and an exact
// "absent" verdict for ids outside it: ChunkManager::InterpolateColor's eight look-ups (ChunkManager.cpp:506-520) take integer VOXEL
// indices for metric positions and land hundreds of chunks away, where one compare answers what was a hash probe per vertex.
constexpr int MC_BBOX = 136;
// mesh_ctl[MC_LATCH] != 0: the recompute in front did not fit its triangle list or its arena and will be emitted again from the voxels AS THEY
// ARE -- set by mesh_triangle_kernel, read by integrate_kernel (every wave leaves at once: the map stays as the recompute saw it), cleared by
// the host when it emits again and then replays the launches that left (host_mesh.h: check_mesh_totals)
constexpr int MC_LATCH = 5;
__device__ inline void bbox_include(int *mesh_ctl, int x, int y, int z) {
    if (!mesh_ctl) return;
    int *b = mesh_ctl + MC_BBOX;
    if (x < b[0]) atomicMin(&b[0], x);
    if (y < b[1]) atomicMin(&b[1], y);
    if (z < b[2]) atomicMin(&b[2], z);
    if (x > b[3]) atomicMax(&b[3], x);
    if (y > b[4]) atomicMax(&b[4], y);
    if (z > b[5]) atomicMax(&b[5], z);
}

// Per slot, behind the dirty list: which signs the OBSERVED voxels (weight > 0.5: what a marching cube asks of a corner,
// ChunkManager.cpp:271 / :352) a chunk has ever held can have -- SUM_POS: some voxel with sdf >= 0, SUM_NEG: some with sdf < 0.  Sticky (a
// carved voxel leaves its bit behind: the summary may say more than is there, never less); every writer of voxels ORs into it, freeing a
// slot clears it.  The mesher skips a chunk that holds no observed voxel (corner 0 of every cube is the chunk's own voxel) or whose
// cubes -- the chunk and its seven "+" neighbours -- cannot see both signs.
constexpr unsigned SUM_POS = 1u, SUM_NEG = 2u, SUM_ANY = 3u;
constexpr size_t SLOT_SUMMARY_PAD = 16;
__device__ inline uint32_t *slot_summary(const MapView &M) { return M.slot_dirty + 2 * (size_t)M.max_chunks + SLOT_SUMMARY_PAD; }

// A chunk was updated (Chisel.h:85 / :167 needsUpdate -> meshesToUpdate, Chisel.h:175-189): its flag, and -- for the mesher, which must
// not have to scan a pool of millions of slots for a few hundred dirty ones -- its slot into the list of dirty slots, once.
// -> true for the caller whose exchange raised the flag
__device__ inline bool mark_slot_dirty(const MapView &M, int slot) {
    if (atomicExch(&M.slot_dirty[slot], 1u) == 0u) {
        const unsigned p = atomicAdd(&M.slot_dirty[2 * (size_t)M.max_chunks], 1u);
        if (p < (unsigned)M.max_chunks) M.slot_dirty[(size_t)M.max_chunks + p] = (unsigned)slot;  // (else: the mesher scans the flags)
        return true;
    }
    return false;
}

// depth min/max pyramid: level l (PYR_L0 <= l <= PYR_L1) has ceil(W/2^l) x ceil(H/2^l) texels of
// (min, max) over the valid pixels of a 2^l x 2^l block; (+inf, -inf) when the block has none.
constexpr int PYR_L0 = 2, PYR_L1 = 6, PYR_LEVELS = PYR_L1 - PYR_L0 + 1;
struct PyramidView {
    float2 *data;
    int off[PYR_LEVELS];
    int w[PYR_LEVELS];
    int h[PYR_LEVELS];
};

struct CameraParams {
    float R[9];  // camera->world rotation, row-major (Transform::linear())
    float t[3];  // Transform::translation()
    float fx, fy, cx, cy;
    int W, H;
};

// One launch set (pyramid -> cull -> integrate) handles up to KMAX frames, applied to every voxel in
// frame order.  All per-frame constants travel in the kernel-argument segment (scalar loads).
constexpr int KMAX = 16;

// ProjectionIntegrator state + map constants, identical for every frame of a batch
struct IntegratorParams {
    int trunc_kind;
    float trunc_param;
    float weight;
    int carving;
    float carving_dist;
    float res;               // Chunk::GetVoxelResolutionMeters
    float half_res;          // ChunkManager.cpp:52  (res * 0.5f)
    float diag;              // ProjectionIntegrator.h:58  2.0 * sqrt(3.0f) * res  (double, narrowed)
    float max_depth;         // 50 (Integrate :74) or 100 (IntegrateColor :141)
    int n_shards, shard_rank, shard_block;
    int single_chunk;        // ProjectionIntegrator::Integrate on ONE chunk (chisel_hip_integrate_chunk): the candidate range is that id, whatever the
                             // frustum says, and the reference's plane test is not part of that call
};

// pixel record of one depth pixel: x = depth reading (NaN when the reference skips the pixel: NaN depth, depth > max_depth),
// y = truncator->GetTruncationDistance(depth).  Built once per frame by depth_pyramid_kernel so the per-voxel loop does no
// truncator arithmetic.  Eight bytes, not sixteen with the band / carve thresholds and the weight precomputed: the gathers of
// these records are what the integration kernel waits for (16-byte records: 115 -> 171 us per launch), not its arithmetic.
// Record -1 of every frame (padding in front of the image) is all NaN.
typedef float2 PixelRec;

// per-frame arguments of the integration kernel
struct FrameCam {
    CameraParams cam;
    CameraParams ccam;       // colour camera (IntegrateColor only)
    const PixelRec *rec;     // W x H records of this frame
    const uint8_t *color;
    int color_channels;
    int same_cam;            // colour camera == depth camera bit for bit: reuse the depth projection
};
struct IntegrateParams {
    IntegratorParams ip;
    int n_frames;
    FrameCam f[KMAX];
};

// per-frame arguments of the cull kernel: the reference's candidate enumeration of that frame
// (ChunkManager.cpp:182-212): ids range_min .. range_min+range_dim-1, and the six frustum planes
struct CullFrame {
    CameraParams cam;
    int range_min[3];
    int range_dim[3];
    float planes[24];        // far, near, top, bottom, left, right: normal xyz + distance (Frustum.cpp:43)
};
struct CullParams {
    IntegratorParams ip;
    int n_frames;
    int range_min[3];        // union of the frames' ranges: one thread per chunk id of it
    int range_dim[3];
    int pyr_stride;          // texels between two frames' pyramids
    CullFrame f[KMAX];
};

// per-frame arguments of the pyramid / pixel-record kernel
struct PyramidParams {
    IntegratorParams ip;
    int W, H;
    int rec_stride;          // records between two frames
    int pyr_stride;
    PixelRec *rec;
    const float *depth[KMAX];
};

// A candidate (written by cull_kernel: geometry only, the map is not consulted) and a work item (written by
// its first wave's look-up: the candidates that can change the map, with their pool slot) share this layout.
struct WorkItem {
    int x, y, z;             // chunk id
    int slot;                // pool slot, -1 = not resident (allocated if any voxel is integrated); candidates: -1;
                             // SLOT_LOOKUP = the previous batch may be creating this chunk: the integration kernel looks it up
    unsigned frame_mask;     // work item: frames of the batch that may touch this chunk
                             // candidate: bits 0-15 frames that may integrate, bits 16-31 frames that may carve
    int box;                 // row of the FrameBox array that holds this chunk's boxes (its candidate index)
    unsigned inband_mask;    // work item: frames that may integrate (used when slot == SLOT_LOOKUP)
    int pad;
};
constexpr int SLOT_LOOKUP = -2;
// chunks the previous batch may create (open-addressing set of packed ids, one per batch buffer set)
constexpr unsigned PENDING_CAPACITY = 1u << 14;
struct FrameBox {            // one per (candidate, frame): the cull kernel's verdict (WI_* flags; 0 = the frame cannot touch the chunk).
    int flags;               // (Until round 3 also the chunk's pixel box and camera-z bounds, which the integration kernel staged and
};                           // pre-tested with; since round 4 its units read need masks -- per cell, since round 6 per brick: brick_kernel -- and the box went.)
constexpr int WI_INBAND = 1;   // some voxel may take the in-band branch
constexpr int WI_CARVE = 2;    // some voxel may take the carve test (only matters while the chunk is resident)
constexpr int WI_TILE = 4;     // u0..v1 is a valid bounding box (else: gather from the whole image)
constexpr int WI_FASTZ = 8;    // camera z of every voxel of the chunk lies in [FASTZ_MIN, FASTZ_MAX]: reciprocal_in_range() applies
constexpr int WI_FASTWU = 16;  // ConstantWeighter(1) and 5 x truncation distance of every valid pixel under the chunk lies in that range:
                               // weight / (5 * truncation) (ConstantWeighter.h:43-46) is reciprocal_in_range(5 * truncation), exactly
constexpr int WI_INSIDE = 32;  // WI_FASTZ, and every voxel of the chunk projects onto the image in this frame (two pixels of margin on every
                               // side, rounding of the per-voxel projection bounded: cull_chunk_frame): IsPointOnImage holds for all of them

// Chunk-level state of one work item while its waves run (device-scope atomics only; zeroed by whoever writes the item).
struct ItemSync {
    unsigned band;         // items without a slot: frames in which some voxel was integrated
    unsigned changed;      // frames in which some voxel changed (each bit is counted by the wave that sets it first)
    int slot;              // items without a slot: 0 = nobody has allocated yet, 1 = being allocated, s + 2 = pool slot s, -1 = failed
    unsigned arrived;      // items without a slot: waves that have deposited their band / carve figures
    unsigned carve[KMAX];  // items without a slot: voxels that took the carve test, per frame
};
static_assert(sizeof(ItemSync) == 80, "ItemSync is zeroed as five 16-byte stores");
__device__ inline void item_sync_init(ItemSync *s) {
    uint4 *p = reinterpret_cast<uint4 *>(s);
#pragma unroll
    for (int i = 0; i < 5; i++) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

// ---- strategy arithmetic (devirtualised Truncator / Weighter) -------------------------------------
// InverseTruncator.h:48-52
constexpr float kBaseLine = 0.10;
constexpr float kFocal = 471.27;
constexpr float kDepSample = 1.0f / (kBaseLine * kFocal);
// QuadraticTruncator.h:65-67
constexpr float kQuadTerm = 0.0019 * 10;
constexpr float kLinTerm = 0.00152 * 10;
constexpr float kConstTerm = 0.001504 * 10;

__host__ __device__ inline float truncation_distance(int kind, float param, float reading) {
    if (kind == 1) {  // InverseTruncator.h:42-46: float inv = 1.0 / reading (double divide narrowed == fp32 divide)
        float inv_reading = 1.0f / reading;
        return (kDepSample / (inv_reading * inv_reading)) * param;
    } else if (kind == 0) {  // ConstantTruncator.h:48-51
        return param;
    } else {  // QuadraticTruncator.h:42-45: double arithmetic via pow(reading, 2)
        double r = (double)reading;
        double v = (double)kQuadTerm * (r * r) + (double)(kLinTerm * reading) + (double)kConstTerm;
        v = v < 0 ? -v : v;
        return (float)(v * (double)param);
    }
}
// 1.0f / z, correctly rounded, for z in [2^-40, 2^40]: the hardware reciprocal (1 ulp) refined by one Newton step whose
// residual is exact thanks to the fused multiply-add.  Checked against the IEEE division for EVERY float of that range on
// the device (chisel_hip_kat_reciprocal, tests/test_gpu_parity.py; the bare v_rcp_f32 fails that test, one step passes).
// Three instructions instead of the eleven of the general division sequence, which also has to scale denormal and huge
// operands; the callers establish the range (cull_kernel: WI_FASTZ).
__device__ inline float reciprocal_in_range(float z) {
    const float r = __builtin_amdgcn_rcpf(z);
    return __builtin_fmaf(__builtin_fmaf(-z, r, 1.0f), r, r);
}
// (int)floorf(x) in one instruction (v_cvt_flr_i32_f32; the compiler emits v_floor_f32 + v_cvt_i32_f32).  Checked against that pair
// for every float that is not a NaN on the device, and that no NaN comes out as a possible pixel coordinate
// (chisel_hip_kat_floor, tests/test_gpu_parity.py).
__device__ inline int floor_to_int(float x) {
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
constexpr float FASTZ_MIN = 9.094947017729282e-13f;  // 2^-40
constexpr float FASTZ_MAX = 1099511627776.0f;        // 2^40
// ConstantWeighter.h:43-46
__host__ __device__ inline float constant_weight(float weight, float truncation) { return weight / (5 * truncation); }
// DistVoxel::Integrate DistVoxel.h:52-60
__host__ __device__ inline void dist_integrate(float &sdf, float &w, float distUpdate, float weightUpdate) {
    float newDist = (w * sdf + weightUpdate * distUpdate) / (weightUpdate + w);
    sdf = newDist;
    w = w + weightUpdate;
}
// ColorVoxel::Integrate ColorVoxel.h:65-85 (one channel):
//   sat_0^255( ((float)weight * old + (float)(weightUpdate * new)) / (float)(weightUpdate + weight) ) truncated to uint8.
// Numerator and denominator are small integers, exact in fp32, and the correctly rounded quotient of integers x / d can
// only reach an integer when x / d is one (the nearest miss is 1/d away, d <= 510), so the truncated result is the
// integer quotient: no fp32 division needed (checked exhaustively for every x, d in tests/test_oracle_kat.py).
__host__ __device__ inline uint8_t color_channel(uint8_t old, uint8_t weight, uint8_t nw, uint8_t weightUpdate) {
    const unsigned x = (unsigned)weight * (unsigned)old + (unsigned)weightUpdate * (unsigned)nw;
    const unsigned d = (unsigned)weightUpdate + (unsigned)weight;
    if (d == 0u) return 0;  // 0 / 0 = NaN: fminf(fmaxf(NaN, 0), 255) = 0
    const unsigned q = x / d;
    return (uint8_t)(q > 255u ? 255u : q);
}
__host__ __device__ inline uchar4 color_integrate(uchar4 c, uint8_t r, uint8_t g, uint8_t b, uint8_t wu) {
    if ((int)c.w >= 255 - (int)wu) return c;
    uchar4 o;
    o.x = color_channel(c.x, c.w, r, wu);
    o.y = color_channel(c.y, c.w, g, wu);
    o.z = color_channel(c.z, c.w, b, wu);
    o.w = (uint8_t)(c.w + wu);
    return o;
}
// color_integrate for weightUpdate == 1 and weight < 8 (the only call of the integrator: ProjectionIntegrator.h:152-157), with
// the new colour given as a packed word, red in bits 0-7, green 8-15, blue 16-23.  The integer quotient x / d, d = weight + 1
// in 1..8, x = weight * old + new <= 255 d, is taken without a division: (x + 0.5) / d lies at least 1/16 away from every
// integer, the reciprocal is good to 1 ulp and x + 0.5 < 2^12, so truncating (x + 0.5) * rcp(d) gives floor(x / d)
// (checked against color_integrate for every (weight, old, new) in tests/test_oracle_kat.py).
__device__ inline unsigned color_integrate_fresh(unsigned c, unsigned rgb) {
    const unsigned w = c >> 24;
    const float wf = (float)w;
    const float r = __builtin_amdgcn_rcpf(wf + 1.0f);
    const float x0 = __builtin_fmaf(wf, (float)(c & 0xffu), (float)(rgb & 0xffu) + 0.5f);
    const float x1 = __builtin_fmaf(wf, (float)((c >> 8) & 0xffu), (float)((rgb >> 8) & 0xffu) + 0.5f);
    const float x2 = __builtin_fmaf(wf, (float)((c >> 16) & 0xffu), (float)((rgb >> 16) & 0xffu) + 0.5f);
    const unsigned q0 = (unsigned)(x0 * r), q1 = (unsigned)(x1 * r), q2 = (unsigned)(x2 * r);
    return q0 | (q1 << 8) | (q2 << 16) | ((w + 1u) << 24);
}
// ColorImage::At ColorImage.h:72-107 -> (red, green, blue)
__device__ inline void color_at(const uint8_t *data, int idx, int channels, uint8_t &r, uint8_t &g, uint8_t &b) {
    const uint8_t *p = data + (size_t)idx * channels;
    if (channels == 1) {
        r = g = b = p[0];
    } else if (channels == 2) {
        r = p[0];
        g = b = p[1];
    } else {  // 3 = BGR, 4 = BGRA
        r = p[2];
        g = p[1];
        b = p[0];
    }
}
// ColorImage::At of 3 / 4 channel images as one 4-byte gather: color_gather() requests the pixel's bytes (byte address
// channels * idx is not aligned; the last pixel of a 3-channel image is read from 1 byte lower so that no byte past the image
// is touched), color_word() turns them into red | green << 8 | blue << 16 once they are needed.  image_bytes = W * H * channels.
typedef unsigned __attribute__((aligned(1))) unaligned_u32;
__device__ inline unsigned color_gather(const uint8_t *data, int idx, int channels, unsigned image_bytes, unsigned &shift) {
    const unsigned b = (unsigned)idx * (unsigned)channels;
    const unsigned lo = (b + 4u > image_bytes) ? image_bytes - 4u : b;
    shift = (b - lo) * 8u;
    return *reinterpret_cast<const unaligned_u32 *>(data + lo);
}
__device__ inline unsigned color_word(unsigned raw, unsigned shift) {
    const unsigned v = raw >> shift;  // B | G << 8 | R << 16 (| A << 24)
    return ((v >> 16) & 0xffu) | (v & 0xff00u) | ((v & 0xffu) << 16);
}

// The same two steps with fewer instructions (round 3).  color_gather2: offset by a 24-bit multiply, the last pixel's clamp by a
// minimum (last_word = image_bytes - 4).  color_integrate_fresh_bgr: color_integrate for weightUpdate == 1 and weight < 8 on the
// pixel's bytes as they lie in memory (v = raw >> shift: blue | green << 8 | red << 16; ColorImage::At's BGR decode is the
// choice of byte per channel).  Per channel x = weight * old + new <= 2040 (exact in fp32), and the integer quotient x / d,
// d = weight + 1 in 1..8, comes from v_cvt_pk_u8_f32, which rounds to nearest (even) and saturates: x / d - 1/2 + 1 / (2 d)
// lies within 1/2 - 1/16 of floor(x / d), and the computed value -- one fma on a reciprocal good to 1 ulp, magnitudes below
// 2^12 -- within 2^-10 of that.  Checked against color_integrate for every (weight, old, new) on the device
// (chisel_hip_kat_color_fresh).
__device__ inline unsigned color_gather2(const uint8_t *data, unsigned idx, unsigned channels, unsigned last_word, unsigned &shift) {
    const unsigned b = __umul24(idx, channels);
    const unsigned lo = b < last_word ? b : last_word;
    shift = (b - lo) * 8u;
    return *reinterpret_cast<const unaligned_u32 *>(data + lo);
}
__device__ inline unsigned color_integrate_fresh_bgr(unsigned c, unsigned v) {
    const float wf = (float)(c >> 24);
    const float r = __builtin_amdgcn_rcpf(wf + 1.0f);
    const float hr = __builtin_fmaf(0.5f, r, -0.5f);
    const float x0 = __builtin_fmaf(wf, (float)(c & 0xffu), (float)((v >> 16) & 0xffu));          // red
    const float x1 = __builtin_fmaf(wf, (float)((c >> 8) & 0xffu), (float)((v >> 8) & 0xffu));    // green
    const float x2 = __builtin_fmaf(wf, (float)((c >> 16) & 0xffu), (float)(v & 0xffu));          // blue
    unsigned q = c + 0x01000000u;  // weight + 1 (< 9); the three colour bytes are replaced below
    q = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(x0, r, hr), 0u, q);
    q = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(x1, r, hr), 1u, q);
    q = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(x2, r, hr), 2u, q);
    return q;
}

// thresholds the reference compares in double against fp32 values, folded to fp32 (exactly equivalent):
//   sdf < 1e-5 (double)  <=>  sdf < kSdfCarveThr,  kSdfCarveThr = smallest float >= 1e-5
//   (ProjectionIntegrator.h:90,168)
// (float)1e-5 = 9.99999974737875e-06 is the largest float below the double 1e-5, so the double comparison is "<=" on it
__host__ __device__ inline bool sdf_below_carve_threshold(float sdf) { return sdf <= 1e-5f; }

}  // namespace chisel_hip
