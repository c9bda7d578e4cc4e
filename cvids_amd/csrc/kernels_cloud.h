// kernels_cloud.h -- point-cloud fusion mode (Chisel::IntegratePointCloud) on gfx950.
//
// Reference: Chisel.cpp:107-157 (driver), ChunkManager.cpp:214-257 (chunks a cloud touches), ProjectionIntegrator.cpp:52-173
// (per-chunk update), geometry/Raycast.cpp:4-128 (Amanatides-Woo walk with an integer direction).
//
// What the reference does: (1) for every point, walk the segment point -+ `truncation` along its viewing ray through the
// CHUNK grid and collect the chunk ids ("listed" chunks); (2) for every listed chunk, for EVERY point of the cloud in cloud
// order, walk the segment point -+ truncator(depth) through the VOXEL grid relative to that chunk, clipped to the chunk's box,
// and update the voxels met, in walk order; (3) chunks that were new and stayed untouched are erased.  A voxel is met by many
// rays and its running average depends on their order, so per voxel the updates must be applied in cloud order.
//
// Here:
//   cloud_tile_count_kernel / cloud_tile_scan_kernel   index of each point among the points that pass the depth limit (the
//                                                      reference's colour index only advances on those, :68-70 / :130-132)
//   cloud_prepare_kernel    per point: world point, ray direction, the two segment ends (CloudRay), colour bytes; walks the
//                           chunk grid and enters the chunks met into a per-cloud table (step 1)
//   cloud_bin_kernel<FILL>  per point: the listed chunks whose box the voxel walk can enter (exact per-axis cell ranges); first
//                           pass counts per chunk, second pass (after cloud_offsets_kernel) writes the (chunk, point) pairs
//   cloud_sort_kernel       per chunk: its points into cloud order (LDS bitmap over the point indices)
//   cloud_integrate_kernel  per chunk: voxel box in LDS; 128 rays walked at a time (one per lane), their cells kept in LDS; then
//                           the rays are applied one after the other in cloud order, the cells of a ray in parallel, the four
//                           waves owning disjoint voxel blocks so that they need no barrier between rays (step 2); the chunk
//                           is created when the first update happens (same outcome as create-then-erase, step 3)
#pragma once
#include "chisel_device.h"

namespace chisel_hip {

constexpr int CLOUD_TILE = 256;                  // points per workgroup in the per-point kernels
constexpr unsigned CLOUD_TABLE_SLOTS = 1u << 17; // open-addressing table of the listed chunks of one cloud
constexpr int CLOUD_MAX_LISTED = 1 << 16;
constexpr int CLOUD_PAIRS_PER_POINT = 16;        // capacity of the (chunk, point) list, per point of the cloud
constexpr int CLOUD_MAX_RANGE = 4096;            // chunk boxes around one ray that are looked at
constexpr int CLOUD_SORT_WORDS = 15360;          // LDS bitmap of cloud_sort_kernel (60 KB): 491 520 point indices per pass
constexpr int CLOUD_RAYS = 128;                  // rays walked at a time by one workgroup of cloud_integrate_kernel
constexpr int CLOUD_GRID = 2048;                 // persistent grids of the per-chunk kernels (<= INTEGRATE_MAX_GRID)
// error_flag values of this path (1, 2: chunk pool / hash, kernels_integrate.h)
constexpr int CLOUD_ERR_CAPACITY = 3;            // too many listed chunks or (chunk, point) pairs
constexpr int CLOUD_ERR_RANGE = 4;               // a ray leaves the supported chunk-id range or is too long

struct CloudRay {            // one point of the cloud, ready for the voxel walk
    float ax, ay, az;        // worldPoint - dir * truncation    (NaN in ax: the point is skipped or dead)
    float bx, by, bz;        // worldPoint + dir * truncation
    float depth;             // point.z (sensor frame)
    float trunc;             // truncator->GetTruncationDistance(depth)
};

struct CloudParams {
    IntegratorParams ip;
    float pose[12];          // Transform, row-major 3x4
    float inv[12];           // Transform::inverse(), computed on the host (host_cloud.h)
    float truncation;        // chunk enumeration only (ChiselServer.cpp:523 passes 0.1)
    float max_dist;
    float depth_limit;       // 2 (ProjectionIntegrator.cpp:69) or 5 with colours (:131)
    int with_color;          // cloud.HasColor() && chunk->HasColors() (:42)
    int n_points;
    int N;
};

struct CloudView {
    const float *points;     // n x 3
    const float *colors;     // n x 3 or null
    CloudRay *rays;          // n
    unsigned *rgb;           // n: red | green << 8 | blue << 16 of the colour the reference pairs with the point
    int *tile_prefix;        // per CLOUD_TILE points: accepted points before the tile
    uint64_t *table_keys;    // CLOUD_TABLE_SLOTS
    int *table_vals;
    uint64_t *listed;        // CLOUD_MAX_LISTED packed ids, in order of discovery
    int *offsets;            // CLOUD_MAX_LISTED + 1: pairs per listed chunk, then their exclusive prefix
    int *cursors;            // CLOUD_MAX_LISTED
    int *pairs;              // pairs_capacity point indices, grouped by chunk, unordered
    int *sorted;             // the same in cloud order
    int pairs_capacity;
    int *ctl;                // [0] listed chunks, [1] pairs
};

// ---- fp32 sequences of the Eigen expressions involved (Eigen 3.3; see oracle/chisel_oracle.cpp) ------------------------------
// Transform * Vec3: ((m0 x + m1 y) + m2 z) + m3 per row
__device__ inline float affine_row(const float *m, float x, float y, float z) { return ((m[0] * x + m[1] * y) + m[2] * z) + m[3]; }
// floor() to int for a coordinate known to be finite and inside the int range
__device__ inline int floor_int(float v) { return (int)floorf(v); }
__device__ inline bool cell_coordinate_ok(float v) {
    const float f = floorf(v);
    return f >= -2147483648.0f && f < 2147483648.0f;
}
// Raycast.cpp:9-12  mod(value, 1.0f): the unqualified fmod binds to the double overload, the sum is taken in double
__device__ inline float ray_mod1(float v) {
    const double f = (double)(v - truncf(v));  // fmod(v, 1): exact
    const double y = f + 1.0;                   // in (0, 2)
    return (float)(y >= 1.0 ? y - 1.0 : y);     // fmod(y, 1)
}
// Raycast.cpp:14-33
__device__ inline float ray_intbound(float s, int ds) {
    if (ds == 0) return __builtin_inff();  // (float)DBL_MAX
    if (ds < 0) {
        s = -s;
        ds = -ds;
    }
    s = ray_mod1(s);
    return (1.0f - s) / (float)ds;
}

// Raycast.cpp:35-128 as a state machine: cell() is the current cell; next() moves on and returns false after the last cell.
struct RayWalk {
    int x, y, z, ex, ey, ez, sx, sy, sz;
    float tmx, tmy, tmz, tdx, tdy, tdz;
    // false: no cell at all (a coordinate is not finite / not an int, or start and end share a cell: Raycast.cpp:79-80)
    __device__ bool begin(float ax, float ay, float az, float bx, float by, float bz) {
        if (!(cell_coordinate_ok(ax) && cell_coordinate_ok(ay) && cell_coordinate_ok(az) && cell_coordinate_ok(bx) &&
              cell_coordinate_ok(by) && cell_coordinate_ok(bz)))
            return false;
        x = floor_int(ax); y = floor_int(ay); z = floor_int(az);
        ex = floor_int(bx); ey = floor_int(by); ez = floor_int(bz);
        const int dx = (int)((unsigned)ex - (unsigned)x), dy = (int)((unsigned)ey - (unsigned)y), dz = (int)((unsigned)ez - (unsigned)z);
        sx = (dx > 0) - (dx < 0); sy = (dy > 0) - (dy < 0); sz = (dz > 0) - (dz < 0);
        if (sx == 0 && sy == 0 && sz == 0) return false;
        tmx = ray_intbound(ax, dx); tmy = ray_intbound(ay, dy); tmz = ray_intbound(az, dz);
        tdx = (float)sx / (float)dx; tdy = (float)sy / (float)dy; tdz = (float)sz / (float)dz;  // 0 / 0 = NaN on an idle axis, never added
        return true;
    }
    __device__ unsigned long long length() const {  // steps of a terminating walk
        return (unsigned long long)abs((long long)ex - (long long)x) + (unsigned long long)abs((long long)ey - (long long)y) +
               (unsigned long long)abs((long long)ez - (long long)z);
    }
    __device__ bool next() {
        if (x == ex && y == ey && z == ez) return false;
        if (tmx < tmy) {
            if (tmx < tmz) {
                if (x == ex) return false;  // would step past the end cell (the reference never returns from there)
                x += sx; tmx += tdx;
            } else {
                if (z == ez) return false;
                z += sz; tmz += tdz;
            }
        } else {
            if (tmy < tmz) {
                if (y == ey) return false;
                y += sy; tmy += tdy;
            } else {
                if (z == ez) return false;
                z += sz; tmz += tdz;
            }
        }
        return true;
    }
};

// ---- per-cloud chunk table ----------------------------------------------------------------------------------------------------
__device__ inline unsigned cloud_table_home(uint64_t key) {
    int x, y, z;
    unpack_id(key, x, y, z);
    return (unsigned)chunk_hash(x, y, z) & (CLOUD_TABLE_SLOTS - 1u);
}
__device__ inline void cloud_table_insert(const CloudView &C, const MapView &M, uint64_t key) {
    unsigned i = cloud_table_home(key);
    for (unsigned probe = 0; probe < CLOUD_TABLE_SLOTS; probe++, i = (i + 1u) & (CLOUD_TABLE_SLOTS - 1u)) {
        const uint64_t k = C.table_keys[i];
        if (k == key) return;
        if (k == KEY_EMPTY) {
            const uint64_t old = atomicCAS((unsigned long long *)&C.table_keys[i], (unsigned long long)KEY_EMPTY, (unsigned long long)key);
            if (old == KEY_EMPTY) {
                const int idx = atomicAdd(&C.ctl[0], 1);
                if (idx < CLOUD_MAX_LISTED) {
                    C.listed[idx] = key;
                    C.table_vals[i] = idx;  // read by later kernels only
                } else {
                    C.table_vals[i] = -1;
                    atomicExch(M.error_flag, CLOUD_ERR_CAPACITY);
                }
                return;
            }
            if (old == key) return;
        }
    }
    atomicExch(M.error_flag, CLOUD_ERR_CAPACITY);
}
__device__ inline int cloud_table_find(const CloudView &C, uint64_t key) {
    unsigned i = cloud_table_home(key);
    for (unsigned probe = 0; probe < CLOUD_TABLE_SLOTS; probe++, i = (i + 1u) & (CLOUD_TABLE_SLOTS - 1u)) {
        const uint64_t k = C.table_keys[i];
        if (k == key) return C.table_vals[i];
        if (k == KEY_EMPTY) return -1;
    }
    return -1;
}

// ---- colour index: rank of a point among the points that pass the depth limit ------------------------------------------------
__global__ __launch_bounds__(CLOUD_TILE) void cloud_tile_count_kernel(CloudParams P, CloudView C) {
    const int p = blockIdx.x * CLOUD_TILE + threadIdx.x;
    bool accept = false;
    if (p < P.n_points) accept = !(C.points[3 * (size_t)p + 2] > P.depth_limit);
    __shared__ int s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const unsigned long long b = __ballot(accept);
    if ((threadIdx.x & 63) == 0) atomicAdd(&s_n, __popcll(b));
    __syncthreads();
    if (threadIdx.x == 0) C.tile_prefix[blockIdx.x] = s_n;
}
// exclusive prefix of `n` ints in place, total into *total (one workgroup of 1024 threads)
__global__ __launch_bounds__(1024) void cloud_scan_kernel(int *data, const int *n_ptr, int n_fixed, int *total, int capacity, int *error_flag) {
    __shared__ int s_part[1024];
    __shared__ int s_carry;
    const int n = n_ptr ? min(*n_ptr, n_fixed) : n_fixed;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < n ? data[i] : 0;
        s_part[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int t = threadIdx.x >= o ? s_part[threadIdx.x - o] : 0;
            __syncthreads();
            s_part[threadIdx.x] += t;
            __syncthreads();
        }
        const int incl = s_part[threadIdx.x], carry = s_carry;
        if (i < n) data[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        data[n] = s_carry;
        if (total) *total = s_carry;
        if (capacity > 0 && s_carry > capacity) atomicExch(error_flag, CLOUD_ERR_CAPACITY);
    }
}

// ---- per point: ray ends, colour, listed chunks (ChunkManager.cpp:214-257, ProjectionIntegrator.cpp:63-82) -------------------
__global__ __launch_bounds__(CLOUD_TILE) void cloud_prepare_kernel(CloudParams P, CloudView C, MapView M) {
    const int p = blockIdx.x * CLOUD_TILE + threadIdx.x;
    const bool live = p < P.n_points;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (live) {
        px = C.points[3 * (size_t)p];
        py = C.points[3 * (size_t)p + 1];
        pz = C.points[3 * (size_t)p + 2];
    }
    const float depth = pz;
    const bool accept = live && !(depth > P.depth_limit);
    // index among the accepted points ("i" of the reference's loop)
    __shared__ int s_wave[CLOUD_TILE / 64];
    const unsigned long long b = __ballot(accept);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_wave[wave] = __popcll(b);
    __syncthreads();
    int before = C.tile_prefix[blockIdx.x];
    for (int w = 0; w < wave; w++) before += s_wave[w];
    const int cidx = before + __popcll(b & ((1ull << lane) - 1ull));
    if (!live) return;

    const float wx = affine_row(P.pose + 0, px, py, pz), wy = affine_row(P.pose + 4, px, py, pz), wz = affine_row(P.pose + 8, px, py, pz);
    const float ddx = wx - P.pose[3], ddy = wy - P.pose[7], ddz = wz - P.pose[11];
    const float z2 = ddx * ddx + (ddy * ddy + ddz * ddz);  // squaredNorm(): a0 + (a1 + a2)
    float dirx = ddx, diry = ddy, dirz = ddz;               // normalized(): n / sqrt(z) when z > 0
    const float len = sqrtf(z2);
    if (z2 > 0.0f) {
        dirx = ddx / len;
        diry = ddy / len;
        dirz = ddz / len;
    }
    CloudRay r;
    r.depth = depth;
    r.trunc = truncation_distance(P.ip.trunc_kind, P.ip.trunc_param, depth);
    r.ax = wx - dirx * r.trunc; r.ay = wy - diry * r.trunc; r.az = wz - dirz * r.trunc;
    r.bx = wx + dirx * r.trunc; r.by = wy + diry * r.trunc; r.bz = wz + dirz * r.trunc;
    if (!accept) r.ax = __builtin_nanf("");
    C.rays[p] = r;
    if (P.with_color && accept) {
        const float cr = C.colors[3 * (size_t)cidx], cg = C.colors[3 * (size_t)cidx + 1], cb = C.colors[3 * (size_t)cidx + 2];
        // (uint8_t)(c * 255.0f): cvttss2si, low byte
        C.rgb[p] = ((unsigned)(int)(cr * 255.0f) & 0xffu) | (((unsigned)(int)(cg * 255.0f) & 0xffu) << 8) |
                   (((unsigned)(int)(cb * 255.0f) & 0xffu) << 16);
    }

    // chunks the segment world -+ dir * truncation passes through
    if (len > P.max_dist) return;
    const float cs = (float)P.N * P.ip.res;  // chunkSize.x() * voxelResolutionMeters
    const float round = 1.0f / cs;
    RayWalk w;
    if (!w.begin((wx - dirx * P.truncation) * round, (wy - diry * P.truncation) * round, (wz - dirz * P.truncation) * round,
                 (wx + dirx * P.truncation) * round, (wy + diry * P.truncation) * round, (wz + dirz * P.truncation) * round))
        return;
    const int lim = ID_BIAS - 2;
    if (w.length() > 4096ull || abs(w.x) > lim || abs(w.y) > lim || abs(w.z) > lim || abs(w.ex) > lim || abs(w.ey) > lim || abs(w.ez) > lim) {
        atomicExch(M.error_flag, CLOUD_ERR_RANGE);
        return;
    }
    do {
        if (chunk_owner(w.x, w.y, w.z, P.ip.n_shards, P.ip.shard_block) == P.ip.shard_rank) cloud_table_insert(C, M, pack_id(w.x, w.y, w.z));
    } while (w.next());
}

// ---- (chunk, point) pairs ------------------------------------------------------------------------------------------------------
// cells of one axis the voxel walk of ray (a, b) can take inside chunk coordinate c: between floor((a - o) * round) and
// floor((b - o) * round), the arithmetic of ProjectionIntegrator.cpp:73-80 with o = Chunk::GetOrigin() (Chunk.cpp:43)
__device__ inline bool axis_enters(float a, float b, int c, int N, float res, float round) {
    const float o = (float)(N * c) * res;
    const float s = (a - o) * round, e = (b - o) * round;
    if (!(cell_coordinate_ok(s) && cell_coordinate_ok(e))) return false;
    const int si = floor_int(s), ei = floor_int(e);
    return max(si, ei) >= 0 && min(si, ei) < N;
}
template <bool FILL>
__global__ __launch_bounds__(CLOUD_TILE) void cloud_bin_kernel(CloudParams P, CloudView C, MapView M) {
    const int p = blockIdx.x * CLOUD_TILE + threadIdx.x;
    if (p >= P.n_points) return;
    const CloudRay r = C.rays[p];
    // skipped points (NaN in ax) and rays with a coordinate that is not finite meet no cell (RayWalk::begin)
    if (!(isfinite(r.ax) && isfinite(r.ay) && isfinite(r.az) && isfinite(r.bx) && isfinite(r.by) && isfinite(r.bz))) return;
    const float lo[3] = {fminf(r.ax, r.bx), fminf(r.ay, r.by), fminf(r.az, r.bz)};
    const float hi[3] = {fmaxf(r.ax, r.bx), fmaxf(r.ay, r.by), fmaxf(r.az, r.bz)};
    const float cs = (float)P.N * P.ip.res;
    int c0[3], c1[3];
    for (int a = 0; a < 3; a++) {
        const float f0 = floorf(lo[a] / cs), f1 = floorf(hi[a] / cs);
        const float lim = (float)(ID_BIAS - 4);
        if (!(f0 >= -lim && f1 <= lim)) {
            if (FILL) atomicExch(M.error_flag, CLOUD_ERR_RANGE);
            return;
        }
        c0[a] = (int)f0 - 1;  // one chunk of slack: the decisive test below uses the reference's own arithmetic
        c1[a] = (int)f1 + 1;
    }
    if ((long long)(c1[0] - c0[0] + 1) * (c1[1] - c0[1] + 1) * (c1[2] - c0[2] + 1) > CLOUD_MAX_RANGE) {
        if (FILL) atomicExch(M.error_flag, CLOUD_ERR_RANGE);
        return;
    }
    const float round = 1.0f / P.ip.res;
    for (int cz = c0[2]; cz <= c1[2]; cz++) {
        if (!axis_enters(r.az, r.bz, cz, P.N, P.ip.res, round)) continue;
        for (int cy = c0[1]; cy <= c1[1]; cy++) {
            if (!axis_enters(r.ay, r.by, cy, P.N, P.ip.res, round)) continue;
            for (int cx = c0[0]; cx <= c1[0]; cx++) {
                if (!axis_enters(r.ax, r.bx, cx, P.N, P.ip.res, round)) continue;
                const int idx = cloud_table_find(C, pack_id(cx, cy, cz));
                if (idx < 0) continue;
                if (!FILL) {
                    atomicAdd(&C.offsets[idx], 1);
                } else {
                    const int at = C.offsets[idx] + atomicAdd(&C.cursors[idx], 1);
                    if (at < C.pairs_capacity) C.pairs[at] = p;
                }
            }
        }
    }
}

// ---- per chunk: point indices into cloud order --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cloud_sort_kernel(CloudParams P, CloudView C) {
    extern __shared__ unsigned s_bits[];  // words_per_pass
    __shared__ int s_scan[256];
    const int tid = threadIdx.x;
    const int n_listed = min(C.ctl[0], CLOUD_MAX_LISTED);
    const int total_words = (P.n_points + 31) >> 5;
    const int pass_words = min(total_words, CLOUD_SORT_WORDS);
    for (int idx = blockIdx.x; idx < n_listed; idx += gridDim.x) {
        const int off = C.offsets[idx];
        const int cnt = min(C.offsets[idx + 1], C.pairs_capacity) - off;
        if (cnt <= 0) continue;
        int written = 0;
        for (int w0 = 0; w0 < total_words; w0 += pass_words) {
            const int words = min(pass_words, total_words - w0);
            for (int i = tid; i < words; i += 256) s_bits[i] = 0u;
            __syncthreads();
            for (int i = tid; i < cnt; i += 256) {
                const int q = C.pairs[off + i] - (w0 << 5);
                if (q >= 0 && q < (words << 5)) atomicOr(&s_bits[q >> 5], 1u << (q & 31));
            }
            __syncthreads();
            const int per = (words + 255) >> 8;
            const int first = min(tid * per, words), last = min(first + per, words);
            int c = 0;
            for (int i = first; i < last; i++) c += __popc(s_bits[i]);
            s_scan[tid] = c;
            __syncthreads();
            for (int o = 1; o < 256; o <<= 1) {
                const int t = tid >= o ? s_scan[tid - o] : 0;
                __syncthreads();
                s_scan[tid] += t;
                __syncthreads();
            }
            int at = off + written + s_scan[tid] - c;
            for (int i = first; i < last; i++) {
                unsigned bits = s_bits[i];
                while (bits) {
                    const int bpos = __ffs(bits) - 1;
                    bits &= bits - 1u;
                    C.sorted[at++] = ((w0 + i) << 5) + bpos;
                }
            }
            written += s_scan[255];
            __syncthreads();
        }
    }
}

// ---- per chunk: the update ----------------------------------------------------------------------------------------------------
template <int N>
struct CloudGeom {
    static constexpr int BOX = N < 16 ? N : 16;      // voxel box held in LDS
    static constexpr int SUB = N / BOX;              // boxes per chunk edge (N = 32: 2)
    static constexpr int BV = BOX * BOX * BOX;
    static constexpr int HITS = 3 * BOX - 2;         // cells a monotone walk can take inside the box
    static constexpr int HP = (HITS + 1) & ~1;       // row pitch of the cell list
    static constexpr int BITS = BOX == 16 ? 4 : 3;   // bits per local coordinate
};

template <int N, bool COLOR>
__global__ __launch_bounds__(256) void cloud_integrate_kernel(CloudParams P, MapView M, const MapView *__restrict__ Mc, CloudView C) {
    using G = CloudGeom<N>;
    constexpr int V = N * N * N;
    __shared__ float s_sdf[G::BV];
    __shared__ float s_wgt[G::BV];
    __shared__ unsigned s_col[COLOR ? G::BV : 1];
    constexpr int RB = CLOUD_RAYS;
    __shared__ unsigned short s_cells[RB * G::HP];
    __shared__ float s_depth[RB], s_trunc[RB], s_weight[RB];
    __shared__ unsigned s_rgb[RB];
    __shared__ unsigned char s_ncells[RB], s_owners[RB];
    __shared__ int s_slot;
    __shared__ unsigned s_updated;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_listed = min(C.ctl[0], CLOUD_MAX_LISTED);
    const float res = P.ip.res, round = 1.0f / res;
    unsigned n_hits = 0, n_sdf = 0, n_carved = 0, n_col = 0, n_new = 0, n_updated = 0, n_items = 0;

    for (int idx = blockIdx.x; idx < n_listed; idx += gridDim.x) {
        const int off = C.offsets[idx];
        const int cnt = min(C.offsets[idx + 1], C.pairs_capacity) - off;
        if (tid == 0) n_items++;
        if (cnt <= 0) continue;  // a listed chunk no ray enters: new and untouched, or resident and unchanged
        int cx, cy, cz;
        unpack_id(C.listed[idx], cx, cy, cz);
        __syncthreads();  // the previous chunk's readers of s_slot are done
        if (tid == 0) s_slot = find_chunk(Mc, cx, cy, cz);
        __syncthreads();
        int slot = s_slot;
        const float ox = (float)(N * cx) * res, oy = (float)(N * cy) * res, oz = (float)(N * cz) * res;  // Chunk.cpp:43
        bool chunk_updated = false;

        for (int sb = 0; sb < G::SUB * G::SUB * G::SUB; sb++) {
            const int bx = (sb % G::SUB) * G::BOX, by = ((sb / G::SUB) % G::SUB) * G::BOX, bz = (sb / (G::SUB * G::SUB)) * G::BOX;
            // voxel box into LDS (a chunk that does not exist yet holds default voxels: DistVoxel.cpp:27-31, ColorVoxel.cpp:27-31)
            for (int i = tid; i < G::BV; i += 256) {
                const int lx = i % G::BOX, ly = (i / G::BOX) % G::BOX, lz = i / (G::BOX * G::BOX);
                const size_t g = (size_t)((bz + lz) * N + (by + ly)) * N + (bx + lx);
                s_sdf[i] = slot >= 0 ? M.sdf[(size_t)slot * V + g] : 99999.0f;
                s_wgt[i] = slot >= 0 ? M.wgt[(size_t)slot * V + g] : 0.0f;
                if (COLOR) s_col[i] = slot >= 0 ? *reinterpret_cast<const unsigned *>(&M.rgbw[(size_t)slot * V + g]) : 0u;
            }
            if (tid == 0) s_updated = 0u;
            __syncthreads();

            for (int b0 = 0; b0 < cnt; b0 += RB) {
                // ---- walk: one ray per lane, its cells inside the box go to LDS (ProjectionIntegrator.cpp:73-84)
                int nc = 0;
                unsigned owners = 0u;
                if (tid < RB && b0 + tid < cnt) {
                    const int p = C.sorted[off + b0 + tid];
                    const CloudRay r = C.rays[p];
                    s_depth[tid] = r.depth;
                    s_trunc[tid] = r.trunc;
                    s_weight[tid] = constant_weight(P.ip.weight, r.trunc);
                    s_rgb[tid] = (COLOR && P.with_color) ? C.rgb[p] : 0u;
                    RayWalk w;
                    if (r.ax == r.ax && w.begin(((r.ax - ox)) * round, ((r.ay - oy)) * round, ((r.az - oz)) * round, ((r.bx - ox)) * round,
                                                ((r.by - oy)) * round, ((r.bz - oz)) * round)) {
                        if (w.length() > (1ull << 20)) {
                            atomicExch(M.error_flag, CLOUD_ERR_RANGE);
                        } else {
                            do {
                                const unsigned lx = (unsigned)(w.x - bx), ly = (unsigned)(w.y - by), lz = (unsigned)(w.z - bz);
                                if (lx < (unsigned)G::BOX && ly < (unsigned)G::BOX && lz < (unsigned)G::BOX) {
                                    s_cells[tid * G::HP + nc] = (unsigned short)(lx | (ly << G::BITS) | (lz << (2 * G::BITS)));
                                    nc++;
                                    owners |= 1u << (((lz >> (G::BITS - 1)) << 1) | (ly >> (G::BITS - 1)));
                                }
                            } while (w.next());
                        }
                    }
                }
                if (tid < RB) {
                    s_ncells[tid] = (unsigned char)nc;
                    s_owners[tid] = (unsigned char)owners;
                }
                __syncthreads();

                // ---- apply: rays in cloud order; wave `wave` owns the voxels with (z half, y half) == wave
                const int nb = min(RB, cnt - b0);
                unsigned upd = 0u;
                for (int g0 = 0; g0 < nb; g0 += 64) {
                    const unsigned mine = (g0 + lane < nb) ? ((s_owners[g0 + lane] >> wave) & 1u) : 0u;
                    unsigned long long todo = __ballot(mine);
                    while (todo) {
                        const int q = g0 + (__ffsll((long long)todo) - 1);
                        todo &= todo - 1ull;
                        const int n = s_ncells[q];
                        if (lane < n) {
                            const unsigned cell = s_cells[q * G::HP + lane];
                            const unsigned lx = cell & (G::BOX - 1), ly = (cell >> G::BITS) & (G::BOX - 1), lz = cell >> (2 * G::BITS);
                            if ((((lz >> (G::BITS - 1)) << 1) | (ly >> (G::BITS - 1))) == (unsigned)wave) {
                                const int li = (int)((lz * G::BOX + ly) * G::BOX + lx);
                                // centroids[id] + origin (ChunkManager.cpp:50-66): (coordinate * res + res / 2) + origin
                                const float vx = ((float)(int)(bx + lx) * res + P.ip.half_res) + ox;
                                const float vy = ((float)(int)(by + ly) * res + P.ip.half_res) + oy;
                                const float vz = ((float)(int)(bz + lz) * res + P.ip.half_res) + oz;
                                const float depth = s_depth[q], trunc = s_trunc[q];
                                const float u = depth - (affine_row(P.inv + 8, vx, vy, vz) - P.pose[11]);  // :89 / :151
                                n_hits++;
                                if (fabsf(u) < trunc) {
                                    float sdf = s_sdf[li], wg = s_wgt[li];
                                    dist_integrate(sdf, wg, u, s_weight[q]);
                                    s_sdf[li] = sdf;
                                    s_wgt[li] = wg;
                                    if (COLOR && P.with_color) {
                                        const unsigned c = s_col[li], rgb = s_rgb[q];
                                        const uchar4 o = color_integrate(make_uchar4(c & 0xffu, (c >> 8) & 0xffu, (c >> 16) & 0xffu, c >> 24),
                                                                         rgb & 0xffu, (rgb >> 8) & 0xffu, (rgb >> 16) & 0xffu, 1);
                                        s_col[li] = (unsigned)o.x | ((unsigned)o.y << 8) | ((unsigned)o.z << 16) | ((unsigned)o.w << 24);
                                        n_col++;
                                    }
                                    upd = 1u;
                                    n_sdf++;
                                } else if (P.ip.carving && u > trunc + P.ip.carving_dist) {
                                    float sdf = s_sdf[li], wg = s_wgt[li];
                                    if (wg > 0.0f) {
                                        dist_integrate(sdf, wg, 1.0e-5f, 5.0f);  // :100 / :165
                                        s_sdf[li] = sdf;
                                        s_wgt[li] = wg;
                                        upd = 1u;
                                        n_carved++;
                                    }
                                }
                            }
                        }
                    }
                }
                if (__any((int)upd) && lane == 0) atomicOr(&s_updated, 1u);
                __syncthreads();  // cells and per-ray values are rewritten by the next rays
            }

            if (s_updated) {  // block-uniform (read after the barrier above)
                if (slot < 0) {
                    if (tid == 0) s_slot = create_chunk(Mc, cx, cy, cz);
                    __syncthreads();
                    slot = s_slot;
                    if (slot >= 0 && tid == 0) n_new++;
                }
                if (slot >= 0) {
                    for (int i = tid; i < G::BV; i += 256) {
                        const int lx = i % G::BOX, ly = (i / G::BOX) % G::BOX, lz = i / (G::BOX * G::BOX);
                        const size_t g = (size_t)((bz + lz) * N + (by + ly)) * N + (bx + lx);
                        M.sdf[(size_t)slot * V + g] = s_sdf[i];
                        M.wgt[(size_t)slot * V + g] = s_wgt[i];
                        if (COLOR) *reinterpret_cast<unsigned *>(&M.rgbw[(size_t)slot * V + g]) = s_col[i];
                    }
                    chunk_updated = true;
                }
            }
            __syncthreads();  // the box is reloaded for the next sub-box
        }
        if (chunk_updated && tid == 0) {
            M.slot_dirty[slot] = 1;  // Chisel.cpp:135-147 (27 neighbours: expanded by the mesher)
            n_updated++;
        }
    }

    // counters (same rows as integrate_kernel: sdf, col, -, probe = cells visited, carved, work chunks, new, updated, frames)
    unsigned long long *row = M.block_counters + (size_t)blockIdx.x * 16;
    unsigned vals[4] = {n_sdf, n_col, n_hits, n_carved};
    const int where[4] = {0, 1, 3, 4};
    for (int k = 0; k < 4; k++) {
        unsigned v = vals[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if (lane == 0 && v) atomicAdd(&row[where[k]], (unsigned long long)v);
    }
    if (tid == 0) {
        if (n_items) atomicAdd(&row[5], (unsigned long long)n_items);
        if (n_new) atomicAdd(&row[6], (unsigned long long)n_new);
        if (n_updated) atomicAdd(&row[7], (unsigned long long)n_updated);
        if (blockIdx.x == 0) atomicAdd(&row[8], 1ull);
    }
}

// known-answer kernel: n rays (start xyz, end xyz) walked inside [lo, hi); cells of ray i at cells[i * cap ..], count[i] = cells met
__global__ void kat_raycast_kernel(const float *rays, int n, int3 lo, int3 hi, int *cells, int cap, int *count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    RayWalk w;
    int c = 0;
    if (w.begin(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2], rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]) && w.length() <= 100000ull) {
        do {
            if (w.x >= lo.x && w.x < hi.x && w.y >= lo.y && w.y < hi.y && w.z >= lo.z && w.z < hi.z) {
                if (c < cap) {
                    cells[((size_t)i * cap + c) * 3] = w.x;
                    cells[((size_t)i * cap + c) * 3 + 1] = w.y;
                    cells[((size_t)i * cap + c) * 3 + 2] = w.z;
                }
                c++;
            }
        } while (w.next());
    }
    count[i] = c;
}

}  // namespace chisel_hip
