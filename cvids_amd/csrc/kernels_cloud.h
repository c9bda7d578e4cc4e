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
//   cloud_bin_kernel<FILL>  per point: the units (boxes of 8 x 8 x 16 voxels of listed chunks) the voxel walk can enter (exact
//                           per-axis cell ranges); first pass counts per unit, second pass (after the prefix sum) writes the
//                           (unit, point) pairs; counts are aggregated per workgroup in LDS first
//   cloud_sort_kernel       per unit: its points into cloud order (LDS bitmap over the range of point indices it holds)
//   cloud_integrate_kernel  per chunk, one wave per unit: the unit's voxels in registers, 64 rays walked at a time (one per lane)
//                           setting their bit in a per-voxel mask in LDS, then every lane applies the rays of its voxels in bit
//                           order = cloud order (step 2); the chunk is created when the first update happens (same outcome as
//                           create-then-erase, step 3)
#pragma once
#include "chisel_device.h"

namespace chisel_hip {

constexpr int CLOUD_TILE = 256;                  // points per workgroup in the per-point kernels
constexpr unsigned CLOUD_TABLE_SLOTS = 1u << 16; // open-addressing table of the listed chunks of one cloud
constexpr int CLOUD_MAX_LISTED = 1 << 14;
constexpr int CLOUD_PAIRS_PER_POINT = 16;        // capacity of the (chunk, point) list, per point of the cloud
constexpr int CLOUD_MAX_RANGE = 4096;            // chunk boxes around one ray that are looked at
constexpr int CLOUD_SORT_WORDS = 8192;           // LDS bitmap of cloud_sort_kernel (32 KB): 262 144 point indices per pass
#ifndef CLOUD_DEPTH_16
#define CLOUD_DEPTH_16 4                         // unit depth for 16-voxel chunks (8, 4 or 2; measured in DESIGN.md 3.3)
#endif
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
    int jaxis;               // world axis each lane of cloud_integrate_kernel keeps in registers (a speed choice, host_cloud.h)
    int depth;               // voxels per lane along that axis = unit size on it (cloud_unit_depth)
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
    int *offsets;            // listed chunks x units per chunk + 1: pairs per unit, then their exclusive prefix
    int *cursors;            // listed chunks x units per chunk
    int *pairs;              // pairs_capacity point indices, grouped by unit, unordered
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
// the table holds packed id + 1: an empty entry is 0, so the table, the counters and the control words are cleared by ONE memset
constexpr uint64_t CLOUD_EMPTY = 0;
__device__ inline void cloud_table_insert(const CloudView &C, const MapView &M, uint64_t key) {
    const uint64_t stored = key + 1;
    unsigned i = cloud_table_home(key);
    for (unsigned probe = 0; probe < CLOUD_TABLE_SLOTS; probe++, i = (i + 1u) & (CLOUD_TABLE_SLOTS - 1u)) {
        // read at L2: a line cached before another CU's insertion would send every later wave into the compare-and-swap
        const uint64_t k = __hip_atomic_load(&C.table_keys[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k == stored) return;
        if (k == CLOUD_EMPTY) {
            const uint64_t old = atomicCAS((unsigned long long *)&C.table_keys[i], (unsigned long long)CLOUD_EMPTY, (unsigned long long)stored);
            if (old == CLOUD_EMPTY) {
                const int idx = atomicAdd(&C.ctl[0], 1);
                if (idx < CLOUD_MAX_LISTED) {
                    C.listed[idx] = key;
                    C.table_vals[i] = idx;  // read by later kernels only
                } else {
                    C.table_vals[i] = -1;
                    raise_error(M.error_flag, CLOUD_ERR_CAPACITY);
                }
                return;
            }
            if (old == stored) return;
        }
    }
    raise_error(M.error_flag, CLOUD_ERR_CAPACITY);
}
__device__ inline int cloud_table_find(const CloudView &C, uint64_t key) {
    const uint64_t stored = key + 1;
    unsigned i = cloud_table_home(key);
    for (unsigned probe = 0; probe < CLOUD_TABLE_SLOTS; probe++, i = (i + 1u) & (CLOUD_TABLE_SLOTS - 1u)) {
        const uint64_t k = C.table_keys[i];
        if (k == stored) return C.table_vals[i];
        if (k == CLOUD_EMPTY) return -1;
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
// exclusive prefix of n = min(*n_ptr, n_fixed) * mult ints in place, total into data[n] and *total (one workgroup of 1024 threads)
__global__ __launch_bounds__(1024) void cloud_scan_kernel(int *data, const int *n_ptr, int n_fixed, int mult, int *total, int capacity,
                                                           int *error_flag) {
    __shared__ int s_part[1024];
    __shared__ int s_carry;
    const int n = (n_ptr ? min(*n_ptr, n_fixed) : n_fixed) * mult;
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
        if (capacity > 0 && s_carry > capacity) raise_error(error_flag, CLOUD_ERR_CAPACITY);
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
    int before = P.with_color ? C.tile_prefix[blockIdx.x] : 0;  // (the index only selects a colour)
    for (int w = 0; w < wave; w++) before += s_wave[w];
    const int cidx = before + __popcll(b & ((1ull << lane) - 1ull));
    bool walking = false;
    RayWalk w;
    if (live) {
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
        if (!(len > P.max_dist)) {
            const float cs = (float)P.N * P.ip.res;  // chunkSize.x() * voxelResolutionMeters
            const float round = 1.0f / cs;
            if (w.begin((wx - dirx * P.truncation) * round, (wy - diry * P.truncation) * round, (wz - dirz * P.truncation) * round,
                        (wx + dirx * P.truncation) * round, (wy + diry * P.truncation) * round, (wz + dirz * P.truncation) * round)) {
                const int lim = ID_BIAS - 2;
                if (w.length() > 4096ull || abs(w.x) > lim || abs(w.y) > lim || abs(w.z) > lim || abs(w.ex) > lim || abs(w.ey) > lim || abs(w.ez) > lim)
                    raise_error(M.error_flag, CLOUD_ERR_RANGE);
                else
                    walking = true;
            }
        }
    }
    // the lanes of a wave step together: neighbouring points meet the same chunks in the same order, so a lane whose left
    // neighbour holds the same id leaves the insertion to it
    while (__any((int)walking)) {
        unsigned long long key = KEY_EMPTY;
        if (walking && chunk_owner(w.x, w.y, w.z, P.ip.n_shards, P.ip.shard_block) == P.ip.shard_rank) key = pack_id(w.x, w.y, w.z);
        const unsigned long long left = __shfl_up(key, 1);
        if (key != KEY_EMPTY && !(lane > 0 && left == key)) cloud_table_insert(C, M, key);
        if (walking) walking = w.next();
    }
}

// ---- units: a chunk is split into boxes of 8 x 8 voxels across the lanes and `depth` voxels along the register axis, one wave each
// (smaller units = fewer rays per wave: the longest ray list of a unit bounds the kernel)
__host__ __device__ constexpr int cloud_unit_depth(int N) { return N == 16 ? CLOUD_DEPTH_16 : (N == 8 ? 4 : 8); }
struct CloudUnits {
    int edge[3];     // unit size per world axis
    int per[3];      // units per chunk edge
    int count;       // units per chunk
    __host__ __device__ CloudUnits(int N, int jaxis, int depth) {
        for (int k = 0; k < 3; k++) {
            edge[k] = k == jaxis ? depth : 8;
            per[k] = N / edge[k];
        }
        count = per[0] * per[1] * per[2];
    }
    __host__ __device__ int index(int ux, int uy, int uz) const { return (uz * per[1] + uy) * per[0] + ux; }
};

// ---- (unit, point) pairs -------------------------------------------------------------------------------------------------------
// Cells of one axis the voxel walk of ray (a, b) takes relative to chunk coordinate c: the walk moves monotonically from
// floor((a - o) * round) to floor((b - o) * round) -- the arithmetic of ProjectionIntegrator.cpp:73-80 with o = Chunk::GetOrigin()
// (Chunk.cpp:43) -- so it enters [0, N) on this axis iff that closed range meets it.  false: it does not.
__device__ inline bool axis_range(float a, float b, int c, int N, float res, float round, int &lo, int &hi) {
    const float o = (float)(N * c) * res;
    const float s = (a - o) * round, e = (b - o) * round;
    if (!(cell_coordinate_ok(s) && cell_coordinate_ok(e))) return false;
    const int si = floor_int(s), ei = floor_int(e);
    lo = max(min(si, ei), 0);
    hi = min(max(si, ei), N - 1);
    return lo <= hi;
}
// calls f(unit) for every unit (listed chunk index * units per chunk + box) the voxel walk of point p can enter
template <class F>
__device__ inline void cloud_enumerate(const CloudParams &P, const CloudView &C, const MapView &M, int p, bool report, F f) {
    const CloudRay r = C.rays[p];
    // skipped points (NaN in ax) and rays with a coordinate that is not finite meet no cell (RayWalk::begin)
    if (!(isfinite(r.ax) && isfinite(r.ay) && isfinite(r.az) && isfinite(r.bx) && isfinite(r.by) && isfinite(r.bz))) return;
    const float a[3] = {r.ax, r.ay, r.az}, b[3] = {r.bx, r.by, r.bz};
    const float cs = (float)P.N * P.ip.res;
    int c0[3], c1[3];
    for (int k = 0; k < 3; k++) {
        const float f0 = floorf(fminf(a[k], b[k]) / cs), f1 = floorf(fmaxf(a[k], b[k]) / cs);
        const float lim = (float)(ID_BIAS - 4);
        if (!(f0 >= -lim && f1 <= lim)) {
            if (report) raise_error(M.error_flag, CLOUD_ERR_RANGE);
            return;
        }
        c0[k] = (int)f0 - 1;  // one chunk of slack: the decisive test below uses the reference's own arithmetic
        c1[k] = (int)f1 + 1;
    }
    if ((long long)(c1[0] - c0[0] + 1) * (c1[1] - c0[1] + 1) * (c1[2] - c0[2] + 1) > CLOUD_MAX_RANGE) {
        if (report) raise_error(M.error_flag, CLOUD_ERR_RANGE);
        return;
    }
    const CloudUnits U(P.N, P.jaxis, P.depth);
    const float round = 1.0f / P.ip.res;
    for (int cz = c0[2]; cz <= c1[2]; cz++) {
        int zl, zh;
        if (!axis_range(r.az, r.bz, cz, P.N, P.ip.res, round, zl, zh)) continue;
        for (int cy = c0[1]; cy <= c1[1]; cy++) {
            int yl, yh;
            if (!axis_range(r.ay, r.by, cy, P.N, P.ip.res, round, yl, yh)) continue;
            for (int cx = c0[0]; cx <= c1[0]; cx++) {
                int xl, xh;
                if (!axis_range(r.ax, r.bx, cx, P.N, P.ip.res, round, xl, xh)) continue;
                const int idx = cloud_table_find(C, pack_id(cx, cy, cz));
                if (idx < 0) continue;
                for (int uz = zl / U.edge[2]; uz <= zh / U.edge[2]; uz++)
                    for (int uy = yl / U.edge[1]; uy <= yh / U.edge[1]; uy++)
                        for (int ux = xl / U.edge[0]; ux <= xh / U.edge[0]; ux++) f(idx * U.count + U.index(ux, uy, uz));
            }
        }
    }
}

// Workgroup-local table unit -> count: the 256 consecutive points of a workgroup meet a few dozen units, so the global counters
// take one atomic per (workgroup, unit) instead of one per (point, unit) (same-address atomics serialise in L2).
constexpr int CLOUD_LOCAL_SLOTS = 256;
__device__ inline int cloud_local_slot(int *s_key, int key) {
    unsigned h = ((unsigned)key * 2654435761u) >> 24;
    for (int probe = 0; probe < CLOUD_LOCAL_SLOTS; probe++, h = (h + 1u) & (CLOUD_LOCAL_SLOTS - 1u)) {
        const int k = s_key[h];
        if (k == key) return (int)h;
        if (k == -1) {
            const int old = atomicCAS(&s_key[h], -1, key);
            if (old == -1 || old == key) return (int)h;
        }
    }
    return -1;
}
template <bool FILL>
__global__ __launch_bounds__(CLOUD_TILE) void cloud_bin_kernel(CloudParams P, CloudView C, MapView M) {
    __shared__ int s_key[CLOUD_LOCAL_SLOTS], s_cnt[CLOUD_LOCAL_SLOTS], s_base[CLOUD_LOCAL_SLOTS];
    const int tid = threadIdx.x;
    const int p = blockIdx.x * CLOUD_TILE + tid;
    const bool live = p < P.n_points;
    for (int i = tid; i < CLOUD_LOCAL_SLOTS; i += CLOUD_TILE) {
        s_key[i] = -1;
        s_cnt[i] = 0;
    }
    __syncthreads();
    if (live)
        cloud_enumerate(P, C, M, p, FILL, [&](int unit) {
            const int e = cloud_local_slot(s_key, unit);
            if (e >= 0) atomicAdd(&s_cnt[e], 1);
            else if (!FILL) atomicAdd(&C.offsets[unit], 1);
        });
    __syncthreads();
    for (int i = tid; i < CLOUD_LOCAL_SLOTS; i += CLOUD_TILE) {
        if (s_key[i] < 0) continue;
        if (!FILL) {
            atomicAdd(&C.offsets[s_key[i]], s_cnt[i]);
        } else {
            s_base[i] = C.offsets[s_key[i]] + atomicAdd(&C.cursors[s_key[i]], s_cnt[i]);
            s_cnt[i] = 0;
        }
    }
    if (!FILL) return;
    __syncthreads();
    if (live)
        cloud_enumerate(P, C, M, p, false, [&](int unit) {
            const int e = cloud_local_slot(s_key, unit);
            const int at = e >= 0 ? s_base[e] + atomicAdd(&s_cnt[e], 1) : C.offsets[unit] + atomicAdd(&C.cursors[unit], 1);
            if (at < C.pairs_capacity) C.pairs[at] = p;
        });
}

// ---- per unit: point indices into cloud order ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cloud_sort_kernel(CloudParams P, CloudView C) {
    __shared__ unsigned s_bits[CLOUD_SORT_WORDS];
    __shared__ int s_scan[256];
    __shared__ int s_lo, s_hi;
    const int tid = threadIdx.x;
    const int n_units = min(C.ctl[0], CLOUD_MAX_LISTED) * CloudUnits(P.N, P.jaxis, P.depth).count;
    for (int unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        const int off = C.offsets[unit];
        const int cnt = min(C.offsets[unit + 1], C.pairs_capacity) - off;
        if (cnt <= 0) continue;
        // the unit's points lie in a narrow index range (a few image rows): bitmap over that range only
        if (tid == 0) {
            s_lo = 0x7fffffff;
            s_hi = -1;
        }
        __syncthreads();
        int lo = 0x7fffffff, hi = -1;
        for (int i = tid; i < cnt; i += 256) {
            const int q = C.pairs[off + i];
            lo = min(lo, q);
            hi = max(hi, q);
        }
        for (int o = 32; o > 0; o >>= 1) {
            lo = min(lo, __shfl_down(lo, o));
            hi = max(hi, __shfl_down(hi, o));
        }
        if ((tid & 63) == 0) {
            atomicMin(&s_lo, lo);
            atomicMax(&s_hi, hi);
        }
        __syncthreads();
        const int w_first = s_lo >> 5, total_words = (s_hi >> 5) - w_first + 1;
        int written = 0;
        for (int w0 = 0; w0 < total_words; w0 += CLOUD_SORT_WORDS) {
            const int words = min(CLOUD_SORT_WORDS, total_words - w0);
            const int bit0 = (w_first + w0) << 5;
            for (int i = tid; i < words; i += 256) s_bits[i] = 0u;
            __syncthreads();
            for (int i = tid; i < cnt; i += 256) {
                const int q = C.pairs[off + i] - bit0;
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
                    C.sorted[at++] = bit0 + (i << 5) + bpos;
                }
            }
            written += s_scan[255];
            __syncthreads();
        }
    }
}

// ---- per chunk: the update ----------------------------------------------------------------------------------------------------
// One wave per unit (box of 8 x 8 x D voxels, D along world axis `jaxis`); the waves of a workgroup take units of the same chunk
// and meet only to look the chunk up, to create it and to move on.  A lane holds the D voxels of one line of its box along `jaxis`
// in REGISTERS for the whole list of rays (lanes = the 8 x 8 positions on the other two axes).  Rays are taken 64 at a time, in cloud order:
//   walk   lane r walks ray r (Raycast.cpp:35-128) and sets bit r in the mask of every voxel of the cube it meets (LDS, 64 bits
//          per voxel);
//   apply  every lane goes through the masks of its voxels and applies the rays whose bits are set, lowest bit first = cloud
//          order, to the voxel in its registers (ProjectionIntegrator.cpp:85-105 / :147-167).
// So the per-voxel order of updates is the reference's, while different voxels advance in parallel.  64 consecutive points of an
// organised cloud are a piece of an image row: their cells spread along the row and along the viewing direction and are thin
// across the rows, so the host picks the world axis closest to the sensor's y axis as `jaxis` and the cells of a batch land on
// many lanes.
template <int N>
struct CloudGeom {
    static constexpr int D = cloud_unit_depth(N);              // voxels per lane
    static constexpr int UV = 64 * D;                          // voxels per unit
    static constexpr int U = (N / 8) * (N / 8) * (N / D);      // units per chunk
    static constexpr int WAVES = U < 16 ? U : 16;              // units in flight per workgroup
};

// ColorVoxel::Integrate(r, g, b, 1) (ColorVoxel.h:65-85) on packed words, any weight: color_integrate_fresh's arithmetic is exact
// up to weight 253 (x + 0.5 < 2^16, quotient <= 255, (x + 0.5) / d at least 1 / 508 away from an integer against an error below
// 1e-4); checked against color_integrate for EVERY (weight, old, new) by chisel_hip_kat_color_any (tests/test_gpu_cloud.py).
__device__ inline unsigned color_integrate_any(unsigned c, unsigned rgb) {
    if ((c >> 24) >= 254u) return c;  // weight >= 255 - weightUpdate
    return color_integrate_fresh(c, rgb);
}
__global__ void kat_color_any_kernel(unsigned *mismatches) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;  // 256 * 256 * 256 cases
    const unsigned w = i >> 16, o = (i >> 8) & 0xffu, n = i & 0xffu;
    const uchar4 c = make_uchar4((uint8_t)o, (uint8_t)(255u - o), (uint8_t)(o ^ 0x5au), (uint8_t)w);
    const uint8_t r = (uint8_t)n, g = (uint8_t)(255u - n), b = (uint8_t)(n ^ 0x5au);
    const uchar4 want = color_integrate(c, r, g, b, 1);
    const unsigned got = color_integrate_any((unsigned)c.x | ((unsigned)c.y << 8) | ((unsigned)c.z << 16) | ((unsigned)c.w << 24),
                                             (unsigned)r | ((unsigned)g << 8) | ((unsigned)b << 16));
    const unsigned want_bits = (unsigned)want.x | ((unsigned)want.y << 8) | ((unsigned)want.z << 16) | ((unsigned)want.w << 24);
    if (got != want_bits) atomicAdd(mismatches, 1u);
}

template <int N, bool COLOR>
__global__ __launch_bounds__(64 * CloudGeom<N>::WAVES) void cloud_integrate_kernel(CloudParams P, MapView M, const MapView *__restrict__ Mc,
                                                                                    CloudView C) {
    using G = CloudGeom<N>;
    constexpr int V = N * N * N;
    __shared__ unsigned s_mask[G::WAVES][G::UV][2];
    __shared__ float4 s_ray[G::WAVES][64];  // depth, truncation, weight update, colour bits
    __shared__ int s_slot;
    __shared__ unsigned s_upd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_listed = min(C.ctl[0], CLOUD_MAX_LISTED);
    const float res = P.ip.res, round = 1.0f / res;
    // axis roles: ja in registers, (aa, ab) across the lanes
    const int ja = P.jaxis, aa = ja == 0 ? 1 : 0, ab = ja == 2 ? 1 : 2;
    const int la = lane & 7, lb = lane >> 3;
    int lane_xyz[3], mask_mult[3];
    lane_xyz[ja] = 0; lane_xyz[aa] = la; lane_xyz[ab] = lb;
    mask_mult[ja] = 64; mask_mult[aa] = 1; mask_mult[ab] = 8;
    const int j_stride = ja == 0 ? 1 : (ja == 1 ? N : N * N);
    const CloudUnits units(N, ja, G::D);
    unsigned n_hits = 0, n_sdf = 0, n_carved = 0, n_col = 0, n_new = 0, n_updated = 0, n_items = 0;

    for (int idx = blockIdx.x; idx < n_listed; idx += gridDim.x) {
        if (tid == 0) n_items++;
        const int total = min(C.offsets[(idx + 1) * G::U], C.pairs_capacity) - C.offsets[idx * G::U];
        if (total <= 0) continue;  // a listed chunk no ray enters: new and untouched, or resident and unchanged
        int cx, cy, cz;
        unpack_id(C.listed[idx], cx, cy, cz);
        __syncthreads();  // the previous chunk's readers of s_slot / s_upd are done
        if (tid == 0) {
            s_slot = find_chunk(Mc, cx, cy, cz);
            s_upd = 0u;
        }
        __syncthreads();
        int slot = s_slot;
        const float ox = (float)(N * cx) * res, oy = (float)(N * cy) * res, oz = (float)(N * cz) * res;  // Chunk.cpp:43

        for (int round_i = 0; round_i < G::U / G::WAVES; round_i++) {
            const int u = round_i * G::WAVES + wave;
            const int bx = (u % units.per[0]) * units.edge[0], by = ((u / units.per[0]) % units.per[1]) * units.edge[1],
                      bz = (u / (units.per[0] * units.per[1])) * units.edge[2];
            const int off = C.offsets[idx * G::U + u];
            const int cnt = min(C.offsets[idx * G::U + u + 1], C.pairs_capacity) - off;
            float sdf[G::D], wgt[G::D];
            unsigned col[COLOR ? G::D : 1];
            bool upd = false;
            // first voxel of the lane's line, in chunk coordinates
            const int x0 = bx + lane_xyz[0], y0 = by + lane_xyz[1], z0 = bz + lane_xyz[2];
            const size_t g0 = (size_t)(z0 * N + y0) * N + x0;
            if (cnt > 0) {
                // the lane's voxels into registers (a chunk that does not exist yet holds default voxels: DistVoxel.cpp:27-31,
                // ColorVoxel.cpp:27-31)
#pragma unroll
                for (int j = 0; j < G::D; j++) {
                    const size_t g = (size_t)slot * V + g0 + (size_t)j * j_stride;
                    sdf[j] = slot >= 0 ? M.sdf[g] : 99999.0f;
                    wgt[j] = slot >= 0 ? M.wgt[g] : 0.0f;
                    if (COLOR) col[j] = slot >= 0 ? *reinterpret_cast<const unsigned *>(&M.rgbw[g]) : 0u;
                }
                for (int i = lane; i < G::UV; i += 64) {
                    s_mask[wave][i][0] = 0u;
                    s_mask[wave][i][1] = 0u;
                }

                // the next batch's rays are fetched while the current batch is applied (two dependent global reads per batch)
                CloudRay r_next;
                unsigned rgb_next = 0u;
                r_next.ax = __builtin_nanf("");
                if (lane < cnt) {
                    const int p = C.sorted[off + lane];
                    r_next = C.rays[p];
                    if (COLOR && P.with_color) rgb_next = C.rgb[p];
                }
                for (int b0 = 0; b0 < cnt; b0 += 64) {
                    // ---- walk: one ray per lane (ProjectionIntegrator.cpp:73-84)
#ifndef CLOUD_ABLATE_WALK
                    const CloudRay r = r_next;
                    const unsigned rgb = rgb_next;
                    const bool have = b0 + lane < cnt;
                    if (b0 + 64 + lane < cnt) {
                        const int p = C.sorted[off + b0 + 64 + lane];
                        r_next = C.rays[p];
                        if (COLOR && P.with_color) rgb_next = C.rgb[p];
                    }
                    if (have) {
                        s_ray[wave][lane] = make_float4(r.depth, r.trunc, constant_weight(P.ip.weight, r.trunc), __uint_as_float(rgb));
                        RayWalk w;
                        if (r.ax == r.ax && w.begin((r.ax - ox) * round, (r.ay - oy) * round, (r.az - oz) * round, (r.bx - ox) * round,
                                                    (r.by - oy) * round, (r.bz - oz) * round)) {
                            if (w.length() > (1ull << 20)) {
                                raise_error(M.error_flag, CLOUD_ERR_RANGE);
                            } else {
                                do {
                                    const unsigned cxl = (unsigned)(w.x - bx), cyl = (unsigned)(w.y - by), czl = (unsigned)(w.z - bz);
                                    if (cxl < (unsigned)units.edge[0] && cyl < (unsigned)units.edge[1] && czl < (unsigned)units.edge[2])
                                        atomicOr(&s_mask[wave][cxl * mask_mult[0] + cyl * mask_mult[1] + czl * mask_mult[2]][lane >> 5],
                                                 1u << (lane & 31));
                                } while (w.next());
                            }
                        }
                    }
#endif
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // same wave: LDS operations complete in order
                    __builtin_amdgcn_wave_barrier();
#ifndef CLOUD_ABLATE_APPLY
                    // ---- apply: per voxel, the rays of this batch that met it, in cloud order
#pragma unroll
                    for (int j = 0; j < G::D; j++) {
                        unsigned m0 = s_mask[wave][j * 64 + lane][0], m1 = s_mask[wave][j * 64 + lane][1];
                        if ((m0 | m1) == 0u) continue;
                        s_mask[wave][j * 64 + lane][0] = 0u;
                        s_mask[wave][j * 64 + lane][1] = 0u;
                        // centroids[id] + origin (ChunkManager.cpp:50-66): (coordinate * res + res / 2) + origin, then the z row of
                        // inversePose * centroid (Transform * Vec3: ((m0 x + m1 y) + m2 z) + m3) minus cameraPose.translation().z
                        // (ProjectionIntegrator.cpp:89 / :151)
                        const float vx = ((float)(x0 + (ja == 0 ? j : 0)) * res + P.ip.half_res) + ox;
                        const float vy = ((float)(y0 + (ja == 1 ? j : 0)) * res + P.ip.half_res) + oy;
                        const float vz = ((float)(z0 + (ja == 2 ? j : 0)) * res + P.ip.half_res) + oz;
                        const float cam_z = affine_row(P.inv + 8, vx, vy, vz) - P.pose[11];
                        unsigned long long m = (unsigned long long)m0 | ((unsigned long long)m1 << 32);
                        while (m) {
                            const int q = __ffsll((long long)m) - 1;
                            m &= m - 1ull;
                            const float4 rp = s_ray[wave][q];
                            const float u_sd = rp.x - cam_z;
                            n_hits++;
                            if (fabsf(u_sd) < rp.y) {
                                dist_integrate(sdf[j], wgt[j], u_sd, rp.z);
                                if (COLOR && P.with_color) {
                                    col[j] = color_integrate_any(col[j], __float_as_uint(rp.w));
                                    n_col++;
                                }
                                upd = true;
                                n_sdf++;
                            } else if (P.ip.carving && u_sd > rp.y + P.ip.carving_dist) {
                                if (wgt[j] > 0.0f) {
                                    dist_integrate(sdf[j], wgt[j], 1.0e-5f, 5.0f);  // :100 / :165
                                    upd = true;
                                    n_carved++;
                                }
                            }
                        }
                    }
#endif
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            const bool unit_updated = __any((int)upd) != 0;
            if (unit_updated && lane == 0) atomicOr(&s_upd, 1u);
            __syncthreads();
            const unsigned any_updated = s_upd;  // some unit of this chunk so far
            __syncthreads();                     // (a fast wave must not raise the flag of the next round before everybody has read it)
            if (any_updated != 0u && slot < 0) {  // block-uniform
                if (tid == 0) {
                    s_slot = create_chunk(Mc, cx, cy, cz);
                    if (s_slot >= 0) n_new++;
                }
                __syncthreads();
                slot = s_slot;
            }
            if (unit_updated && slot >= 0) {
#pragma unroll
                for (int j = 0; j < G::D; j++) {
                    const size_t g = (size_t)slot * V + g0 + (size_t)j * j_stride;
                    M.sdf[g] = sdf[j];
                    M.wgt[g] = wgt[j];
                    if (COLOR) *reinterpret_cast<unsigned *>(&M.rgbw[g]) = col[j];
                }
            }
        }
        __syncthreads();
        if (tid == 0 && s_upd != 0u && slot >= 0) {
            mark_slot_dirty(M, slot);  // Chisel.cpp:135-147 (27 neighbours: expanded by the mesher)
            slot_summary(M)[slot] = SUM_ANY;  // (this path does not classify what it writes: the mesher looks at the chunk)
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
