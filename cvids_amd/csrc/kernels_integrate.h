// kernels_integrate.h -- projective SDF / weight / colour integration over the work-list.
//
// Replaces ProjectionIntegrator::Integrate<float> (ProjectionIntegrator.h:51-99) and
// ::IntegrateColor<float,uint8_t> (:101-183) plus the allocate-everything / erase-untouched protocol of
// Chisel::IntegrateDepthScan[Color] (Chisel.h:77-108, 133-143, 202-207):
//   - one workgroup per work-list chunk (grid-stride over the device-resident list, no host round trip);
//   - a lane owns 4 consecutive x voxels, so every voxel-plane access of a wave is one contiguous
//     1 KiB segment (float4 per lane) and sdf/weight/colour are only read for quads that can change
//     and only written for quads that did change;
//   - the depth pixels under the chunk are staged once in LDS (the chunk's conservative pixel box
//     from the cull kernel) and gathered from there; boxes too large for the tile buffer fall back
//     to gathers from global memory (near-camera chunks);
//   - a chunk that is not resident is first classified without touching memory; only if some voxel
//     is updated does thread 0 pop a pool slot and insert the id into the hash, and the chunk is
//     then written in full -- the outcome of the reference's "create, integrate, erase if untouched"
//     without ever allocating the ~98 % of candidates that stay untouched;
//   - per-voxel arithmetic follows the reference operation by operation in fp32 (compiled with
//     -ffp-contract=off, IEEE divide), 3-term sums in Eigen's a0 + (a1 + a2) order.
#pragma once
#include "chisel_device.h"

namespace chisel_hip {

constexpr int TILE_MAX_PIXELS = 4096;  // 16 KiB depth tile in LDS per workgroup

struct Tally {
    unsigned sdf, col, colsat, probe, carved;
};

struct TileCtx {
    const float *tile;   // LDS
    int u0, v0, tw, th;  // tile origin / size; tw == 0: no tile
};

// classification of one voxel against the depth image: 0 = untouched, 1 = in band, 2 = carve test
// Out: sd (surfaceDist), wu (weight update), cpix (colour pixel index or -1)
template <bool COLOR>
__device__ inline int classify_voxel(const FrameParams &P, const TileCtx &T, float pcx, float pcy, float pcz, float vx,
                                     float vy, float vz, float &sd, float &wu, int &cpix) {
    const CameraParams &C = P.cam;
    // PinholeCamera::ProjectPoint (PinholeCamera.cpp:38-45)
    const float invZ = 1.0f / pcz;
    const float u = C.fx * pcx * invZ + C.cx;
    const float v = C.fy * pcy * invZ + C.cy;
    // IsPointOnImage (PinholeCamera.cpp:61-64) || z < 0 (ProjectionIntegrator.h:68 / :126)
    const bool on = (u >= 0.0f) && (v >= 0.0f) && (u < (float)C.W) && (v < (float)C.H) && !(pcz < 0.0f);
    cpix = -1;
    sd = 0.0f;
    wu = 1.0f;
    if (!on) return 0;
    const int iu = (int)u, iv = (int)v;  // truncating lookup (:72 / :131)
    float depth;
    const int tu = iu - T.u0, tv = iv - T.v0;
    if ((unsigned)tu < (unsigned)T.tw && (unsigned)tv < (unsigned)T.th) {
        depth = T.tile[tv * T.tw + tu];
    } else {
        depth = P.depth[(size_t)iv * C.W + iu];  // DepthAt(row, col) DepthImage.h:72-76
    }
    if (COLOR) {
        if (depth != depth) return 0;  // :134
    } else {
        if (depth > 50.0f) return 0;  // :74
    }
    const float truncation = truncation_distance(P.trunc_kind, P.trunc_param, depth);
    const float surfaceDist = depth - pcz;
    if (COLOR) {
        if (depth > 100.0f) return 0;  // :141
    }
    sd = surfaceDist;
    if (fabsf(surfaceDist) < truncation + P.diag) {
        if (COLOR) {
            // colour camera projection (:146-147); voxel centre (vx,vy,vz) is in world coordinates
            const CameraParams &K = P.ccam;
            const float dx = vx - K.t[0], dy = vy - K.t[1], dz = vz - K.t[2];
            const float qx = K.R[0] * dx + (K.R[3] * dy + K.R[6] * dz);
            const float qy = K.R[1] * dx + (K.R[4] * dy + K.R[7] * dz);
            const float qz = K.R[2] * dx + (K.R[5] * dy + K.R[8] * dz);
            const float iq = 1.0f / qz;
            const float cu = K.fx * qx * iq + K.cx;
            const float cv = K.fy * qy * iq + K.cy;
            if ((cu >= 0.0f) && (cv >= 0.0f) && (cu < (float)K.W) && (cv < (float)K.H)) cpix = (int)cv * K.W + (int)cu;
            wu = constant_weight(P.weight, truncation);  // :161-162
        }
        return 1;
    }
    if (P.carving && surfaceDist > truncation + P.carving_dist) return 2;
    return 0;
}

template <int N>
struct Geom {
    static constexpr int V = N * N * N;
    static constexpr int QX = N / 4;              // quads per x-row
    static constexpr int QUADS = V / 4;
    static constexpr int BLOCK = (QUADS < 256) ? QUADS : 256;
    static constexpr int PASSES = QUADS / BLOCK;
    static constexpr int PPI = (PASSES < 4) ? PASSES : 4;  // passes in flight per iteration
    static constexpr int ITERS = PASSES / PPI;
};

// camera-space position of the 4 voxels of quad q, exactly as the reference computes it:
//   voxelCenter = centroids[i] + origin            (ChunkManager.cpp:61: Vec3(x,y,z)*res + half; ProjectionIntegrator.h:63)
//   inCamera    = R^T * (voxelCenter - t)          (:64), row i of R^T summed as a0 + (a1 + a2)
template <int N>
__device__ inline void quad_geometry(const FrameParams &P, float ox, float oy, float oz, int q, float (&pcx)[4],
                                     float (&pcy)[4], float (&pcz)[4], float (&wx)[4], float &wy, float &wz) {
    using G = Geom<N>;
    const CameraParams &C = P.cam;
    const int xq = q % G::QX, y = (q / G::QX) % N, z = q / (G::QX * N);
    wy = ((float)y * P.res + P.half_res) + oy;
    wz = ((float)z * P.res + P.half_res) + oz;
    const float dy = wy - C.t[1], dz = wz - C.t[2];
    const float s0 = C.R[3] * dy + C.R[6] * dz;
    const float s1 = C.R[4] * dy + C.R[7] * dz;
    const float s2 = C.R[5] * dy + C.R[8] * dz;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int x = xq * 4 + j;
        wx[j] = ((float)x * P.res + P.half_res) + ox;
        const float dx = wx[j] - C.t[0];
        pcx[j] = C.R[0] * dx + s0;
        pcy[j] = C.R[1] * dx + s1;
        pcz[j] = C.R[2] * dx + s2;
    }
}

template <int N, bool COLOR>
__global__ __launch_bounds__(Geom<N>::BLOCK) void integrate_kernel(FrameParams P, MapView M, const WorkItem *items,
                                                                    const int *work_count, int max_items) {
    using G = Geom<N>;
    __shared__ float s_tile[TILE_MAX_PIXELS];
    __shared__ int s_flag[2];
    __shared__ int s_slot;
    const int tid = threadIdx.x;
    int n_items = *work_count;
    if (n_items > max_items) n_items = max_items;
    Tally tally = {0, 0, 0, 0, 0};
    unsigned n_new = 0, n_updated = 0;

    for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
        const WorkItem wi = items[it];
        // Chunk origin (Chunk.cpp:43): numVoxels * ID (int) * resolution
        const float ox = (float)(N * wi.x) * P.res, oy = (float)(N * wi.y) * P.res, oz = (float)(N * wi.z) * P.res;
        __syncthreads();  // previous item's tile / flags fully consumed
        if (tid < 2) s_flag[tid] = 0;
        // ---- stage the depth pixels under the chunk in LDS -------------------------------------
        TileCtx T;
        T.tile = s_tile;
        T.u0 = wi.u0;
        T.v0 = wi.v0;
        T.tw = 0;
        T.th = 0;
        if (wi.flags & WI_TILE) {
            const int tw = wi.u1 - wi.u0 + 1, th = wi.v1 - wi.v0 + 1;
            if (tw * th <= TILE_MAX_PIXELS) {
                T.tw = tw;
                T.th = th;
                for (int r = tid / 64; r < th; r += G::BLOCK / 64) {  // one wave per tile row: coalesced row segments
                    const float *src = P.depth + (size_t)(wi.v0 + r) * P.cam.W + wi.u0;
                    for (int c = tid & 63; c < tw; c += 64) s_tile[r * tw + c] = src[c];
                }
            }
        }
        __syncthreads();

        int slot = wi.slot;
        const bool fresh = slot < 0;
        if (fresh) {
            // ---- classification only: would any voxel be integrated? (carving cannot touch w == 0 voxels)
            bool any = false;
            for (int pass = 0; pass < G::PASSES; pass++) {
                const int q = pass * G::BLOCK + tid;
                float pcx[4], pcy[4], pcz[4], wx[4], wy, wz;
                quad_geometry<N>(P, ox, oy, oz, q, pcx, pcy, pcz, wx, wy, wz);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float sd, wu;
                    int cpix;
                    any |= classify_voxel<COLOR>(P, T, pcx[j], pcy[j], pcz[j], wx[j], wy, wz, sd, wu, cpix) == 1;
                }
                if ((pass & 3) == 3 || pass == G::PASSES - 1) {
                    if (__syncthreads_or(any)) {
                        any = true;
                        break;
                    }
                }
            }
            if (!any) continue;  // block-uniform: the reference would create and then erase this chunk
            if (tid == 0) {
                // ChunkManager::CreateChunk (ChunkManager.cpp:171-174) on the device
                int s = -1;
                int top = atomicSub(M.free_top, 1) - 1;
                if (top < 0) {
                    atomicAdd(M.free_top, 1);
                    atomicExch(M.error_flag, 1);
                } else {
                    s = M.free_list[top];
                    const uint64_t key = pack_id(wi.x, wi.y, wi.z);
                    const uint64_t h = chunk_hash(wi.x, wi.y, wi.z) & M.hash_mask;
                    bool placed = false;
                    for (uint64_t i = 0; i <= M.hash_mask && !placed; i++) {
                        const uint64_t idx = (h + i) & M.hash_mask;
                        const uint64_t cur = M.hash_keys[idx];
                        if (cur == KEY_EMPTY || cur == KEY_TOMB) {
                            if (atomicCAS((unsigned long long *)&M.hash_keys[idx], (unsigned long long)cur,
                                          (unsigned long long)key) == cur) {
                                M.hash_vals[idx] = s;
                                placed = true;
                            }
                        }
                    }
                    if (!placed) {
                        atomicExch(M.error_flag, 2);
                        s = -1;
                    } else {
                        M.slot_key[s] = key;
                    }
                }
                s_slot = s;
            }
            __syncthreads();
            slot = s_slot;
            if (slot < 0) continue;
            n_new += (tid == 0);
        }

        float *sdf_base = M.sdf + (size_t)slot * G::V;
        float *wgt_base = M.wgt + (size_t)slot * G::V;
        uchar4 *col_base = COLOR ? (M.rgbw + (size_t)slot * G::V) : nullptr;
        bool updated = false;

        for (int iter = 0; iter < G::ITERS; iter++) {
            // ---- phase A: geometry + depth gather + classification for PPI quads -----------------
            unsigned cls[G::PPI];          // 2 bits per voxel
            float sd[G::PPI][4], wu[G::PPI][4];
            int cpix[G::PPI][4];
#pragma unroll
            for (int p = 0; p < G::PPI; p++) {
                const int q = (iter * G::PPI + p) * G::BLOCK + tid;
                float pcx[4], pcy[4], pcz[4], wx[4], wy, wz;
                quad_geometry<N>(P, ox, oy, oz, q, pcx, pcy, pcz, wx, wy, wz);
                cls[p] = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    int c = classify_voxel<COLOR>(P, T, pcx[j], pcy[j], pcz[j], wx[j], wy, wz, sd[p][j], wu[p][j], cpix[p][j]);
                    cls[p] |= (unsigned)c << (2 * j);
                }
            }
            // ---- phase B: load voxel state only where something can change ----------------------
            float4 s4[G::PPI], w4[G::PPI];
            uint4 c4[G::PPI];
#pragma unroll
            for (int p = 0; p < G::PPI; p++) {
                const int q = (iter * G::PPI + p) * G::BLOCK + tid;
                s4[p] = make_float4(99999.0f, 99999.0f, 99999.0f, 99999.0f);  // DistVoxel() DistVoxel.cpp:27-31
                w4[p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                c4[p] = make_uint4(0u, 0u, 0u, 0u);                           // ColorVoxel() ColorVoxel.cpp:27-31
                if (!fresh && cls[p] != 0) {
                    s4[p] = *reinterpret_cast<const float4 *>(sdf_base + 4 * q);
                    w4[p] = *reinterpret_cast<const float4 *>(wgt_base + 4 * q);
                    if (COLOR && (cls[p] & 0x55u)) c4[p] = *reinterpret_cast<const uint4 *>(col_base + 4 * q);
                }
            }
            // ---- phase C: update + write back ----------------------------------------------------
#pragma unroll
            for (int p = 0; p < G::PPI; p++) {
                const int q = (iter * G::PPI + p) * G::BLOCK + tid;
                float s[4] = {s4[p].x, s4[p].y, s4[p].z, s4[p].w};
                float w[4] = {w4[p].x, w4[p].y, w4[p].z, w4[p].w};
                unsigned cw[4] = {c4[p].x, c4[p].y, c4[p].z, c4[p].w};
                bool dchg = false, cchg = false;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int c = (cls[p] >> (2 * j)) & 3;
                    if (c == 1) {
                        if (COLOR) {
                            if (cpix[p][j] >= 0) {
                                uchar4 cv = *reinterpret_cast<uchar4 *>(&cw[j]);
                                if (cv.w < 8) {  // ProjectionIntegrator.h:152
                                    uint8_t r, g, b;
                                    color_at(P.color, cpix[p][j], P.color_channels, r, g, b);
                                    cv = color_integrate(cv, r, g, b, 1);
                                    cw[j] = *reinterpret_cast<unsigned *>(&cv);
                                    cchg = true;
                                    tally.col++;
                                } else {
                                    tally.colsat++;
                                }
                            }
                        }
                        dist_integrate(s[j], w[j], sd[p][j], COLOR ? wu[p][j] : 1.0f);
                        dchg = true;
                        tally.sdf++;
                    } else if (c == 2) {
                        if (!fresh) tally.probe++;
                        if (w[j] > 0.0f && sdf_below_carve_threshold(s[j])) {
                            if (COLOR) {  // :166-177
                                if (w[j] < 5.0f) {
                                    s[j] = 99999.0f;
                                    w[j] = 0.0f;
                                } else {
                                    w[j] = w[j] - 1.0f;
                                }
                            } else {  // :88-95 Carve() == Reset()
                                s[j] = 99999.0f;
                                w[j] = 0.0f;
                            }
                            dchg = true;
                            tally.carved++;
                        }
                    }
                }
                updated |= dchg;
                if (dchg || fresh) {
                    *reinterpret_cast<float4 *>(sdf_base + 4 * q) = make_float4(s[0], s[1], s[2], s[3]);
                    *reinterpret_cast<float4 *>(wgt_base + 4 * q) = make_float4(w[0], w[1], w[2], w[3]);
                }
                if (COLOR && (cchg || fresh)) *reinterpret_cast<uint4 *>(col_base + 4 * q) = make_uint4(cw[0], cw[1], cw[2], cw[3]);
            }
        }
        // ---- "needsUpdate" of the chunk (Chisel.h:85 / :167): mark the slot dirty for the mesher
        if (updated) s_flag[1] = 1;  // benign race: every writer stores 1
        __syncthreads();
        if (tid == 0 && s_flag[1]) {
            M.slot_dirty[slot] = 1;
            n_updated++;
        }
    }

    // ---- counters: block reduction in LDS, then a plain read-modify-write of this block's private row.
    // (Same-address atomics run at ~90 per microsecond on this part: a few thousand of them per frame would
    // cost more than the integration itself.  Rows are summed lazily by reduce_counters_kernel.)
    __syncthreads();
    unsigned *s_cnt = reinterpret_cast<unsigned *>(s_tile);
    if (tid < 8) s_cnt[tid] = 0;
    __syncthreads();
    unsigned vals[5] = {tally.sdf, tally.col, tally.colsat, tally.probe, tally.carved};
#pragma unroll
    for (int k = 0; k < 5; k++) {
        unsigned v = vals[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((tid & 63) == 0 && v) atomicAdd(&s_cnt[k], v);  // LDS atomic, <= 4 per counter
    }
    __syncthreads();
    if (tid == 0) {
        unsigned long long *row = M.block_counters + (size_t)blockIdx.x * 16;
        for (int k = 0; k < 5; k++) row[k] += s_cnt[k];
        row[6] += n_new;
        row[7] += n_updated;
        if (blockIdx.x == 0) {
            row[5] += (unsigned long long)n_items;
            row[8] += 1ull;
        }
    }
}

// sums the per-block rows into counters[] (CHISEL_HIP_NUM_COUNTERS = 9 entries); one block of 256 threads
__global__ void reduce_counters_kernel(MapView M, int n_rows) {
    __shared__ unsigned long long s[256];
    for (int k = 0; k < 9; k++) {
        unsigned long long v = 0;
        for (int r = threadIdx.x; r < n_rows; r += 256) v += M.block_counters[(size_t)r * 16 + k];
        s[threadIdx.x] = v;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) M.counters[k] = s[0];
        __syncthreads();
    }
}

}  // namespace chisel_hip
