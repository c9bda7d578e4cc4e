// kernels_integrate.h -- projective SDF / weight / colour integration over the work-list.
//
// Replaces ProjectionIntegrator::Integrate<float> (ProjectionIntegrator.h:51-99) and
// ::IntegrateColor<float,uint8_t> (:101-183) plus the allocate-everything / erase-untouched protocol of
// Chisel::IntegrateDepthScan[Color] (Chisel.h:77-108, 133-143, 202-207):
//   - one workgroup per work-list chunk (persistent grid, grid-stride over the device-resident list,
//     no host round trip);
//   - a lane owns quads of 4 consecutive x voxels, so every voxel-plane access of a wave is one
//     contiguous 1 KiB segment (float4 per lane); sdf/weight/colour are read only for quads whose
//     camera-z interval can meet the depth band or the carve region of the pixels under the chunk
//     (bounds from the cull kernel; the reads are issued before the tile is staged so both overlap)
//     and written only for quads that did change;
//   - the pixel records (depth, truncation distance: built once per frame by depth_pyramid_kernel) under
//     the chunk are staged in LDS (the chunk's conservative pixel box from the cull kernel) and
//     gathered from there; boxes too large for the tile buffer fall back to gathers from global
//     memory (near-camera chunks);
//   - a launch applies up to KMAX frames: for chunks of <= 4096 voxels the voxel state stays in
//     registers from its first use to the end of the batch (read once, written once, frames applied in
//     order per voxel -- DistVoxel::Integrate is order dependent); 32^3 chunks are streamed slab by
//     slab per frame;
//   - a chunk that is not resident is allocated (thread 0 pops a pool slot and inserts the id into the
//     hash) only once some voxel of it is integrated: the outcome of the reference's "create,
//     integrate, erase if untouched" without ever allocating the ~98 % of candidates that stay
//     untouched.  Free slots hold default voxels, so nothing else needs writing;
//   - per-voxel arithmetic follows the reference operation by operation in fp32 (compiled with
//     -ffp-contract=off, IEEE divide), 3-term sums in Eigen's a0 + (a1 + a2) order.
#pragma once
#include "chisel_device.h"

namespace chisel_hip {

#ifndef INTEGRATE_MIN_WAVES
#define INTEGRATE_MIN_WAVES 4  // waves per SIMD the register allocator must leave room for: two 512-thread workgroups per CU
#endif
#ifndef INTEGRATE_TILE
#define INTEGRATE_TILE 4096    // pixel records per LDS tile buffer (two buffers of 8 bytes per record)
#endif
#ifndef INTEGRATE_QG
#define INTEGRATE_QG 1         // quads of a thread whose pixel-record fetches are in flight together (2: more ILP, more registers)
#endif
#ifndef INTEGRATE_QPT
#define INTEGRATE_QPT 2        // quads (of 4 voxels) per thread; 1024 / QPT threads per 16^3 chunk
#endif

struct Tally {
    unsigned sdf, col, colsat, probe, carved;
#ifdef CHISEL_STAMPS
    unsigned long long cyc[6];  // shader cycles of wave 0 in the frame loop: scalars + z tests, tile wait + barrier, apply, hand-shake, item prologue; [5] = iterations
    unsigned lanes[4];          // lanes that entered: the projection of a quad, the band update, the colour update, the carve test
    float wave64[4];            // 64 / (lanes of the wave that entered together): sums to 64 per wave-level execution
#endif
};
#ifdef CHISEL_STAMPS
#define PHASE_BEGIN() do { } while (0)
#define PHASE_END(i) do { } while (0)
#define REGION(i) do { tally.lanes[i] += 1u; tally.wave64[i] += 64.0f / (float)__popcll(__ballot(1)); } while (0)
#else
#define REGION(i) do { } while (0)
#define PHASE_BEGIN() do { } while (0)
#define PHASE_END(i) do { } while (0)
#endif

struct TileCtx {
    const PixelRec *rec;   // the frame's full record image (global)
    int u0, v0, tw, th;    // origin / size of the box staged in LDS; tw == 0: nothing staged
    // camera-z bounds from the cull kernel (conservative): a voxel can be in band only if z_near < z < z_far and
    // can take the carve test only if z < z_carve
    float z_near, z_far, z_carve;
    bool fastz;            // every voxel's camera z is in the range of reciprocal_in_range() (WI_FASTZ)
};

template <int N>
struct Geom {
    static constexpr int V = N * N * N;
    static constexpr int QX = N / 4;                                   // quads per x-row
    static constexpr int QUADS = V / 4;
    static constexpr int LAYER_QUADS = QX * N;                         // quads per z-layer
    static constexpr int SLAB_QUADS = (N == 8) ? 128 : 1024;           // quads a workgroup holds at a time
    static constexpr int QPT = (N == 8 && INTEGRATE_QPT > 2) ? 2 : INTEGRATE_QPT;  // quads per thread: same x, y, different z
    static constexpr int BLOCK = SLAB_QUADS / QPT;
    static constexpr int PASSES = QUADS / SLAB_QUADS;                  // 1 (8^3, 16^3) or 8 (32^3)
    // record tile in LDS per workgroup (16 / 32 KiB).  Larger boxes belong to near-camera chunks whose voxels map to
    // distinct pixels: staging the whole box would move more bytes than gathering the records from L2 directly.
    static constexpr int TILE_PIXELS = (N == 8) ? 2048 : INTEGRATE_TILE;
#ifndef INTEGRATE_GRID
#define INTEGRATE_GRID 1024
#endif
    static constexpr int GRID = (N == 8) ? 4096 : ((BLOCK > 512) ? 512 : INTEGRATE_GRID);  // persistent grid: >= what is resident at once
    static constexpr int MIN_WAVES = (N == 8) ? 1 : INTEGRATE_MIN_WAVES;
    static_assert(BLOCK % LAYER_QUADS == 0, "a thread's quads must share x and y");
    static_assert(GRID <= INTEGRATE_MAX_GRID, "per-workgroup counter rows");
};

// colour-camera pixel of a voxel centre given in world coordinates (ProjectionIntegrator.h:146-149); -1 = off the image
__device__ inline int color_pixel(const CameraParams &K, float vx, float vy, float vz) {
    const float dx = vx - K.t[0], dy = vy - K.t[1], dz = vz - K.t[2];
    const float qx = K.R[0] * dx + (K.R[3] * dy + K.R[6] * dz);
    const float qy = K.R[1] * dx + (K.R[4] * dy + K.R[7] * dz);
    const float qz = K.R[2] * dx + (K.R[5] * dy + K.R[8] * dz);
    const float iq = 1.0f / qz;
    const float cu = K.fx * qx * iq + K.cx;
    const float cv = K.fy * qy * iq + K.cy;
    if ((cu >= 0.0f) && (cv >= 0.0f) && (cu < (float)K.W) && (cv < (float)K.H)) return (int)cv * K.W + (int)cu;
    return -1;
}

// voxel state a thread holds in registers: QPT quads at the same (x, y) in different z-layers
template <int QPT>
struct ThreadState {
    float wx[4], wy;          // world coordinates of the voxel centres shared by the thread's quads
    float wz[QPT];
    float4 s4[QPT], w4[QPT];  // sdf / weight of the quads, kept as the 16-byte tuples the loads and stores move
    uint4 c4[QPT];            // packed RGBW
    unsigned have, havec;     // bit p: sdf/weight (colour) of quad p hold the chunk's values (read, or defaults of a new chunk)
    unsigned dchg, cchg;      // bit p: sdf/weight (colour) of quad p differ from memory
};

__device__ inline float &f4(float4 &v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }
__device__ inline unsigned &u4(uint4 &v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }

// voxelCenter = centroids[i] + origin (ChunkManager.cpp:61: Vec3(x,y,z)*res + half; ProjectionIntegrator.h:63)
template <int N>
__device__ inline void thread_place(const IntegratorParams &ip, float ox, float oy, float oz, int q0, ThreadState<Geom<N>::QPT> &S) {
    using G = Geom<N>;
    const int xq = q0 % G::QX, y = (q0 / G::QX) % N, z0 = q0 / G::LAYER_QUADS;
    S.wy = ((float)y * ip.res + ip.half_res) + oy;
#pragma unroll
    for (int j = 0; j < 4; j++) S.wx[j] = ((float)(xq * 4 + j) * ip.res + ip.half_res) + ox;
#pragma unroll
    for (int p = 0; p < G::QPT; p++) S.wz[p] = ((float)(z0 + p * (G::BLOCK / G::LAYER_QUADS)) * ip.res + ip.half_res) + oz;
}

// default voxels: DistVoxel() DistVoxel.cpp:27-31, ColorVoxel() ColorVoxel.cpp:27-31
template <int QPT>
__device__ inline void thread_defaults(ThreadState<QPT> &S, bool existed) {
#pragma unroll
    for (int p = 0; p < QPT; p++) {
        S.s4[p] = make_float4(99999.0f, 99999.0f, 99999.0f, 99999.0f);
        S.w4[p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        S.c4[p] = make_uint4(0u, 0u, 0u, 0u);
    }
    S.have = S.havec = existed ? 0u : ~0u;  // a chunk without a slot has default voxels: nothing to read
    S.dchg = S.cchg = 0u;
}

template <int N, bool COLOR>
__device__ inline void thread_store(const ThreadState<Geom<N>::QPT> &S, float *sdf_base, float *wgt_base, uchar4 *col_base, int q0) {
    using G = Geom<N>;
#pragma unroll
    for (int p = 0; p < G::QPT; p++) {
        const int q = q0 + p * G::BLOCK;
        if (S.dchg & (1u << p)) {
            *reinterpret_cast<float4 *>(sdf_base + 4 * q) = S.s4[p];
            *reinterpret_cast<float4 *>(wgt_base + 4 * q) = S.w4[p];
        }
        if (COLOR && (S.cchg & (1u << p))) *reinterpret_cast<uint4 *>(col_base + 4 * q) = S.c4[p];
    }
}

// Step 1 of a frame, before the tile is staged: camera z of the thread's voxels (three additions, no
// division) against the depth interval of the pixels under the chunk.  Sets bit p of `need` when quad p can be
// touched by this frame and issues the reads of its state if the registers do not hold it yet, so that the
// voxel traffic overlaps the staging of the tile.
template <int N, bool COLOR>
__device__ inline unsigned prefetch_frame(const FrameCam &F, const TileCtx &T, ThreadState<Geom<N>::QPT> &S, const float *sdf_base,
                                          const float *wgt_base, const uchar4 *col_base, int q0) {
    using G = Geom<N>;
    const CameraParams &C = F.cam;
    const float ay2 = C.R[5] * (S.wy - C.t[1]);
    float ax2[4];
#pragma unroll
    for (int j = 0; j < 4; j++) ax2[j] = C.R[2] * (S.wx[j] - C.t[0]);
    unsigned need = 0;
#pragma unroll
    for (int p = 0; p < G::QPT; p++) {
        const float s2 = ay2 + C.R[8] * (S.wz[p] - C.t[2]);
        const float z0 = ax2[0] + s2, z1 = ax2[1] + s2, z2 = ax2[2] + s2, z3 = ax2[3] + s2;
        const float zlo = fminf(fminf(z0, z1), fminf(z2, z3));
        const float zhi = fmaxf(fmaxf(z0, z1), fmaxf(z2, z3));
        const bool may_band = (zhi > T.z_near) & (zlo < T.z_far);
        const bool may_carve = zlo < T.z_carve;
        if (may_band | may_carve) {
            need |= 1u << p;
            const int q = q0 + p * G::BLOCK;
            if (!(S.have & (1u << p))) {
                S.s4[p] = *reinterpret_cast<const float4 *>(sdf_base + 4 * q);  // straight into the state tuple: no use, no wait
                S.w4[p] = *reinterpret_cast<const float4 *>(wgt_base + 4 * q);
                S.have |= 1u << p;
            }
            if (COLOR && may_band && !(S.havec & (1u << p))) {
                S.c4[p] = *reinterpret_cast<const uint4 *>(col_base + 4 * q);
                S.havec |= 1u << p;
            }
        }
    }
    return need;
}

// Step 2 of a frame, after the tile is staged: apply the frame to the quads of `need`.  `resident`: the
// reference's map holds the chunk before this frame (only the probe counter depends on it).  Returns bit 0:
// some voxel integrated (in band), bit 1: something changed (the reference's `updated`).
//
// Per quad the four voxels are classified branch-free so that their dependency chains (IEEE reciprocal ->
// pixel -> LDS record -> band tests) overlap; the update is predicated per quad, not per voxel.
template <int N, bool COLOR, bool SAMECAM>
__device__ inline int apply_frame(const IntegratorParams &ip, const FrameCam &F, const TileCtx &T, const PixelRec *s_tile,
                                  unsigned need, bool resident, ThreadState<Geom<N>::QPT> &S, Tally &tally) {
    using G = Geom<N>;
    const CameraParams &C = F.cam;
    // inCamera = R^T * (voxelCenter - t) (ProjectionIntegrator.h:64), row i of R^T summed as a0 + (a1 + a2);
    // the products that do not depend on z are shared by the thread's quads
    const float dy = S.wy - C.t[1];
    const float ay0 = C.R[3] * dy, ay1 = C.R[4] * dy, ay2 = C.R[5] * dy;
    float ax0[4], ax1[4], ax2[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float dx = S.wx[j] - C.t[0];
        ax0[j] = C.R[0] * dx;
        ax1[j] = C.R[1] * dx;
        ax2[j] = C.R[2] * dx;
    }
    int ret = 0;
    constexpr int QG = (G::QPT < INTEGRATE_QG) ? G::QPT : INTEGRATE_QG;  // quads whose record fetches are in flight together
#pragma unroll
    for (int g = 0; g < G::QPT; g += QG) {
        if (!((need >> g) & ((1u << QG) - 1u))) continue;
        bool on[QG][4], band[QG][4], carve[QG][4];
        int iu[QG][4], iv[QG][4], tidx[QG][4];
        float pcz[QG][4];
        bool any_out = false;
        PixelRec r[QG][4];
        REGION(0);
        PHASE_BEGIN();
        // ---- phase A: geometry + projection -> pixel of every voxel of the group ------------------------------
#pragma unroll
        for (int e = 0; e < QG; e++) {
            const int p = g + e;
            const bool wanted = (need >> p) & 1u;
            const float dz = S.wz[p] - C.t[2];
            const float s0 = ay0 + C.R[6] * dz;
            const float s1 = ay1 + C.R[7] * dz;
            const float s2 = ay2 + C.R[8] * dz;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float pcx = ax0[j] + s0, pcy = ax1[j] + s1;
                pcz[e][j] = ax2[j] + s2;
                // PinholeCamera::ProjectPoint (PinholeCamera.cpp:38-45)
                const float invZ = T.fastz ? reciprocal_in_range(pcz[e][j]) : 1.0f / pcz[e][j];
                const float u = C.fx * pcx * invZ + C.cx;
                const float v = C.fy * pcy * invZ + C.cy;
                // IsPointOnImage (PinholeCamera.cpp:61-64): 0 <= u < W && 0 <= v < H, and not z < 0 (ProjectionIntegrator.h:68 /
                // :126).  z == +-0 or NaN gives u, v = +-inf / NaN, which fail the image test, so "z > 0" is the same
                // predicate; for u not NaN, floor(u) in [0, W) <=> 0 <= u < W, and there floor(u) == (int)u (:72 / :131).
                iu[e][j] = (int)floorf(u);
                iv[e][j] = (int)floorf(v);
                on[e][j] = wanted & (pcz[e][j] > 0.0f) & ((unsigned)iu[e][j] < (unsigned)C.W) & ((unsigned)iv[e][j] < (unsigned)C.H) &
                           (u == u) & (v == v);
                const int tu = iu[e][j] - T.u0, tv = iv[e][j] - T.v0;
                const bool in_tile = ((unsigned)tu < (unsigned)T.tw) & ((unsigned)tv < (unsigned)T.th);
                tidx[e][j] = in_tile ? (int)__umul24((unsigned)tv, (unsigned)T.tw) + tu : 0;  // both below 2^12
                any_out |= on[e][j] & !in_tile;
            }
        }
        PHASE_END(0);
        // ---- phase B: records of all voxels of the group, in flight together -----------------------------------
#pragma unroll
        for (int e = 0; e < QG; e++)
#pragma unroll
#ifdef CHISEL_ABLATE_LDS
            for (int j = 0; j < 4; j++) r[e][j] = make_float2(pcz[e][j] + 0.01f, 0.05f);  // diagnostic: no gather
#else
            for (int j = 0; j < 4; j++) r[e][j] = s_tile[tidx[e][j]];
#endif
        // pixels outside the staged box (box too large for LDS: near-camera chunks; otherwise never, the box is conservative)
        if (any_out) {
#pragma unroll
            for (int e = 0; e < QG; e++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int tu = iu[e][j] - T.u0, tv = iv[e][j] - T.v0;
                    const bool in_tile = ((unsigned)tu < (unsigned)T.tw) & ((unsigned)tv < (unsigned)T.th);
                    const int gi = (on[e][j] & !in_tile) ? iv[e][j] * C.W + iu[e][j] : 0;  // DepthAt(row, col) DepthImage.h:72-76
                    const PixelRec gr = T.rec[gi];                                          // unconditional: all loads in flight
                    if (on[e][j] & !in_tile) r[e][j] = gr;
                }
        }
        PHASE_END(1);
        // ---- phase C: band tests and updates, quad by quad ------------------------------------------------------
#pragma unroll
        for (int e = 0; e < QG; e++) {
            const int p = g + e;
            float sd[4];
            bool any_band = false, any_carve = false;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // r.x is NaN for the pixels the reference skips (:74 depth > 50 / :134 isnan / :141 depth > 100): both tests fail
                sd[j] = r[e][j].x - pcz[e][j];                                                                   // surfaceDist :79 / :139
                band[e][j] = on[e][j] & (fabsf(sd[j]) < r[e][j].y + ip.diag);                                    // :81 / :144
                carve[e][j] = on[e][j] & !band[e][j] & (ip.carving != 0) & (sd[j] > r[e][j].y + ip.carving_dist);  // :86 / :164
                any_band |= band[e][j];
                any_carve |= carve[e][j];
                tally.sdf += band[e][j];
                tally.probe += carve[e][j] & resident;
            }
            PHASE_END(2);
            if (any_band) {
                REGION(1);
                // colour first: the pixels of the in-band voxels whose colour weight is below 8 are requested now (one 4-byte
                // gather per voxel, all in flight) and consumed after the sdf arithmetic
                bool fresh[4];
                unsigned cw[4], csh[4];
                int cpix[4];
                bool any_fresh = false;
                const bool word_gather = COLOR && F.color_channels >= 3;  // wave-uniform
                if (COLOR) {
                    const unsigned image_bytes = (unsigned)(F.ccam.W * F.ccam.H * F.color_channels);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        cpix[j] = SAMECAM ? (iv[e][j] * C.W + iu[e][j]) : (band[e][j] ? color_pixel(F.ccam, S.wx[j], S.wy, S.wz[p]) : -1);
                        const bool has = band[e][j] & (cpix[j] >= 0);
                        fresh[j] = has & ((u4(S.c4[p], j) >> 24) < 8u);  // colorVoxel.GetWeight() < 8, ProjectionIntegrator.h:152
                        if (!SAMECAM) tally.colsat += has & !fresh[j];  // one camera: every in-band voxel has a colour pixel, colsat = sdf - col
                        tally.col += fresh[j];
                        any_fresh |= fresh[j];
                    }
                    if (word_gather && any_fresh) {
#pragma unroll
                        for (int j = 0; j < 4; j++) cw[j] = color_gather(F.color, fresh[j] ? cpix[j] : 0, F.color_channels, image_bytes, csh[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float wu = 1.0f;                                         // Integrate: voxel.Integrate(surfaceDist, 1.0f) :84
#ifdef CHISEL_ABLATE_DIV
                    if (COLOR) wu = ip.weight * __builtin_amdgcn_rcpf(5 * r[e][j].y);
#else
                    if (COLOR) wu = constant_weight(ip.weight, r[e][j].y);  // IntegrateColor: weighter->GetWeight(.., truncation) :161-162
#endif
                    float ns = f4(S.s4[p], j), nw = f4(S.w4[p], j);
#ifdef CHISEL_ABLATE_DIV
                    ns = (nw * ns + wu * sd[j]) * __builtin_amdgcn_rcpf(wu + nw);
                    nw = nw + wu;
#else
                    dist_integrate(ns, nw, sd[j], wu);
#endif
                    f4(S.s4[p], j) = band[e][j] ? ns : f4(S.s4[p], j);
                    f4(S.w4[p], j) = band[e][j] ? nw : f4(S.w4[p], j);
                }
                S.dchg |= 1u << p;
                ret |= 3;
                PHASE_END(3);
                if (COLOR && any_fresh) {
                    REGION(2);
                    if (word_gather) {
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const unsigned nc = color_integrate_fresh(u4(S.c4[p], j), color_word(cw[j], csh[j]));
                            u4(S.c4[p], j) = fresh[j] ? nc : u4(S.c4[p], j);
                        }
                    } else {  // 1 / 2 channel images
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            if (fresh[j]) {
                                unsigned cbits = u4(S.c4[p], j);
                                uchar4 cv = *reinterpret_cast<uchar4 *>(&cbits);
                                uint8_t cr, cg, cb;
                                color_at(F.color, cpix[j], F.color_channels, cr, cg, cb);
                                cv = color_integrate(cv, cr, cg, cb, 1);
                                u4(S.c4[p], j) = *reinterpret_cast<unsigned *>(&cv);
                            }
                        }
                    }
                    S.cchg |= 1u << p;
                }
            }
            PHASE_END(4);
            if (any_carve) {
                REGION(3);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const bool hit = carve[e][j] && (f4(S.w4[p], j) > 0.0f) && sdf_below_carve_threshold(f4(S.s4[p], j));
                    tally.carved += hit;
                    if (hit) {
                        if (COLOR && !(f4(S.w4[p], j) < 5.0f)) {  // :166-177: decay
                            f4(S.w4[p], j) = f4(S.w4[p], j) - 1.0f;
                        } else {                                 // :88-95 / :170 Carve() == Reset()
                            f4(S.s4[p], j) = 99999.0f;
                            f4(S.w4[p], j) = 0.0f;
                        }
                        S.dchg |= 1u << p;
                        ret |= 2;
                    }
                }
            }
            PHASE_END(5);
        }
    }
    return ret;
}

// ChunkManager::CreateChunk (ChunkManager.cpp:171-174) on the device; one thread.  Returns the slot or -1.
// Takes the MapView from device memory so that its hash / free-list pointers do not occupy scalar registers
// in the integration loop.
__device__ __attribute__((noinline)) int create_chunk(const MapView *__restrict__ Mc, int x, int y, int z) {
    const MapView M = *Mc;
    int s = -1;
    const int top = atomicSub(M.free_top, 1) - 1;
    if (top < 0) {
        atomicAdd(M.free_top, 1);
        raise_error(M.error_flag, 1);
        return -1;
    }
    s = M.free_list[top];
    const uint64_t key = pack_id(x, y, z);
    const uint64_t h = chunk_hash(x, y, z) & M.hash_mask;
    for (uint64_t i = 0; i <= M.hash_mask; i++) {
        const uint64_t idx = (h + i) & M.hash_mask;
        const uint64_t cur = M.hash_keys[idx];
        if (cur == KEY_EMPTY || cur == KEY_TOMB) {
            if (atomicCAS((unsigned long long *)&M.hash_keys[idx], (unsigned long long)cur, (unsigned long long)key) == cur) {
                M.hash_vals[idx] = s;
                M.slot_key[s] = key;
                return s;
            }
        }
    }
    raise_error(M.error_flag, 2);
    return -1;
}

// ChunkManager::HasChunk on the device for a work item marked SLOT_LOOKUP; one thread.  Returns the slot or -1.
__device__ __attribute__((noinline)) int find_chunk(const MapView *__restrict__ Mc, int x, int y, int z) {
    const MapView M = *Mc;
    const uint64_t key = pack_id(x, y, z);
    const uint64_t h = chunk_hash(x, y, z) & M.hash_mask;
    for (uint64_t i = 0; i <= M.hash_mask; i++) {
        const uint64_t idx = (h + i) & M.hash_mask;
        const uint64_t k = M.hash_keys[idx];
        if (k == key) return M.hash_vals[idx];
        if (k == KEY_EMPTY) break;
    }
    return -1;
}

// Stage the pixel records of box (u0, v0, tw x th) of `rec` (row stride W) into the LDS tile with LDS-DMA
// (global_load_lds_dwordx4: 16 bytes = two records per lane, no register staging, asynchronous: the transfer of the
// next frame's tile runs under the current frame's arithmetic).  u0 and tw are even (cull kernel), so a record pair
// never straddles a row and every source address is 16-byte aligned.  magic = ceil(2^32 / (tw / 2)).
// Completion: the issuing wave's vmcnt, then a workgroup barrier before other waves read the tile.
template <int BLOCK>
__device__ inline void issue_tile_dma(PixelRec *lds_tile, const PixelRec *__restrict__ rec, int W, int u0, int v0, int tw, int npx,
                                      unsigned magic, int tid) {
#ifdef CHISEL_ABLATE_DMA
    return;
#endif
    const int npairs = npx >> 1, tw2 = tw >> 1;
    const int lane = tid & 63;
    const PixelRec *src0 = rec + (size_t)v0 * W + u0;
    for (int base = (tid & ~63); base < npairs; base += BLOCK) {  // wave-uniform trip count
        const int pp = base + lane;
        if (pp < npairs) {
            const int row = (int)__umulhi((unsigned)pp, magic);
            const PixelRec *src = src0 + (size_t)row * W + 2 * (pp - row * tw2);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(lds_tile + 2 * base), 16, 0, 0);
        }
    }
}

template <int N, bool COLOR, bool SAMECAM>
__global__ __launch_bounds__(Geom<N>::BLOCK, Geom<N>::MIN_WAVES) void integrate_kernel(IntegrateParams P, MapView M,
                                                                                         const MapView *__restrict__ Mc,
                                                                                         const WorkItem *__restrict__ items,
                                                                                         const FrameBox *__restrict__ boxes,
                                                                                         const int *__restrict__ work_count,
                                                                                         int max_items) {
    using G = Geom<N>;
    __shared__ __attribute__((aligned(16))) PixelRec s_tiles[2][G::TILE_PIXELS];  // double buffer: the next frame's tile lands while this one is used
    __shared__ int s_flags[4];  // [2 * parity + 0]: a voxel was integrated this frame, [+1]: something changed this frame
    __shared__ int s_slot;
    __shared__ unsigned s_changed[2];  // resident chunks: bit k = frame k changed some voxel (gathered once per item); the two
                                       // words alternate per item: the next item's word is cleared one item ahead, because
                                       // free-running waves need not meet a barrier between an item's start and this gather
    const int tid = threadIdx.x;
#ifdef CHISEL_STAMPS
#define STAMP(i) do { if (tid == 0 && M.stamps) M.stamps[(size_t)blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
    STAMP(0);
#ifdef CHISEL_STAMPS
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime();
#endif
    int n_items = *work_count;
    if (n_items > max_items) n_items = max_items;
    STAMP(1);
    if ((int)blockIdx.x >= n_items && blockIdx.x != 0) return;  // nothing to do, nothing to count (block 0 counts the frames)
    Tally tally = {};
    unsigned n_new = 0, n_updated = 0;
    const IntegratorParams &ip = P.ip;
    if (tid < 2) s_changed[tid] = 0u;
    int changed_word = 0;  // which of s_changed the current free-running item uses (block-uniform)
    __syncthreads();

    for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
        const WorkItem wi = items[it];
        if (it == (int)blockIdx.x) STAMP(2);
        const int cxi = __builtin_amdgcn_readfirstlane(wi.x), cyi = __builtin_amdgcn_readfirstlane(wi.y),
                  czi = __builtin_amdgcn_readfirstlane(wi.z);
        int slot = __builtin_amdgcn_readfirstlane(wi.slot);
        unsigned mask = (unsigned)__builtin_amdgcn_readfirstlane((int)wi.frame_mask);
        const int box_row = __builtin_amdgcn_readfirstlane(wi.box);
        if (slot == SLOT_LOOKUP) {
            // the previous batch may have created this chunk while the work-list was built: it has finished now
            __syncthreads();  // s_slot of the previous item consumed
            if (tid == 0) s_slot = find_chunk(Mc, cxi, cyi, czi);
            __syncthreads();
            slot = s_slot;
            if (slot < 0) {
                // not resident: frames before the first one that may integrate could only carve, i.e. do nothing
                const unsigned inband = (unsigned)__builtin_amdgcn_readfirstlane((int)wi.inband_mask);
                if (inband == 0u) continue;
                mask &= ~((1u << __builtin_ctz(inband)) - 1u);
            }
        }
        // Chunk origin (Chunk.cpp:43): numVoxels * ID (int) * resolution
        const float ox = (float)(N * cxi) * ip.res, oy = (float)(N * cyi) * ip.res, oz = (float)(N * czi) * ip.res;
        const bool existed = slot >= 0;  // memory of `slot` holds this chunk's voxels
        bool resident = existed;         // the reference's map contains the chunk before the current frame
        bool updated_any = false;
#ifdef CHISEL_STAMPS
        unsigned long long lt = __builtin_amdgcn_s_memtime();
#define LOOPT(i) do { const unsigned long long ln = __builtin_amdgcn_s_memtime(); tally.cyc[i] += ln - lt; lt = ln; } while (0)
#else
#define LOOPT(i) do { } while (0)
#endif
        __syncthreads();  // previous item's tile / flags / s_slot fully consumed
        if (tid < 4) s_flags[tid] = 0;
        // A chunk that already exists is "resident" for every frame, so its waves need no per-frame agreement: they
        // run through the batch independently (a barrier only where a staged tile must become visible) and report the
        // frames that changed something once, at the end.  New chunks keep the per-frame hand-shake below.
        const bool free_running = existed && (G::PASSES == 1);
        unsigned lane_changed = 0u;

        ThreadState<G::QPT> S;
        thread_defaults(S, existed);
        if (G::PASSES == 1) thread_place<N>(ip, ox, oy, oz, tid, S);
        const size_t base0 = (size_t)(existed ? slot : 0) * G::V;

        // ---- frame loop, software pipelined: tile of frame k+1 in flight (LDS-DMA) while frame k is applied --------
        auto frame_ctx = [&](int k, TileCtx &T, unsigned &magic, int &flags) {
            const FrameBox fb = boxes[(size_t)box_row * P.n_frames + k];
            flags = __builtin_amdgcn_readfirstlane(fb.flags);
            T.rec = P.f[k].rec;
            T.u0 = __builtin_amdgcn_readfirstlane((int)fb.u0);
            T.v0 = __builtin_amdgcn_readfirstlane((int)fb.v0);
            T.tw = 0;
            T.th = 0;
            T.fastz = (flags & WI_FASTZ) != 0;
            T.z_near = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fb.z_near)));
            T.z_far = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fb.z_far)));
            T.z_carve = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fb.z_carve)));
            magic = (unsigned)__builtin_amdgcn_readfirstlane((int)fb.magic);
            if (flags & WI_TILE) {
                const int tw = __builtin_amdgcn_readfirstlane((int)fb.u1) - T.u0 + 1;
                const int th = __builtin_amdgcn_readfirstlane((int)fb.v1) - T.v0 + 1;
                if (tw * th <= G::TILE_PIXELS) {
                    T.tw = tw;
                    T.th = th;
                }
            }
        };
        int parity = 0;
        TileCtx T;
        unsigned magic;
        int flags;
        int k = __builtin_ctz(mask);
        mask &= mask - 1;
        frame_ctx(k, T, magic, flags);
        if (T.tw) issue_tile_dma<G::BLOCK>(s_tiles[0], T.rec, P.f[k].cam.W, T.u0, T.v0, T.tw, T.tw * T.th, magic, tid);
        LOOPT(4);
        while (true) {
            const FrameCam &F = P.f[k];
            PixelRec *s_tile = s_tiles[parity];
            // next frame's scalars are requested first: their latency hides under this frame's vector work
            const bool more = mask != 0;
            int k_next = 0, flags_next = 0;
            unsigned magic_next = 0;
            TileCtx T_next;
            if (more) {
                k_next = __builtin_ctz(mask);
                mask &= mask - 1;
                frame_ctx(k_next, T_next, magic_next, flags_next);
            }
            // ---- register-resident chunk: which quads can this frame touch?  Their state is requested now ------
            unsigned need = 0;
            if (G::PASSES == 1)
                need = prefetch_frame<N, COLOR>(F, T, S, M.sdf + base0, M.wgt + base0, COLOR ? (M.rgbw + base0) : nullptr, tid);
            if (it == (int)blockIdx.x) STAMP(16);
            LOOPT(0);
            // [B] needed when this frame's tile must become visible to every wave, and when the next frame's tile is about to
            // be moved into the other buffer: a wave that runs ahead must not overwrite the records a slower wave is still
            // reading for the previous frame (free-running waves meet nowhere else).
            if (T.tw || (more && T_next.tw) || !free_running) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the tile has landed (and its voxel reads)
#ifndef CHISEL_ABLATE_BARRIER
                __syncthreads();  // tile visible to every wave; the other buffer and the other parity's flags are free
#endif
            }
            LOOPT(1);
            if (it == (int)blockIdx.x) STAMP(17);
            if (it == (int)blockIdx.x) STAMP(3);
#ifdef CHISEL_STAMPS
            if (it == (int)blockIdx.x && tid == 0 && M.stamps) {
                M.stamps[(size_t)blockIdx.x * 32 + 14] = (unsigned long long)(T.tw * T.th);
                M.stamps[(size_t)blockIdx.x * 32 + 15] = (unsigned long long)flags | ((unsigned long long)__popc(need) << 40);
            }
#endif
            if (tid == 0) {
                s_flags[2 * (parity ^ 1)] = 0;
                s_flags[2 * (parity ^ 1) + 1] = 0;
            }
            // next frame: its tile starts to move now
            if (more && T_next.tw)
                issue_tile_dma<G::BLOCK>(s_tiles[parity ^ 1], T_next.rec, P.f[k_next].cam.W, T_next.u0, T_next.v0, T_next.tw,
                                         T_next.tw * T_next.th, magic_next, tid);
            int t_ret = 0;
            if (G::PASSES == 1) {
#ifdef CHISEL_ABLATE_APPLY
                t_ret = need ? 3 : 0;
#else
                t_ret = apply_frame<N, COLOR, SAMECAM>(ip, F, T, s_tile, need, resident, S, tally);
#endif
            } else {
                // streamed chunk (32^3): slab by slab, state re-read per frame; a chunk that does not exist yet is
                // created by the first slab that integrates a voxel
                for (int pass = 0; pass < G::PASSES; pass++) {
                    const int q0 = pass * G::SLAB_QUADS + tid;
                    const size_t base = (size_t)(slot >= 0 ? slot : 0) * G::V;
                    thread_defaults(S, slot >= 0);
                    thread_place<N>(ip, ox, oy, oz, q0, S);
                    need = prefetch_frame<N, COLOR>(F, T, S, M.sdf + base, M.wgt + base, COLOR ? (M.rgbw + base) : nullptr, q0);
                    const int p_ret = apply_frame<N, COLOR, SAMECAM>(ip, F, T, s_tile, need, resident, S, tally);
                    t_ret |= p_ret;
                    if (slot < 0) {  // block-uniform
                        if (__syncthreads_or(p_ret & 1)) {
                            if (tid == 0) s_slot = create_chunk(Mc, cxi, cyi, czi);
                            __syncthreads();
                            slot = s_slot;
                            if (slot >= 0) n_new += (tid == 0);
                        }
                    }
                    if (slot >= 0)
                        thread_store<N, COLOR>(S, M.sdf + (size_t)slot * G::V, M.wgt + (size_t)slot * G::V,
                                               COLOR ? (M.rgbw + (size_t)slot * G::V) : nullptr, q0);
                }
            }
            LOOPT(2);
#ifdef CHISEL_STAMPS
            tally.cyc[5] += 1;
#endif
            if (free_running) {
                lane_changed |= (t_ret & 2) ? (1u << k) : 0u;
            } else {
                if (t_ret & 1) s_flags[2 * parity] = 1;      // benign race: every writer stores 1
                if (t_ret & 2) s_flags[2 * parity + 1] = 1;
                __syncthreads();  // [C] flags complete; tile consumed
                const bool f_in = s_flags[2 * parity] != 0, f_ch = s_flags[2 * parity + 1] != 0;
                resident |= f_in;
                updated_any |= f_ch;
                n_updated += (tid == 0 && f_ch);  // "needsUpdate" of the chunk for this frame (Chisel.h:85 / :167)
            }
            LOOPT(3);
            if (it == (int)blockIdx.x) STAMP(4);
            parity ^= 1;
            if (!more) break;
            k = k_next;
            T = T_next;
            magic = magic_next;
            flags = flags_next;
        }

        if (free_running) {
            // frames that changed something, over all lanes of the workgroup
            unsigned wmask = 0u;
            for (int bit = 0; bit < P.n_frames; bit++)
                if (__any((int)((lane_changed >> bit) & 1u))) wmask |= 1u << bit;
            if ((tid & 63) == 0 && wmask) atomicOr(&s_changed[changed_word], wmask);
            __syncthreads();
            const unsigned word = s_changed[changed_word];
            changed_word ^= 1;
            if (tid == 0) s_changed[changed_word] = 0u;  // for the next free-running item; its readers finished before this item began
            updated_any = word != 0u;
            n_updated += (tid == 0) ? (unsigned)__popc(word) : 0u;  // "needsUpdate" per frame (Chisel.h:85 / :167)
        }
        if (G::PASSES == 1) {
            if (!existed) {
                if (!resident) continue;  // block-uniform: the reference creates and then erases this chunk in every frame
                if (tid == 0) s_slot = create_chunk(Mc, cxi, cyi, czi);
                __syncthreads();
                slot = s_slot;
                if (slot < 0) continue;
                n_new += (tid == 0);
            }
            thread_store<N, COLOR>(S, M.sdf + (size_t)slot * G::V, M.wgt + (size_t)slot * G::V,
                                   COLOR ? (M.rgbw + (size_t)slot * G::V) : nullptr, tid);
        }
        // mark the slot for the mesher (Chisel.h:175-189 marks the 27-neighbourhood on the host)
        if (tid == 0 && updated_any && slot >= 0) M.slot_dirty[slot] = 1;
        if (it == (int)blockIdx.x) STAMP(5);
    }

    // ---- counters: wave reduction, then one no-return atomic per wave and counter into this block's private row
    // (no same-address contention across blocks: those run at ~90 per microsecond on this part; rows are summed
    // lazily by reduce_counters_kernel).  Nothing waits for the atomics.
    unsigned long long *row = M.block_counters + (size_t)blockIdx.x * 16;
    unsigned vals[5] = {tally.sdf, tally.col, (COLOR && SAMECAM) ? (tally.sdf - tally.col) : tally.colsat, tally.probe, tally.carved};
#pragma unroll
    for (int k = 0; k < 5; k++) {
        unsigned v = vals[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((tid & 63) == 0 && v) atomicAdd(&row[k], (unsigned long long)v);
    }
    if (tid == 0) {
        if (n_new) atomicAdd(&row[6], (unsigned long long)n_new);
        if (n_updated) atomicAdd(&row[7], (unsigned long long)n_updated);
        if (blockIdx.x == 0) {
            atomicAdd(&row[5], (unsigned long long)n_items);
            atomicAdd(&row[8], (unsigned long long)P.n_frames);
        }
    }
    STAMP(6);
#ifdef CHISEL_STAMPS
    if (M.stamps) {
        for (int k = 0; k < 4; k++) {
            unsigned a = tally.lanes[k];
            float b = tally.wave64[k];
            for (int o = 32; o > 0; o >>= 1) {
                a += __shfl_down(a, o);
                b += __shfl_down(b, o);
            }
            if ((tid & 63) == 0) {
                atomicAdd(&M.stamps[(size_t)blockIdx.x * 32 + 18 + k], (unsigned long long)a);
                atomicAdd(&M.stamps[(size_t)blockIdx.x * 32 + 22 + k], (unsigned long long)(b + 0.5f));
            }
        }
    }
    if (tid == 0 && M.stamps) {
        M.stamps[(size_t)blockIdx.x * 32 + 7] = __builtin_amdgcn_s_memtime() - clk0;  // shader-clock cycles
        for (int k = 0; k < 6; k++) M.stamps[(size_t)blockIdx.x * 32 + 8 + k] = tally.cyc[k];
    }
#endif
#undef STAMP
#undef LOOPT
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
