// kernels_cull.h -- per-frame work-list construction on the GPU.
//
// Replaces the host loops of Chisel::IntegrateDepthScan[Color] (Chisel.h:62-76, 118-143) and
// ChunkManager::GetChunkIDsIntersecting (ChunkManager.cpp:182-212), which enumerate every chunk of
// the frustum's bounding box, heap-allocate each missing one and integrate all of them.
//
//   depth_pyramid_kernel : min/max of the valid depth over 4x4 .. 64x64 pixel blocks
//   cull_kernel          : one thread per chunk id of the reference's candidate range; keeps a chunk
//                          only if (a) the reference would enumerate it (same range + same plane test),
//                          (b) this shard owns it, (c) a conservative projection/depth-range test
//                          cannot rule out that one of its voxels is updated or carved;
//                          looks the survivors up in the chunk hash and compacts them into the
//                          work-list with a wave ballot + prefix popcount (one atomic per wave).
//
// Dropping a chunk is parity-safe only when no voxel of it can change: untouched new chunks are
// erased again by the reference (Chisel.h:202-207) and untouched resident voxels keep their value.
// Every bound below is therefore conservative (margins for fp32 rounding), never exact.
#pragma once
#include "chisel_device.h"

namespace chisel_hip {

__device__ inline bool depth_valid(float d, float max_depth) {
    // NaN never updates (Integrate: every comparison false; IntegrateColor: isnan skip :134); d > max_depth is
    // skipped (:74 / :141); +-inf cannot satisfy |sd| < t+diag nor sd > t+cd for any of the truncators.
    return (d == d) && !(d > max_depth) && (fabsf(d) <= 3.0e38f);
}

// grid: (ceil(W/64), ceil(H/64)), block 256: thread = one 4x4 pixel block of a 64x64 tile
__global__ __launch_bounds__(256) void depth_pyramid_kernel(const float *__restrict__ depth, int W, int H,
                                                             float max_depth, PyramidView pyr, int *work_count) {
    __shared__ float2 red[256];
    const int tid = threadIdx.x;
    const int bx = tid & 15, by = tid >> 4;
    const int px0 = blockIdx.x * 64 + bx * 4, py0 = blockIdx.y * 64 + by * 4;
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *work_count = 0;  // consumed by cull_kernel (next launch)
    float mn = INFINITY, mx = -INFINITY;
    if (px0 < W && py0 < H) {
        const bool vec = ((W & 3) == 0) && (px0 + 3 < W);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int py = py0 + r;
            if (py >= H) break;
            if (vec) {
                float4 d = *reinterpret_cast<const float4 *>(depth + (size_t)py * W + px0);
                float v[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int c = 0; c < 4; c++)
                    if (depth_valid(v[c], max_depth)) {
                        mn = fminf(mn, v[c]);
                        mx = fmaxf(mx, v[c]);
                    }
            } else {
                for (int c = 0; c < 4 && px0 + c < W; c++) {
                    float v = depth[(size_t)py * W + px0 + c];
                    if (depth_valid(v, max_depth)) {
                        mn = fminf(mn, v);
                        mx = fmaxf(mx, v);
                    }
                }
            }
        }
    }
    // level 2 texel of this thread
    {
        int tx = px0 >> 2, ty = py0 >> 2;
        if (tx < pyr.w[0] && ty < pyr.h[0]) pyr.data[pyr.off[0] + ty * pyr.w[0] + tx] = make_float2(mn, mx);
    }
    red[tid] = make_float2(mn, mx);
    __syncthreads();
    // levels 3..6: 8x8, 4x4, 2x2, 1x1 texels per tile
    int dim = 16;
#pragma unroll
    for (int l = 1; l < PYR_LEVELS; l++) {
        int nd = dim >> 1;
        float2 v = make_float2(INFINITY, -INFINITY);
        int ox = tid % nd, oy = tid / nd;
        if (tid < nd * nd) {
            float2 a = red[(2 * oy) * dim + 2 * ox], b = red[(2 * oy) * dim + 2 * ox + 1];
            float2 c = red[(2 * oy + 1) * dim + 2 * ox], d = red[(2 * oy + 1) * dim + 2 * ox + 1];
            v.x = fminf(fminf(a.x, b.x), fminf(c.x, d.x));
            v.y = fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y));
        }
        __syncthreads();
        if (tid < nd * nd) {
            red[oy * nd + ox] = v;
            int tx = blockIdx.x * nd + ox, ty = blockIdx.y * nd + oy;
            if (tx < pyr.w[l] && ty < pyr.h[l]) pyr.data[pyr.off[l] + ty * pyr.w[l] + tx] = v;
        }
        __syncthreads();
        dim = nd;
    }
}

// extrema of the truncation distance over readings in [d0, d1]: all three strategies are quadratics
// in the reading (Inverse: s*k*d^2, vertex 0; Quadratic: |q(d)|*s with q > 0 everywhere, vertex -b/2a),
// so the extrema sit at the end points or at the vertex.
__device__ inline void truncation_range(int kind, float param, float d0, float d1, float &tmin, float &tmax) {
    float a = truncation_distance(kind, param, d0);
    float b = truncation_distance(kind, param, d1);
    tmin = fminf(a, b);
    tmax = fmaxf(a, b);
    if (kind != 0) {
        float vert = (kind == 1) ? 0.0f : (-kLinTerm / (2.0f * kQuadTerm));
        if (d0 <= vert && vert <= d1) {
            float c = (kind == 1) ? 0.0f : truncation_distance(kind, param, vert);
            tmin = fminf(tmin, c);
            tmax = fmaxf(tmax, c);
        }
    }
    // rounding slack
    float slack = 1e-5f * fmaxf(fabsf(tmin), fabsf(tmax)) + 1e-7f;
    tmin -= slack;
    tmax += slack;
}

template <int N>
__global__ __launch_bounds__(256) void cull_kernel(FrameParams P, MapView M, PyramidView pyr, WorkItem *items,
                                                    int *work_count, int max_items) {
    __shared__ float2 s_global;  // min/max over the whole image
    __shared__ float2 s_red[256];
    // whole-image extrema from the coarsest level (a few hundred texels at most)
    {
        const int L = PYR_LEVELS - 1;
        float2 v = make_float2(INFINITY, -INFINITY);
        int n = pyr.w[L] * pyr.h[L];
        for (int i = threadIdx.x; i < n; i += 256) {
            float2 t = pyr.data[pyr.off[L] + i];
            v.x = fminf(v.x, t.x);
            v.y = fmaxf(v.y, t.y);
        }
        s_red[threadIdx.x] = v;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < s) {
                s_red[threadIdx.x].x = fminf(s_red[threadIdx.x].x, s_red[threadIdx.x + s].x);
                s_red[threadIdx.x].y = fmaxf(s_red[threadIdx.x].y, s_red[threadIdx.x + s].y);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) s_global = s_red[0];
        __syncthreads();
    }

    const int total = P.range_dim[0] * P.range_dim[1] * P.range_dim[2];
    const int gid = blockIdx.x * 256 + threadIdx.x;
    bool keep = false;
    WorkItem wi;
    if (gid < total) {
        // reference order: x outer, y, z inner (ChunkManager.cpp:195-199)
        int iz = gid % P.range_dim[2];
        int iy = (gid / P.range_dim[2]) % P.range_dim[1];
        int ix = gid / (P.range_dim[2] * P.range_dim[1]);
        const int cx = P.range_min[0] + ix, cy = P.range_min[1] + iy, cz = P.range_min[2] + iz;
        keep = chunk_owner(cx, cy, cz, P.n_shards, P.shard_block) == P.shard_rank;

        // chunk box exactly as the reference builds it (ChunkManager.cpp:201-203)
        const float bminx = (float)(cx * N) * P.res, bminy = (float)(cy * N) * P.res, bminz = (float)(cz * N) * P.res;
        const float ext = (float)N * P.res;
        const float bmaxx = bminx + ext, bmaxy = bminy + ext, bmaxz = bminz + ext;
        if (keep) {
            // Frustum::Intersects (Frustum.cpp:41-79): true as soon as ONE plane has the p-vertex on its positive side
            bool hit = false;
#pragma unroll
            for (int p = 0; p < 6; p++) {
                float nx = P.planes[4 * p], ny = P.planes[4 * p + 1], nz = P.planes[4 * p + 2], dd = P.planes[4 * p + 3];
                float vx = (nx < 0.0f) ? bminx : bmaxx;
                float vy = (ny < 0.0f) ? bminy : bmaxy;
                float vz = (nz < 0.0f) ? bminz : bmaxz;
                float dotv = __fadd_rn(__fmul_rn(vx, nx), __fadd_rn(__fmul_rn(vy, ny), __fmul_rn(vz, nz)));  // a0 + (a1 + a2)
                if (__fadd_rn(dotv, dd) > 0.0f) hit = true;
            }
            keep = hit;
        }
        if (keep) {
            // conservative camera-space bounds of the box (voxel centres lie strictly inside it)
            const CameraParams &C = P.cam;
            float zmin = INFINITY, zmax = -INFINITY, umin = INFINITY, umax = -INFINITY, vmin = INFINITY, vmax = -INFINITY;
            bool any_behind = false;
            const float zeps = 0.25f * P.res;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                float wx = ((k & 1) ? bmaxx : bminx) - C.t[0];
                float wy = ((k & 2) ? bmaxy : bminy) - C.t[1];
                float wz = ((k & 4) ? bmaxz : bminz) - C.t[2];
                float px = C.R[0] * wx + C.R[3] * wy + C.R[6] * wz;
                float py = C.R[1] * wx + C.R[4] * wy + C.R[7] * wz;
                float pz = C.R[2] * wx + C.R[5] * wy + C.R[8] * wz;
                zmin = fminf(zmin, pz);
                zmax = fmaxf(zmax, pz);
                if (pz < zeps) {
                    any_behind = true;
                } else {
                    float iz_ = 1.0f / pz;
                    float u = C.fx * px * iz_ + C.cx, v = C.fy * py * iz_ + C.cy;
                    umin = fminf(umin, u); umax = fmaxf(umax, u);
                    vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
                }
            }
            const float zslack = 1e-4f * (fabsf(zmin) + fabsf(zmax)) + 0.01f * P.res;
            zmin -= zslack;
            zmax += zslack;
            int u0 = 0, v0 = 0, u1 = C.W - 1, v1 = C.H - 1;
            bool tile = false;
            if (zmax < 0.0f) {
                keep = false;  // every voxel has z < 0 (ProjectionIntegrator.h:68)
            } else if (!any_behind) {
                // all corners in front: the projection of the box is inside the bbox of the projected corners
                float fu0 = floorf(umin) - 2.0f, fu1 = floorf(umax) + 2.0f, fv0 = floorf(vmin) - 2.0f, fv1 = floorf(vmax) + 2.0f;
                if (fu1 < 0.0f || fv1 < 0.0f || fu0 > (float)(C.W - 1) || fv0 > (float)(C.H - 1)) {
                    keep = false;  // projects entirely off the image (IsPointOnImage fails for every voxel)
                } else {
                    u0 = (int)fmaxf(fu0, 0.0f); v0 = (int)fmaxf(fv0, 0.0f);
                    u1 = (int)fminf(fu1, (float)(C.W - 1)); v1 = (int)fminf(fv1, (float)(C.H - 1));
                    tile = true;
                }
            }
            if (keep) {
                // depth extrema over the pixel box from the pyramid
                float dmin = INFINITY, dmax = -INFINITY;
                int l = -1;
#pragma unroll
                for (int k = 0; k < PYR_LEVELS; k++) {
                    int s = PYR_L0 + k;
                    if (l < 0 && ((u1 >> s) - (u0 >> s)) <= 2 && ((v1 >> s) - (v0 >> s)) <= 2) l = k;
                }
                if (l < 0) {
                    dmin = s_global.x;
                    dmax = s_global.y;
                } else {
                    int s = PYR_L0 + l;
                    for (int ty = (v0 >> s); ty <= (v1 >> s); ty++)
                        for (int tx = (u0 >> s); tx <= (u1 >> s); tx++) {
                            float2 t = pyr.data[pyr.off[l] + ty * pyr.w[l] + tx];
                            dmin = fminf(dmin, t.x);
                            dmax = fmaxf(dmax, t.y);
                        }
                }
                if (!(dmin <= dmax)) {
                    keep = false;  // no valid depth under the chunk
                } else {
                    float tmin, tmax;
                    truncation_range(P.trunc_kind, P.trunc_param, dmin, dmax, tmin, tmax);
                    const float zlo = fmaxf(zmin, 0.0f) - zslack;
                    const float band = tmax + P.diag;
                    // |d - z| < t + diag for some pixel/voxel pair  =>  dmin - band < zmax  and  dmax + band > zlo
                    bool inband = (dmin - band < zmax) && (dmax + band > zlo);
                    // d - z > t + carvingDist for some pair  =>  dmax - zlo > tmin + carvingDist
                    bool carve = P.carving && (dmax - zlo > tmin + P.carving_dist - 1e-6f);
                    // hash lookup
                    uint64_t key = pack_id(cx, cy, cz);
                    uint64_t h = chunk_hash(cx, cy, cz) & M.hash_mask;
                    int slot = -1;
                    for (uint64_t i = 0; i <= M.hash_mask; i++) {
                        uint64_t k = M.hash_keys[(h + i) & M.hash_mask];
                        if (k == key) {
                            slot = M.hash_vals[(h + i) & M.hash_mask];
                            break;
                        }
                        if (k == KEY_EMPTY) break;
                    }
                    int flags = (inband ? WI_INBAND : 0) | ((carve && slot >= 0) ? WI_CARVE : 0) | (tile ? WI_TILE : 0);
                    keep = (flags & (WI_INBAND | WI_CARVE)) != 0;
                    wi.x = cx; wi.y = cy; wi.z = cz;
                    wi.slot = slot;
                    wi.u0 = (short)u0; wi.v0 = (short)v0; wi.u1 = (short)u1; wi.v1 = (short)v1;
                    wi.flags = flags;
                    wi.pad = 0;
                }
            }
        }
    }
    // wave64 compaction: ballot + prefix popcount, one atomic per wave
    const unsigned long long mask = __ballot(keep);
    if (mask) {
        const int lane = threadIdx.x & 63;
        int base = 0;
        if (lane == (int)__builtin_ctzll(mask)) base = atomicAdd(work_count, __popcll(mask));
        base = __shfl(base, (int)__builtin_ctzll(mask));
        if (keep) {
            int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
            if (pos < max_items) items[pos] = wi;
        }
    }
}

}  // namespace chisel_hip
