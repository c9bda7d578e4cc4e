// kernels_cull.h -- per-batch work-list construction on the GPU.
//
// Replaces the host loops of Chisel::IntegrateDepthScan[Color] (Chisel.h:62-76, 118-143) and
// ChunkManager::GetChunkIDsIntersecting (ChunkManager.cpp:182-212), which enumerate every chunk of
// the frustum's bounding box, heap-allocate each missing one and integrate all of them.
//
//   depth_pyramid_kernel : per frame: pixel records (depth, truncation distance) and the min/max of
//                          the valid depth over 4x4 .. 64x64 pixel blocks
//   cull_kernel          : one thread per (chunk id of the union of the frames' candidate ranges, frame): keeps
//                          the pair only if (a) the reference would enumerate it
//                          (same range + same plane test), (b) this shard owns it, (c) a conservative
//                          projection/depth-range test cannot rule out that one of its voxels is
//                          updated or carved; compacts the surviving candidates with a wave ballot +
//                          prefix popcount (one atomic per wave).  Reads the frames only, never the map,
//                          so it runs (with the pyramid) on the auxiliary stream while the previous
//                          batch is still being integrated.
//                          Since round 6 the first wave of every block also looks its survivors up (ChunkManager::HasChunk),
//                          drops the candidates that could only carve a chunk that is not resident, and compacts the rest
//                          straight into the work-list (rounds 2-5: resolve_kernel / order_kernel).  In the pipelined form it runs
//                          while the previous batch is being integrated: the chunks that batch may still create are known (its
//                          work items without a slot, kept in a small "pending" set), and a candidate found there is passed on
//                          with slot = SLOT_LOOKUP for the integration kernel to look up itself.
//   brick_kernel         : one wave per work item: which frames can touch which 8 x 8 x 4-voxel brick of the chunk (what a wave of
//                          the integration kernel owns) -> a 16-bit frame mask per brick (rounds 4-5: refine_kernel, per cell)
//
// Dropping a (chunk, frame) pair is parity-safe only when no voxel of the chunk can change in that
// frame: untouched new chunks are erased again by the reference (Chisel.h:202-207) and untouched
// resident voxels keep their value.  Every bound below is therefore conservative (margins for fp32
// rounding), never exact.
#pragma once
#include "chisel_device.h"

namespace chisel_hip {

// per-batch device counters: candidates, work items, pending-set overflow flag, a constant 1, work items per cost class,
// placement cursors per cost class
constexpr int COUNT_CANDS = 0, COUNT_ITEMS = 1, COUNT_OVERFLOW = 2, COUNT_ONE = 3, COUNT_CLASS0 = 4, COUNT_CURSOR0 = 12, COUNT_QUEUE0 = 32, COUNT_INTS = COUNT_QUEUE0 + 128 * 32;  // the queue heads of the integration kernel (QUEUE_HEADS x QUEUE_STRIDE)

__device__ inline bool depth_valid(float d, float max_depth) {
    // NaN never updates (Integrate: every comparison false; IntegrateColor: isnan skip :134); d > max_depth is
    // skipped (:74 / :141); +-inf cannot satisfy |sd| < t+diag nor sd > t+cd for any of the truncators.
    return (d == d) && !(d > max_depth) && (fabsf(d) <= 3.0e38f);
}

// pixel record: what ProjectionIntegrator.h:72-79 / :131-141 derive from the depth pixel alone
__device__ inline PixelRec make_record(const IntegratorParams &ip, float d) {
    PixelRec r;
    r.y = truncation_distance(ip.trunc_kind, ip.trunc_param, d);
    r.x = (d > ip.max_depth) ? __builtin_nanf("") : d;  // skipped pixels: NaN fails both the band and the carve test
    return r;
}

// grid: (ceil(W/64), ceil(H/64), n_frames), block 256: thread = one 4x4 pixel block of a 64x64 tile
// Also resets the batch's counters (COUNT_* below) and empties its pending set.
#ifndef PYRAMID_NT
#define PYRAMID_NT 0
#endif
__global__ __launch_bounds__(256) void depth_pyramid_kernel(PyramidParams P, PyramidView pyr, int *counts, uint64_t *pending) {
    __shared__ float2 red[256];
    const int tid = threadIdx.x;
    const int k = blockIdx.z;
    const float *__restrict__ depth = P.depth[k];
    PixelRec *__restrict__ rec = P.rec + (size_t)k * P.rec_stride;
    float2 *__restrict__ pdata = pyr.data + (size_t)k * P.pyr_stride;
    const int W = P.W, H = P.H;
    const float max_depth = P.ip.max_depth;
    const int bx = tid & 15, by = tid >> 4;
    const int px0 = blockIdx.x * 64 + bx * 4, py0 = blockIdx.y * 64 + by * 4;
    {  // consumed by cull_kernel / brick_kernel (next launches)
        const unsigned gid = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 256u + tid;
        const unsigned n_threads = gridDim.x * gridDim.y * gridDim.z * 256u;
        for (unsigned i = gid; i < (unsigned)COUNT_INTS; i += n_threads)
            if (i != (unsigned)COUNT_ONE) counts[i] = 0;
        for (unsigned i = gid; i < PENDING_CAPACITY; i += n_threads) pending[i] = KEY_EMPTY;
        if (gid == 0) pending[PENDING_CAPACITY] = 0;  // the set's overflow flag travels with it (the sets rotate independently of the counters)
    }
    float mn = INFINITY, mx = -INFINITY;
    if (px0 < W && py0 < H) {
        const bool vec = ((W & 3) == 0) && (px0 + 3 < W);
        if (vec) {
            // the thread's rows requested two at a time (a row past the image reads the last one again and is dropped below): two round
            // trips for the tile instead of four -- the kernel is one generation of waves, as long as its chain of dependent accesses.
            // (All four at once take 39 registers: beside an integration kernel that leaves 32 per SIMD the kernel would wait for its drain.)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                float4 d4[2];
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const int py = py0 + 2 * h + r < H ? py0 + 2 * h + r : H - 1;
#if PYRAMID_NT
                    {
                        typedef float v4f __attribute__((ext_vector_type(4)));
                        const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(depth + (size_t)py * W + px0));
                        d4[r] = make_float4(t.x, t.y, t.z, t.w);
                    }
#else
                    d4[r] = *reinterpret_cast<const float4 *>(depth + (size_t)py * W + px0);
#endif
                }
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const int py = py0 + 2 * h + r;
                    if (py < H) {
                        const float v[4] = {d4[r].x, d4[r].y, d4[r].z, d4[r].w};
                        PixelRec o[4];
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            o[c] = make_record(P.ip, v[c]);
                            if (depth_valid(v[c], max_depth)) {
                                mn = fminf(mn, v[c]);
                                mx = fmaxf(mx, v[c]);
                            }
                        }
                        float4 *dst = reinterpret_cast<float4 *>(rec + (size_t)py * W + px0);
#if PYRAMID_NT
                        // streaming stores: the records of a batch that is built BESIDE the integration of the batch before it must not push
                        // that batch's records (what its voxels gather) out of the XCDs' L2s
                        typedef float v4f __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store((v4f){o[0].x, o[0].y, o[1].x, o[1].y}, reinterpret_cast<v4f *>(dst));
                        __builtin_nontemporal_store((v4f){o[2].x, o[2].y, o[3].x, o[3].y}, reinterpret_cast<v4f *>(dst) + 1);
#else
                        dst[0] = make_float4(o[0].x, o[0].y, o[1].x, o[1].y);
                        dst[1] = make_float4(o[2].x, o[2].y, o[3].x, o[3].y);
#endif
                    }
                }
            }
        } else {
            for (int r = 0; r < 4 && py0 + r < H; r++) {
                const int py = py0 + r;
                for (int c = 0; c < 4 && px0 + c < W; c++) {
                    float v = depth[(size_t)py * W + px0 + c];
                    rec[(size_t)py * W + px0 + c] = make_record(P.ip, v);
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
        if (tx < pyr.w[0] && ty < pyr.h[0]) pdata[pyr.off[0] + ty * pyr.w[0] + tx] = make_float2(mn, mx);
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
            if (tx < pyr.w[l] && ty < pyr.h[l]) pdata[pyr.off[l] + ty * pyr.w[l] + tx] = v;
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

// min / max of the valid depth over pixel box (u0, v0) .. (u1, v1): the finest pyramid level that covers the box with
// at most 3 x 3 texels, read as nine unconditional (clamped, possibly repeated) loads so that they are in flight together
// (T = texels per side, 3 for the chunks of the cull kernel; the bricks of brick_kernel take 4: a finer level, a tighter range)
template <int T = 3>
__device__ inline void pyramid_minmax(const PyramidView &pyr, const float2 *__restrict__ pdata, float2 whole, int u0, int v0, int u1,
                                      int v1, float &dmin, float &dmax) {
    // (the level's offset and width are carried along as selects between the view's scalars: an index into the view's arrays that differs per
    // lane would have to go through memory)
    int l = PYR_LEVELS - 1, loff = pyr.off[PYR_LEVELS - 1], lw = pyr.w[PYR_LEVELS - 1];
#pragma unroll
    for (int k = PYR_LEVELS - 2; k >= 0; k--) {
        const int s = PYR_L0 + k;
        if (((u1 >> s) - (u0 >> s)) <= T - 1 && ((v1 >> s) - (v0 >> s)) <= T - 1) {
            l = k;
            loff = pyr.off[k];
            lw = pyr.w[k];
        }
    }
    const int s = PYR_L0 + l;
    const float2 *lvl = pdata + loff;
    const int tx0 = u0 >> s, ty0 = v0 >> s, tx1 = u1 >> s, ty1 = v1 >> s;
    dmin = INFINITY;
    dmax = -INFINITY;
    if (tx1 - tx0 <= T - 1 && ty1 - ty0 <= T - 1) {
        float2 t[T * T];
#pragma unroll
        for (int i = 0; i < T * T; i++) t[i] = lvl[min(ty0 + i / T, ty1) * lw + min(tx0 + i % T, tx1)];
#pragma unroll
        for (int i = 0; i < T * T; i++) {
            dmin = fminf(dmin, t[i].x);
            dmax = fmaxf(dmax, t[i].y);
        }
    } else {  // box wider than T texels of the coarsest level (near-camera chunks): the extrema of the whole image
        dmin = whole.x;
        dmax = whole.y;
    }
}

// min / max of the valid depth over the whole image of one frame: a wave reduces the coarsest level (<= a few hundred texels)
__device__ inline float2 whole_image_minmax(const PyramidView &pyr, const float2 *__restrict__ pdata) {
    const int L = PYR_LEVELS - 1, n = pyr.w[L] * pyr.h[L];
    float mn = INFINITY, mx = -INFINITY;
    for (int i = (int)(threadIdx.x & 63); i < n; i += 64) {
        const float2 t = pdata[pyr.off[L] + i];
        mn = fminf(mn, t.x);
        mx = fmaxf(mx, t.y);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    return make_float2(mn, mx);
}

// The per-frame test of one chunk, in two steps so that a wave whose 64 chunks (one 4 x 4 x 4 block of ids: cull_kernel) all fail
// the cheap part retires after it -- empty space and everything outside the view is most of the enumerated range.
//   cull_pre : the reference's id range (ChunkManager.cpp:189-199), then the bounding sphere of the chunk's box against the
//              view: one camera-space point, approximate reciprocals, every bound widened accordingly -> the pixel box under
//              the sphere and its camera-z interval.  No memory access.
//   cull_post: depth range under that pixel box (pyramid) against the z interval; then the reference's plane test
//              (Frustum::Intersects, kept for fidelity: it never prunes) and the exact bounds from the eight corners.
// Both are conservative: a (chunk, frame) pair is dropped only when no voxel of the chunk can be updated or carved by the frame.
struct CullPre {
    int su0, sv0, su1, sv1;   // pixel box under the bounding sphere (the whole image when the sphere reaches behind the camera)
    float zs0, zs1, slack;    // camera-z interval of the sphere, rounding slack
};
// the geometric part for the axis-aligned box of N^3 voxels with index (cx, cy, cz) in units of N voxels: a chunk
template <int N>
__device__ inline bool box_pre(const IntegratorParams &ip, const CameraParams &C, int cx, int cy, int cz, CullPre &pre);
template <int N>
__device__ inline bool cull_pre(const IntegratorParams &ip, const CullFrame &F, int cx, int cy, int cz, CullPre &pre) {
    if ((unsigned)(cx - F.range_min[0]) >= (unsigned)F.range_dim[0] || (unsigned)(cy - F.range_min[1]) >= (unsigned)F.range_dim[1] ||
        (unsigned)(cz - F.range_min[2]) >= (unsigned)F.range_dim[2])
        return false;
    return box_pre<N>(ip, F.cam, cx, cy, cz, pre);
}
template <int N>
__device__ inline bool box_pre(const IntegratorParams &ip, const CameraParams &C, int cx, int cy, int cz, CullPre &pre) {
    const float bminx = (float)(cx * N) * ip.res, bminy = (float)(cy * N) * ip.res, bminz = (float)(cz * N) * ip.res;
    const float ext = (float)N * ip.res;
    const float hx = 0.5f * ext;
    const float rad = hx * 1.7320508f * 1.001f + 1e-6f;
    const float wx = (bminx + hx) - C.t[0], wy = (bminy + hx) - C.t[1], wz = (bminz + hx) - C.t[2];
    const float px = C.R[0] * wx + C.R[3] * wy + C.R[6] * wz;
    const float py = C.R[1] * wx + C.R[4] * wy + C.R[7] * wz;
    const float pz = C.R[2] * wx + C.R[5] * wy + C.R[8] * wz;
    const float slack = 1e-4f * (fabsf(px) + fabsf(py) + fabsf(pz) + rad);
    const float zs1 = pz + rad + slack;
    if (zs1 < 0.0f) return false;  // every voxel has z < 0 (ProjectionIntegrator.h:68)
    const float zs0 = pz - rad - slack;
    pre.su0 = 0; pre.sv0 = 0; pre.su1 = C.W - 1; pre.sv1 = C.H - 1;
    if (zs0 > 0.25f * ip.res) {
        // camera-space box [px +- rad] x [py +- rad] x [zs0, zs1] projects inside these bounds
        const float i0 = __builtin_amdgcn_rcpf(zs0) * 1.00001f, i1 = __builtin_amdgcn_rcpf(zs1) * 0.99999f;
        const float xl = px - rad - slack, xh = px + rad + slack, yl = py - rad - slack, yh = py + rad + slack;
        const float ul = C.fx * xl * (xl < 0.0f ? i0 : i1) + C.cx, uh = C.fx * xh * (xh < 0.0f ? i1 : i0) + C.cx;
        const float vl = C.fy * yl * (yl < 0.0f ? i0 : i1) + C.cy, vh = C.fy * yh * (yh < 0.0f ? i1 : i0) + C.cy;
        const float fu0 = floorf(ul) - 3.0f, fu1 = floorf(uh) + 3.0f, fv0 = floorf(vl) - 3.0f, fv1 = floorf(vh) + 3.0f;
        if (fu1 < 0.0f || fv1 < 0.0f || fu0 > (float)(C.W - 1) || fv0 > (float)(C.H - 1)) return false;  // off the image
        pre.su0 = (int)fmaxf(fu0, 0.0f); pre.sv0 = (int)fmaxf(fv0, 0.0f);
        pre.su1 = (int)fminf(fu1, (float)(C.W - 1)); pre.sv1 = (int)fminf(fv1, (float)(C.H - 1));
    }
    pre.zs0 = zs0; pre.zs1 = zs1; pre.slack = slack;
    return true;
}
// A BRICK of the integration kernel (kernels_integrate.h: a unit = one wave = 8 x 8 x 4 voxels at 4 voxels per lane): the same bound for
// the centres of its voxels -- a box of half-extents (3.5, 3.5, 1.5) voxels around their centre (vx0, vy0, vz0 = the brick's first voxel,
// counted from the world origin) --, per camera axis i the exact extent of the rotated box, sum_j |R_ji| h_j (a bounding sphere around the
// brick would be 5.1 voxels, this 1.5 to 3.5 per axis).
constexpr int BRICK_X = 8, BRICK_Y = 8, BRICK_Z = 4;
template <int N>
struct BrickGrid {
    static constexpr int NBX = N / BRICK_X, NBY = N / BRICK_Y, NBZ = N / BRICK_Z, PER_CHUNK = NBX * NBY * NBZ;  // 2 (8^3), 16 (16^3), 128 (32^3)
    static_assert(NBX >= 1 && NBY >= 1 && NBZ >= 1, "bricks tile the chunk");
};
__device__ inline bool brick_pre(const IntegratorParams &ip, const CameraParams &C, int vx0, int vy0, int vz0, CullPre &pre) {
    const float hx = 0.5f * (float)(BRICK_X - 1) * ip.res, hy = 0.5f * (float)(BRICK_Y - 1) * ip.res, hz = 0.5f * (float)(BRICK_Z - 1) * ip.res;
    // centre of the voxel centres: (v0 + (B - 1) / 2 + 0.5) res
    const float wx = ((float)vx0 * ip.res + (hx + ip.half_res)) - C.t[0], wy = ((float)vy0 * ip.res + (hy + ip.half_res)) - C.t[1],
                wz = ((float)vz0 * ip.res + (hz + ip.half_res)) - C.t[2];
    const float px = C.R[0] * wx + C.R[3] * wy + C.R[6] * wz;
    const float py = C.R[1] * wx + C.R[4] * wy + C.R[7] * wz;
    const float pz = C.R[2] * wx + C.R[5] * wy + C.R[8] * wz;
    // rounding: as cell_pre (1e-4 relative is three orders of magnitude more than the integration kernel's own coordinates carry)
    const float slack = 1e-4f * (fabsf(px) + fabsf(py) + fabsf(pz) + (hx + hy + hz)) + 1e-6f;
    const float ex = (fabsf(C.R[0]) * hx + fabsf(C.R[3]) * hy + fabsf(C.R[6]) * hz) * 1.001f + slack;
    const float ey = (fabsf(C.R[1]) * hx + fabsf(C.R[4]) * hy + fabsf(C.R[7]) * hz) * 1.001f + slack;
    const float ez = (fabsf(C.R[2]) * hx + fabsf(C.R[5]) * hy + fabsf(C.R[8]) * hz) * 1.001f + slack;
    const float zs1 = pz + ez;
    if (zs1 < 0.0f) return false;  // every voxel has z < 0 (ProjectionIntegrator.h:68)
    const float zs0 = pz - ez;
    pre.su0 = 0; pre.sv0 = 0; pre.su1 = C.W - 1; pre.sv1 = C.H - 1;
    if (zs0 > 0.25f * ip.res) {
        const float i0 = __builtin_amdgcn_rcpf(zs0) * 1.00001f, i1 = __builtin_amdgcn_rcpf(zs1) * 0.99999f;
        const float xl = px - ex, xh = px + ex, yl = py - ey, yh = py + ey;
        const float ul = C.fx * xl * (xl < 0.0f ? i0 : i1) + C.cx, uh = C.fx * xh * (xh < 0.0f ? i1 : i0) + C.cx;
        const float vl = C.fy * yl * (yl < 0.0f ? i0 : i1) + C.cy, vh = C.fy * yh * (yh < 0.0f ? i1 : i0) + C.cy;
        const float fu0 = floorf(ul) - 2.0f, fu1 = floorf(uh) + 2.0f, fv0 = floorf(vl) - 2.0f, fv1 = floorf(vh) + 2.0f;
        if (fu1 < 0.0f || fv1 < 0.0f || fu0 > (float)(C.W - 1) || fv0 > (float)(C.H - 1)) return false;  // off the image
        pre.su0 = (int)fmaxf(fu0, 0.0f); pre.sv0 = (int)fmaxf(fv0, 0.0f);
        pre.su1 = (int)fminf(fu1, (float)(C.W - 1)); pre.sv1 = (int)fminf(fv1, (float)(C.H - 1));
    }
    pre.zs0 = zs0; pre.zs1 = zs1; pre.slack = slack;
    return true;
}
// does pyramid_minmax fall back to the extrema of the whole image for this box?
__device__ inline bool box_needs_whole_image(int u0, int v0, int u1, int v1) {
    return ((u1 >> PYR_L1) - (u0 >> PYR_L1)) > 2 || ((v1 >> PYR_L1) - (v0 >> PYR_L1)) > 2;
}

template <int N>
__device__ inline int cull_post(const IntegratorParams &ip, const CullFrame &F, const PyramidView &pyr, const float2 *__restrict__ pdata,
                                float2 whole, int cx, int cy, int cz, const CullPre &pre, FrameBox &fb) {
    const CameraParams &C = F.cam;
    {
        float dmin, dmax;
        pyramid_minmax(pyr, pdata, whole, pre.su0, pre.sv0, pre.su1, pre.sv1, dmin, dmax);
        if (!(dmin <= dmax)) return 0;  // no valid depth under the chunk
        float tmin, tmax;
        truncation_range(ip.trunc_kind, ip.trunc_param, dmin, dmax, tmin, tmax);
        const float zlo = fmaxf(pre.zs0, 0.0f) - pre.slack;
        const float band = tmax + ip.diag;
        const bool inband = (dmin - band < pre.zs1) && (dmax + band > zlo);
        const bool carve = ip.carving && (dmax - zlo > tmin + ip.carving_dist - 1e-6f);
        if (!inband && !carve) return 0;
    }
    // chunk box exactly as the reference builds it (ChunkManager.cpp:201-203)
    const float bminx = (float)(cx * N) * ip.res, bminy = (float)(cy * N) * ip.res, bminz = (float)(cz * N) * ip.res;
    const float ext = (float)N * ip.res;
    const float bmaxx = bminx + ext, bmaxy = bminy + ext, bmaxz = bminz + ext;
    // Frustum::Intersects (Frustum.cpp:41-79): true as soon as ONE plane has the p-vertex on its positive side
    if (!ip.single_chunk) {
        bool hit = false;
#pragma unroll
        for (int p = 0; p < 6; p++) {
            float nx = F.planes[4 * p], ny = F.planes[4 * p + 1], nz = F.planes[4 * p + 2], dd = F.planes[4 * p + 3];
            float vx = (nx < 0.0f) ? bminx : bmaxx;
            float vy = (ny < 0.0f) ? bminy : bmaxy;
            float vz = (nz < 0.0f) ? bminz : bmaxz;
            float dotv = __fadd_rn(__fmul_rn(vx, nx), __fadd_rn(__fmul_rn(vy, ny), __fmul_rn(vz, nz)));  // a0 + (a1 + a2)
            if (__fadd_rn(dotv, dd) > 0.0f) hit = true;
        }
        if (!hit) return 0;
    }
    // ---- survivors: conservative camera-space bounds of the box itself (voxel centres lie strictly inside it)
    float zmin = INFINITY, zmax = -INFINITY, umin = INFINITY, umax = -INFINITY, vmin = INFINITY, vmax = -INFINITY;
    bool any_behind = false;
    const float zeps = 0.25f * ip.res;
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
    const float zslack = 1e-4f * (fabsf(zmin) + fabsf(zmax)) + 0.01f * ip.res;
    zmin -= zslack;
    zmax += zslack;
    int u0 = 0, v0 = 0, u1 = C.W - 1, v1 = C.H - 1;
    bool tile = false, box_inside = false;
    if (zmax < 0.0f) return 0;  // every voxel has z < 0 (ProjectionIntegrator.h:68)
    if (!any_behind) {
        // all corners in front: the projection of the box is inside the bbox of the projected corners
        float fu0 = floorf(umin) - 2.0f, fu1 = floorf(umax) + 2.0f, fv0 = floorf(vmin) - 2.0f, fv1 = floorf(vmax) + 2.0f;
        if (fu1 < 0.0f || fv1 < 0.0f || fu0 > (float)(C.W - 1) || fv0 > (float)(C.H - 1))
            return 0;  // projects entirely off the image (IsPointOnImage fails for every voxel)
        box_inside = fu0 >= 0.0f && fv0 >= 0.0f && fu1 <= (float)(C.W - 1) && fv1 <= (float)(C.H - 1);
        u0 = (int)fmaxf(fu0, 0.0f); v0 = (int)fmaxf(fv0, 0.0f);
        u1 = (int)fminf(fu1, (float)(C.W - 1)); v1 = (int)fminf(fv1, (float)(C.H - 1));
        tile = true;
    }
    // depth extrema over the pixel box from the pyramid
    float dmin, dmax;
    pyramid_minmax(pyr, pdata, whole, u0, v0, u1, v1, dmin, dmax);
    if (!(dmin <= dmax)) return 0;  // no valid depth under the chunk
    float tmin, tmax;
    truncation_range(ip.trunc_kind, ip.trunc_param, dmin, dmax, tmin, tmax);
    const float zlo = fmaxf(zmin, 0.0f) - zslack;
    const float band = tmax + ip.diag;
    // |d - z| < t + diag for some pixel/voxel pair  =>  dmin - band < zmax  and  dmax + band > zlo
    const bool inband = (dmin - band < zmax) && (dmax + band > zlo);
    // d - z > t + carvingDist for some pair  =>  dmax - zlo > tmin + carvingDist
    const bool carve = ip.carving && (dmax - zlo > tmin + ip.carving_dist - 1e-6f);
    // (`tile`: all corners in front and an even image width -- kept as a flag of the verdict, WI_TILE)
    if (C.W & 1) tile = false;
    // all corners at least a quarter voxel in front of the camera: every voxel centre's camera z lies between the corner
    // extrema (widened by the slack above), so the short reciprocal of the projection is exact for this chunk.  The
    // kernel's own z carries the rounding of three products of magnitude <= mag: keep well clear of it.
    const float mag = fabsf(C.t[0]) + fabsf(C.t[1]) + fabsf(C.t[2]) + fabsf(bminx) + fabsf(bminy) + fabsf(bminz) + 3.0f * ext;
    const bool fastz = !any_behind && (zmin >= FASTZ_MIN + 1e-5f * mag) && (zmax <= FASTZ_MAX);
    // the in-band update's weight 1 / (5 * truncation) by the same short reciprocal: every valid pixel under the chunk has its
    // truncation distance in [tmin, tmax] (conservative, see truncation_range), a factor of two inside the checked range
    const bool fastwu = ip.weight == 1.0f && (5.0f * tmin >= 2.0f * FASTZ_MIN) && (5.0f * tmax <= 0.5f * FASTZ_MAX);
    // Every voxel on the image: the box of the projected corners keeps two pixels from every border (above), and the integration
    // kernel's own u, v differ from the exact projection by a small fraction of a pixel: its camera coordinates carry the rounding of
    // three products of magnitude <= mag (<= 4 ulp: e = 5e-7 mag); u = fx x / z + cx turns an error e of x and of z into at most
    // fx e (1 + |x| / z) / z, and on the image |x| / z <= (W + |cx|) / fx -- so z >= 1e-5 (fx + W + |cx|) mag bounds it by 0.05 pixels
    // (the three roundings of u itself add a few ulp of a value below 2^15).
    const float span = fmaxf(fabsf(C.fx), fabsf(C.fy)) + (float)(C.W + C.H) + fabsf(C.cx) + fabsf(C.cy);
    const bool inside = box_inside && fastz && (zmin >= 1e-5f * span * mag);
    return (inband ? WI_INBAND : 0) | (carve ? WI_CARVE : 0) | (tile ? WI_TILE : 0) | (fastz ? WI_FASTZ : 0) | (fastwu ? WI_FASTWU : 0) |
           (inside ? WI_INSIDE : 0);
}

#ifndef REFINE_TEXELS
#define REFINE_TEXELS 4
#endif

// Which frames of the launch can touch which BRICK of every work item -> brick_masks[item][brick] (16 bits: the frames), in work-list order.
// The chunk-level test of the cull kernel looks at the depth range under the whole chunk's pixel box (40-60 pixels wide at 1 cm / 2 m): its
// bounds let 40-45 % more voxel-frames into the integration kernel than take the band or the carve branch.  A brick's box is half of that on
// a side and the depth range under it tight, and a brick is exactly what one wave of the integration kernel owns: its mask is what that wave
// walks.  (Rounds 4-5: refine_kernel, one wave per (work item, frame) pair, lane = one of 64 cells, a 64-bit cell mask per pair -- four
// times the tests, a wave per pair; 14 us on the driver's window.)  Here ONE WAVE PER WORK ITEM: lane = (brick, frame group), the frames of
// a group one after the other (16^3 chunks: 16 bricks x 4 groups, 16 frames in four rounds; a round is one round trip to the pyramid), the
// groups' verdicts merged by shuffles.  A brick is needed by frame k if one of its voxels may integrate, or may take the carve test while
// the chunk is resident: resident now, or (conservatively, as the cull kernel has it) created by an earlier frame of the launch; SLOT_LOOKUP
// items count as resident (the integration kernel drops what it must once it knows).  `full` (test hook): every brick takes its item's mask.
#ifndef BRICK_BLOCK
#define BRICK_BLOCK 256   // threads per workgroup of brick_kernel at most (its waves never meet; the host picks 64 or this per launch)
#endif
template <int N>
__global__ __launch_bounds__(256) void brick_kernel(IntegrateParams P, PyramidView pyr, int pyr_stride, const WorkItem *__restrict__ items,
                                                    const int *__restrict__ work_count, int max_items, unsigned short *__restrict__ brick_masks, int full) {
    constexpr int BPC = BrickGrid<N>::PER_CHUNK;
    constexpr int BR = BPC > 64 ? BPC / 64 : 1;   // bricks per lane (32^3 chunks: two)
    constexpr int G = BPC >= 64 ? 1 : 64 / BPC;   // frame groups: lane = group * BPC + brick
    const IntegratorParams &ip = P.ip;
    const int n_frames = P.n_frames;
    const int lane = threadIdx.x & 63;
    const int g = BPC >= 64 ? 0 : lane / BPC;
    int n_items = *work_count;
    if (n_items > max_items) n_items = max_items;
    const int waves = (int)gridDim.x * (int)(blockDim.x >> 6);
    // launches of more than eight frames give an item two waves, frames 0-7 and 8-15 (a wave's chain is its rounds: at most two then); each
    // writes its byte of the bricks' 16-bit masks
    const int halves = n_frames > 8 ? 2 : 1;
    for (int p = (int)blockIdx.x * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6); p < n_items * halves; p += waves) {
        const int it = halves == 2 ? p >> 1 : p, hh = halves == 2 ? (p & 1) : 0;
        const int k_lo = 8 * hh, k_hi = halves == 2 ? (hh ? n_frames : 8) : n_frames;  // this wave's frames
        // (the first round's camera is requested beside the work item, the next round's while this one's texels are under way: the
        // cameras come from the argument segment with a per-lane index, i.e. through the vector memory path)
        CameraParams Cn = P.f[min(k_lo + g, n_frames - 1)].cam;
        const WorkItem wi = items[it];
        const unsigned fmask = (unsigned)__builtin_amdgcn_readfirstlane((int)wi.frame_mask);
        const int slot = __builtin_amdgcn_readfirstlane(wi.slot);
        const unsigned inband_all = (unsigned)__builtin_amdgcn_readfirstlane((int)wi.inband_mask);
        const int cxi = __builtin_amdgcn_readfirstlane(wi.x), cyi = __builtin_amdgcn_readfirstlane(wi.y), czi = __builtin_amdgcn_readfirstlane(wi.z);
        unsigned m[BR];
#pragma unroll
        for (int h = 0; h < BR; h++) m[h] = 0u;
        if (full) {
#pragma unroll
            for (int h = 0; h < BR; h++) m[h] = fmask;
        } else {
            for (int j = 0; k_lo + j * G < k_hi; j++) {  // (a round; lanes of one group share the frame, a group whose frame is not in the mask idles)
                const int kf = k_lo + g + j * G;
                const CameraParams C = Cn;
                if (k_lo + (j + 1) * G < k_hi) Cn = P.f[min(kf + G, n_frames - 1)].cam;
                if (kf >= k_hi || !((fmask >> kf) & 1u)) continue;
                const bool resident = slot >= 0 || slot == SLOT_LOOKUP || (inband_all & ((1u << kf) - 1u)) != 0u;
#pragma unroll
                for (int h = 0; h < BR; h++) {
                    const int b = (BPC >= 64 ? lane : lane % BPC) + 64 * h;
                    const int bx = b % BrickGrid<N>::NBX, by = (b / BrickGrid<N>::NBX) % BrickGrid<N>::NBY, bz = b / (BrickGrid<N>::NBX * BrickGrid<N>::NBY);
                    CullPre pre;
                    bool need = false;
                    if (brick_pre(ip, C, N * cxi + bx * BRICK_X, N * cyi + by * BRICK_Y, N * czi + bz * BRICK_Z, pre)) {
                        if (box_needs_whole_image(pre.su0, pre.sv0, pre.su1, pre.sv1)) {
                            need = true;  // (a brick next to the camera: not worth the reduction over the image)
                        } else {
                            float dmin, dmax;
                            pyramid_minmax<REFINE_TEXELS>(pyr, pyr.data + (size_t)kf * pyr_stride, make_float2(INFINITY, -INFINITY), pre.su0, pre.sv0, pre.su1, pre.sv1, dmin, dmax);
                            if (dmin <= dmax) {
                                float tmin, tmax;
                                truncation_range(ip.trunc_kind, ip.trunc_param, dmin, dmax, tmin, tmax);
                                const float zlo = fmaxf(pre.zs0, 0.0f) - pre.slack;
                                const float band = tmax + ip.diag;
                                const bool inb = (dmin - band < pre.zs1) && (dmax + band > zlo);
                                const bool crv = ip.carving && (dmax - zlo > tmin + ip.carving_dist - 1e-6f);
                                need = inb || (crv && resident);
                            }
                        }
                    }
                    m[h] |= need ? (1u << kf) : 0u;
                }
            }
            // the groups' verdicts on one brick meet in the group-0 lane
#pragma unroll
            for (int o = 32; o >= BPC && o > 0; o >>= 1) {
#pragma unroll
                for (int h = 0; h < BR; h++) m[h] |= (unsigned)__shfl_xor((int)m[h], o);
            }
        }
        if (BPC >= 64 || lane < BPC) {
#pragma unroll
            for (int h = 0; h < BR; h++) {
                if (halves == 2) reinterpret_cast<unsigned char *>(brick_masks)[2 * ((size_t)it * BPC + lane + 64 * h) + hh] = (unsigned char)((m[h] >> (8 * hh)) & 0xffu);
                else brick_masks[(size_t)it * BPC + lane + 64 * h] = (unsigned short)m[h];
            }
        }
    }
}

__device__ inline bool pending_contains(const uint64_t *__restrict__ set, uint64_t key, uint64_t h) {
    for (unsigned i = 0; i < PENDING_CAPACITY; i++) {
        const uint64_t k = set[(h + i) & (PENDING_CAPACITY - 1)];
        if (k == key) return true;
        if (k == KEY_EMPTY) return false;
    }
    return false;
}
__device__ inline bool pending_insert(uint64_t *set, uint64_t key, uint64_t h) {
    for (unsigned i = 0; i < PENDING_CAPACITY / 2; i++) {  // give up on a crowded table: the caller raises the overflow flag
        unsigned long long *p = (unsigned long long *)&set[(h + i) & (PENDING_CAPACITY - 1)];
        const unsigned long long cur = atomicCAS(p, (unsigned long long)KEY_EMPTY, (unsigned long long)key);
        if (cur == KEY_EMPTY || cur == key) return true;
    }
    return false;
}

// The candidate ids one shard has to judge.  Unsharded: every id of the union range (the reference walks it x outer, z inner,
// ChunkManager.cpp:195-199; the order of the candidates has no bearing on the voxel fields, and the work-list is cost-ordered
// anyway), here in blocks of 4 x 4 x 4 ids per wave.  Sharded: only the ids this shard owns -- chunk_owner() is (bx + 3 by + 5 bz) mod n on
// super-blocks of b^3 chunks, so for each (by, bz) the owned bx are one residue class: slot j of (by, bz) is the j-th owned bx
// at or after the range's first super-block.  (Before, every shard walked the whole range with n - 1 of n lanes idle.)
constexpr int CULL_BLOCK = 4;
#ifndef CULL_EARLY_EXIT
#define CULL_EARLY_EXIT 1
#endif
// CULL_EARLY_EXIT lets waves of cull_kernel leave in front of the workgroup's barriers.  That is sound on GCN / CDNA, where s_barrier
// counts the waves of the workgroup that are still alive and a terminated wave's LDS writes have retired -- a property of this hardware
// family, not of the HIP programming model.  This library is built for gfx950 only; any other target must either be vetted for the same
// behaviour or build with -DCULL_EARLY_EXIT=0 (the exiting waves then sit through the barriers).  The forced-shape tests
// (CHISEL_HIP_CULL_WAVES, tests/test_gpu_parity.py::test_integration_schedules[cull4 / cull16]) are the guard on the GPU.
#if CULL_EARLY_EXIT && defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "CULL_EARLY_EXIT relies on CDNA barrier semantics (s_barrier counts surviving waves only): build this target with -DCULL_EARLY_EXIT=0"
#endif
struct CullSpace {
    int sharded;
    int b;                 // super-block edge in chunks
    int n;                 // shards
    int rank;
    int sb0[3], nsb[3];    // first super-block of the range and super-blocks per axis
    int per_row;           // owned super-blocks per (by, bz): ceil(nsb[0] / n)
    int total;             // candidate slots (some fall outside the range and are skipped)
    __host__ __device__ explicit CullSpace(const CullParams &P) {
        sharded = P.ip.n_shards > 1;
        b = P.ip.shard_block > 0 ? P.ip.shard_block : 1;
        n = P.ip.n_shards;
        rank = P.ip.shard_rank;
        if (!sharded) {
            // blocks of CULL_BLOCK^3 ids (64 = the lanes of a wave: spatially compact, so that most waves see nothing but empty space
            // or nothing but space outside the view and retire early); nsb = blocks per axis, cells beyond the range are skipped
            for (int a = 0; a < 3; a++) {
                sb0[a] = 0;
                nsb[a] = (P.range_dim[a] + CULL_BLOCK - 1) / CULL_BLOCK;
            }
            total = nsb[0] * nsb[1] * nsb[2] * CULL_BLOCK * CULL_BLOCK * CULL_BLOCK;
            per_row = 0;
            return;
        }
        for (int a = 0; a < 3; a++) {
            sb0[a] = floor_div(P.range_min[a], b);
            nsb[a] = floor_div(P.range_min[a] + P.range_dim[a] - 1, b) - sb0[a] + 1;
        }
        per_row = (nsb[0] + n - 1) / n;
        total = per_row * nsb[1] * nsb[2] * b * b * b;
    }
    // slot c -> chunk id; false: nothing to judge in this slot
    __host__ __device__ bool id(const CullParams &P, int c, int &cx, int &cy, int &cz) const {
        if (c >= total) return false;
        if (!sharded) {
            constexpr int B3 = CULL_BLOCK * CULL_BLOCK * CULL_BLOCK;
            const int cell = c % B3, blk = c / B3;
            const int bz = blk % nsb[2], by = (blk / nsb[2]) % nsb[1], bx = blk / (nsb[2] * nsb[1]);
            const int ix = bx * CULL_BLOCK + cell % CULL_BLOCK, iy = by * CULL_BLOCK + (cell / CULL_BLOCK) % CULL_BLOCK,
                      iz = bz * CULL_BLOCK + cell / (CULL_BLOCK * CULL_BLOCK);
            cx = P.range_min[0] + ix; cy = P.range_min[1] + iy; cz = P.range_min[2] + iz;
            return ix < P.range_dim[0] && iy < P.range_dim[1] && iz < P.range_dim[2];
        }
        const int cell = c % (b * b * b), s = c / (b * b * b);
        const int ibz = s % nsb[2], iby = (s / nsb[2]) % nsb[1], j = s / (nsb[2] * nsb[1]);
        const int by = sb0[1] + iby, bz = sb0[2] + ibz;
        int r = (rank - 3 * by - 5 * bz - sb0[0]) % n;  // first owned bx at or after sb0[0]: bx = sb0[0] + r (mod n)
        if (r < 0) r += n;
        const int bx = sb0[0] + r + j * n;
        if (bx >= sb0[0] + nsb[0]) return false;
        cx = bx * b + cell % b; cy = by * b + (cell / b) % b; cz = bz * b + cell / (b * b);
        return cx >= P.range_min[0] && cx < P.range_min[0] + P.range_dim[0] && cy >= P.range_min[1] && cy < P.range_min[1] + P.range_dim[1] &&
               cz >= P.range_min[2] && cz < P.range_min[2] + P.range_dim[2];
    }
};

// One wave per frame of the batch over the same 64 chunk ids (block = 64 * KL threads, KL = frames rounded up to a
// power of two): every per-frame constant is wave-uniform (scalar loads), the per-frame verdicts meet in LDS.
// for its survivors -- hash lookup, frame mask, compaction straight into the work-list -- and one launch set is three
// kernels on one stream instead of five on two.
// WV = waves per workgroup at most; a launch of more frames gives each wave several (k, k + WV, ...).  One wave per frame (WV = 16) is
// the shortest chain -- and a workgroup of sixteen waves, which needs sixteen free slots on ONE CU: beside an integration kernel whose
// single-wave workgroups take every slot as it frees up it can wait for hundreds of microseconds (4 agents: every second batch of
// the stream stood still for 85 us behind such a cull kernel, rocprofv3 timeline).  Four waves find room in what a 6-waves-per-SIMD
// integration kernel leaves free.  The host picks 4 for launches whose frames look at different parts of the space (their common id
// range is much larger than any frame's own: most (block, frame) pairs die in the range test anyway), 16 otherwise.
template <int KL, int WV>
struct CullGeom {
    static constexpr int WAVES = KL < WV ? KL : WV;
    static constexpr int FPW = KL / WAVES;  // frames per wave
};
template <int N, int KL, int WV>
__global__ __launch_bounds__((64 * CullGeom<KL, WV>::WAVES)) void cull_kernel(CullParams P, PyramidView pyr, WorkItem *items, FrameBox *boxes, int *counts,
                                                        int max_items, MapView M, const uint64_t *__restrict__ prev_pending,
                                                        const uint64_t *__restrict__ prev2_pending, const int *__restrict__ force_uncertain,
                                                        uint64_t *my_pending, ItemSync *sync, int contig, unsigned short *brick_masks) {
    constexpr int WAVES = CullGeom<KL, WV>::WAVES, FPW = CullGeom<KL, WV>::FPW;
    // which frames a wave takes when it takes several: k, k + WAVES, ... or (contig) k * FPW, k * FPW + 1, ...  The host picks the one
    // that gives a wave frames looking at DIFFERENT parts of the space: of a wave's frames few then survive the range test for any one
    // block of ids, and the survivors of a block are spread over its waves instead of queueing up in one of them (interleaved agents,
    // frame i from agent i mod 4: strided hands wave k the four frames of agent k -- four cull_post in a row in one wave, three waves idle)

    int *item_count = counts + COUNT_ITEMS;
    __shared__ int s_flags[KL][64];
    __shared__ int s_pos[64];
    const int lane = threadIdx.x & 63;
    const int k = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // this wave: frames k, k + WAVES, ...
    int cx = 0, cy = 0, cz = 0;
    bool have_id;
    if (P.ip.n_shards > 1) {
        const CullSpace space(P);
        have_id = space.id(P, (int)blockIdx.x * 64 + lane, cx, cy, cz);
    } else {
        // unsharded: a three-dimensional grid of 4 x 4 x 4 blocks, lane = cell (CullSpace::id without its integer divisions)
        const int ix = (int)blockIdx.z * CULL_BLOCK + (lane & 3), iy = (int)blockIdx.y * CULL_BLOCK + ((lane >> 2) & 3),
                  iz = (int)blockIdx.x * CULL_BLOCK + (lane >> 4);
        cx = P.range_min[0] + ix; cy = P.range_min[1] + iy; cz = P.range_min[2] + iz;
        have_id = ix < P.range_dim[0] && iy < P.range_dim[1] && iz < P.range_dim[2];
    }
    const bool mine = have_id && chunk_owner(cx, cy, cz, P.ip.n_shards, P.ip.shard_block) == P.ip.shard_rank;
    constexpr int UNROLL = FPW <= 4 ? FPW : 1;
    bool any_flag = false;
    // (the verdicts live in LDS: a wave of the one-wave form takes all KL frames, one after the other -- no unrolling, no register arrays)
#pragma unroll UNROLL
    for (int j = 0; j < FPW; j++) {
        const int kf = (FPW > 1 && contig) ? k * FPW + j : k + j * WAVES;
        FrameBox fb;
        fb.flags = 0;
        int fl = 0;
        CullPre pre;
        bool alive = false;
        if (mine && kf < P.n_frames) alive = cull_pre<N>(P.ip, P.f[kf], cx, cy, cz, pre);
        if (__any(alive)) {  // wave-uniform: most (wave, frame) pairs stop here
            // the extrema of the whole image (a wave-wide reduction of the coarsest pyramid level) only where a box is wider than three
            // of its texels: chunks next to the camera
            float2 whole = make_float2(INFINITY, -INFINITY);
            if (__any(alive && box_needs_whole_image(pre.su0, pre.sv0, pre.su1, pre.sv1))) whole = whole_image_minmax(pyr, pyr.data + (size_t)kf * P.pyr_stride);
            if (alive) fl = cull_post<N>(P.ip, P.f[kf], pyr, pyr.data + (size_t)kf * P.pyr_stride, whole, cx, cy, cz, pre, fb);
        }
        s_flags[kf][lane] = fl;
        any_flag = any_flag || fl != 0;
    }
#if CULL_EARLY_EXIT
    // A wave none of whose frames can touch any of its 64 chunks is done: its flags are in LDS, it owns no row of `boxes` (the
    // integration kernel reads a box only for the frames of an item's mask), and wave 0 does the merge.  Leaving NOW instead of sitting
    // through the two barriers below gives its slot back while the block's live waves -- with agents looking in different directions a
    // quarter of them -- are still in cull_post (a barrier waits for the surviving waves of a workgroup only; a terminated wave's LDS
    // write has retired).
    if (k != 0 && !__any(any_flag)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        return;
    }
#endif
    __syncthreads();
    if (k == 0) {
        // ---- merge the frames of each chunk; look the survivors up; compact them into the work-list ---------------------------------
        unsigned inband = 0, carve = 0;
#pragma unroll
        for (int j = 0; j < KL; j++) {
            const int f = s_flags[j][lane];
            inband |= (f & WI_INBAND) ? (1u << j) : 0u;
            carve |= (f & WI_CARVE) ? (1u << j) : 0u;
        }
        bool keep = (inband | carve) != 0u;
        int slot = -1;
        unsigned mask = 0u;
        if (__any(keep)) {  // wave-uniform: most blocks of ids hold nothing
            // a pending set of the batches in flight is incomplete (its overflow flag sits behind its last bucket), or the test hook
            const bool forced = force_uncertain && *force_uncertain != 0;  // (test hook: everything through the look-up path of the integration kernel)
            const bool all_uncertain = forced || (prev_pending && prev_pending[PENDING_CAPACITY] != 0) || (prev2_pending && prev2_pending[PENDING_CAPACITY] != 0);
            if (keep) {
                // hash lookup (ChunkManager::HasChunk ChunkManager.h:79-82)
                const uint64_t key = pack_id(cx, cy, cz);
                const uint64_t h0 = chunk_hash(cx, cy, cz), h = h0 & M.hash_mask;
                // (the home bucket's value is requested with its key: most probes end there, and the wave waits once instead of twice)
                {
                    const uint64_t k0 = M.hash_keys[h];
                    const int v0 = M.hash_vals[h];
                    if (k0 == key) {
                        slot = v0;
                    } else if (k0 != KEY_EMPTY) {
                        for (uint64_t i = 1; i <= M.hash_mask; i++) {
                            const uint64_t kk = M.hash_keys[(h + i) & M.hash_mask];
                            if (kk == key) {
                                slot = M.hash_vals[(h + i) & M.hash_mask];
                                break;
                            }
                            if (kk == KEY_EMPTY) break;
                        }
                    }
                }
                // While the batches before this one are being integrated a key can already be visible whose slot value is not (create_chunk
                // writes key, then value, then slot_key[slot]): a lookup result that does not check out against slot_key only counts for
                // chunks those batches cannot be creating.  One that does check out is final, whoever is in flight -- the value read IS the
                // slot that holds this key -- and such a chunk is resident for good: without this, a chunk created by some batch stayed
                // "uncertain" for as long as it stayed in view (an uncertain item has no slot here, so it went into this batch's pending
                // set and was uncertain again for the next two: most items of a steady stream were looked up again by every one of their
                // units in the integration kernel, 4 us of dependent round trips at the head of each).
                // (nothing in flight -- the short form: no pending sets --: the map is at rest and every value read is final, no third round trip)
                const bool at_rest = !forced && !prev_pending && !prev2_pending;
                const bool verified = !forced && slot >= 0 && slot < M.max_chunks && (at_rest || M.slot_key[slot] == key);
                if (!verified && slot >= 0) slot = -1;
                const bool uncertain = !verified && (all_uncertain || (prev_pending && pending_contains(prev_pending, key, h0)) ||
                                                     (prev2_pending && pending_contains(prev2_pending, key, h0)));
                if (uncertain) {
                    // the batches in flight may be creating this chunk: the integration kernel looks it up itself, when they are over
                    slot = SLOT_LOOKUP;
                    mask = inband | carve;
                } else {
                    // a frame can only matter if it may integrate, or may carve a chunk that is resident by then
                    // (resident now, or created by an earlier frame of this batch)
                    bool resident = slot >= 0;
#pragma unroll
                    for (int j = 0; j < KL; j++) {
                        const bool in = (inband >> j) & 1u;
                        if (in || (((carve >> j) & 1u) && resident)) mask |= 1u << j;
                        resident |= in;
                    }
                }
                keep = mask != 0u;
                // the chunks this batch may create: what the next two batches' look-ups must not trust
                if (keep && slot < 0 && inband != 0u && !pending_insert(my_pending, key, h0)) my_pending[PENDING_CAPACITY] = 1;
            }
        }
        // wave64 compaction: ballot + prefix popcount, one atomic per wave
        const unsigned long long bal = __ballot(keep);
        int pos = -1;
        if (bal) {
            int base = 0;
            if (lane == (int)__builtin_ctzll(bal)) base = atomicAdd(item_count, __popcll(bal));
            base = __shfl(base, (int)__builtin_ctzll(bal));
            if (keep) {
                pos = base + __popcll(bal & ((1ull << lane) - 1ull));
                if (pos < max_items) {
                    WorkItem wi;
                    wi.x = cx; wi.y = cy; wi.z = cz;
                    wi.slot = slot;
                    wi.frame_mask = mask;
                    wi.box = pos;
                    wi.inband_mask = inband;
                    wi.pad = 0;
                    items[pos] = wi;
                    item_sync_init(sync + pos);
                    // one-frame launches are not worth brick_kernel (a launch for a kernel's worth of nothing): every brick takes the item's mask
                    if (brick_masks) {
                        constexpr int BPC = BrickGrid<N>::PER_CHUNK;
                        for (int b = 0; b < BPC; b++) brick_masks[(size_t)pos * BPC + b] = (unsigned short)mask;
                    }
                } else {
                    pos = -1;
                }
            }
        }
        s_pos[lane] = pos;
    }
    __syncthreads();
    // the flags of every (work item, frame), in work-list order: lane k of a unit of the integration kernel reads frame k's
    const int pos = s_pos[lane];
#pragma unroll UNROLL
    for (int j = 0; j < FPW; j++) {
        const int kf = (FPW > 1 && contig) ? k * FPW + j : k + j * WAVES;
        if (pos >= 0 && kf < P.n_frames) {
            FrameBox fb;
            fb.flags = s_flags[kf][lane];
            boxes[(size_t)pos * P.n_frames + kf] = fb;
        }
    }
}

}  // namespace chisel_hip
