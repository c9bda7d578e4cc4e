// kernels_mesh.h -- marching-cubes mesh extraction on the GPU.
//
// Replaces ChunkManager::RecomputeMeshes / RecomputeMesh / GenerateMesh / ExtractInsideVoxelMesh /
// ExtractBorderVoxelMesh (ChunkManager.cpp:91-169, 259-447), MarchingCubes::MeshCube (MarchingCubes.h:73-146),
// ColorizeMesh / InterpolateColor (ChunkManager.cpp:501-573, 628-639) and ComputeNormalsFromGradients /
// GetSDFAndGradient / GetSDF (ChunkManager.cpp:449-499, 609-626).
//
// The reference walks the cubes of a chunk serially and push_back()s into std::vectors.  Here a chunk is one
// workgroup and the output position of every cube comes from a prefix sum over the cubes *in the reference's
// traversal order* (interior cubes, then the max-x, max-y and max-z planes: ChunkManager.cpp:395-441), so the
// vertex / normal / colour / grid arrays of a chunk are element-for-element the reference's:
//   mesh_count_kernel    : per chunk: vertices and grids (case table popcount + block scan), the chunk's range in the
//                          batch and the list of its triangles
//   mesh_triangle_kernel : one thread per triangle: vertices, gradient normals and colours into one arena
// A cube is meshed only when all 8 corner voxels have weight > 0.5 and every chunk they live in exists
// (:271-276, :316-357); an absent neighbour simply reads as weight 0.
#pragma once
#include "chisel_device.h"
#include "kernels_map.h"
#include "mc_tables.h"

namespace chisel_hip {

__constant__ unsigned long long c_mc_cases[256] = CHISEL_MC_PACKED_CASES;
__constant__ __attribute__((aligned(4))) unsigned char c_mc_counts[256] = CHISEL_MC_VERTEX_COUNTS;
__constant__ unsigned char c_mc_edges[12] = CHISEL_MC_EDGE_CORNERS;

struct MeshJob {
    int x, y, z;     // chunk id
    int nb[27];      // pool slots of the 27-neighbourhood: nb[(dz+1)*9 + (dy+1)*3 + (dx+1)], [13] = the chunk itself; -1 = absent.
                     // Cube corners need the 7 "+" neighbours, the gradient / colour lookups of border vertices any of the 26.
    int pad[2];
};
constexpr int NB_SELF = 13;

struct MeshParams {
    float res;        // voxelResolutionMeters
    float half_res;   // ChunkManager.cpp:52
    float rf_chunk;   // 1.0f / (chunkSize * res)  (ChunkManager::GetIDAt ChunkManager.h:138-140)
    float rf_voxel;   // 1.0f / res                (Chunk::GetVoxelCoords Chunk.cpp:74)
    int use_color;
    int stages;       // bit 0: ComputeNormalsFromGradients, bit 1: ColorizeMesh (both after ChunkManager::RecomputeMesh; neither: GenerateMesh alone)
};

struct f3v {
    float x, y, z;
};
__device__ inline f3v mk3(float x, float y, float z) { return f3v{x, y, z}; }
__device__ inline f3v add3(f3v a, f3v b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ inline f3v sub3(f3v a, f3v b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ inline f3v scl3(f3v a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ inline float sum3f(float a0, float a1, float a2) { return a0 + (a1 + a2); }  // Eigen 3-term reduction order
__device__ inline f3v cross3v(f3v a, f3v b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ inline f3v normalized3(f3v a) {  // Eigen 3.3 MatrixBase::normalized(): z > 0 ? a / sqrt(z) : a
    const float z = sum3f(a.x * a.x, a.y * a.y, a.z * a.z);
    if (z > 0.0f) {
        const float s = sqrtf(z);
        return mk3(a.x / s, a.y / s, a.z / s);
    }
    return a;
}

// the reference's traversal position `r` (0 .. N^3-1) -> cube index (ChunkManager.cpp:395-441)
template <int N>
__device__ inline void cube_of_rank(int r, int &x, int &y, int &z) {
    constexpr int A = (N - 1) * (N - 1) * (N - 1), B = (N - 1) * N, C = (N - 1) * (N - 1);
    if (r < A) {  // interior: z, y, x < N-1, x fastest
        x = r % (N - 1);
        y = (r / (N - 1)) % (N - 1);
        z = r / ((N - 1) * (N - 1));
    } else if (r < A + B) {  // max x plane: z < N-1 outer, y < N inner
        r -= A;
        x = N - 1;
        y = r % N;
        z = r / N;
    } else if (r < A + B + C) {  // max y plane: z < N-1 outer, x < N-1 inner
        r -= A + B;
        y = N - 1;
        x = r % (N - 1);
        z = r / (N - 1);
    } else {  // max z plane: y < N outer, x < N inner
        r -= A + B + C;
        z = N - 1;
        x = r % N;
        y = r / N;
    }
}

// ChunkManager::GetIDAt (ChunkManager.h:136-145)
__device__ inline void id_at(const MeshParams &P, f3v pos, int &ix, int &iy, int &iz) {
    ix = (int)floorf(pos.x * P.rf_chunk);
    iy = (int)floorf(pos.y * P.rf_chunk);
    iz = (int)floorf(pos.z * P.rf_chunk);
}

// slot of the chunk containing `pos` (GetChunkAt ChunkManager.h:147-161); (hx, hy, hz) / nb: a chunk whose 27-neighbourhood
// slots are already known (MeshJob::nb; nullptr: none) -- a hash probe only for chunks further away
template <int N>
__device__ inline int chunk_at(const MapView &M, const MeshParams &P, f3v pos, int hx, int hy, int hz, const int *nb, f3v &origin) {
    int ix, iy, iz;
    id_at(P, pos, ix, iy, iz);
    // Chunk origin (Chunk.cpp:43): numVoxels * ID (int) * resolution
    origin = mk3((float)(N * ix) * P.res, (float)(N * iy) * P.res, (float)(N * iz) * P.res);
    if (nb) {
        const unsigned dx = (unsigned)(ix - hx + 1), dy = (unsigned)(iy - hy + 1), dz = (unsigned)(iz - hz + 1);
        if (dx < 3u && dy < 3u && dz < 3u) return nb[(dz * 3u + dy) * 3u + dx];
    }
    if (ix < -ID_BIAS + 2 || ix > ID_BIAS - 2 || iy < -ID_BIAS + 2 || iy > ID_BIAS - 2 || iz < -ID_BIAS + 2 || iz > ID_BIAS - 2) return -1;
    return hash_find(M, ix, iy, iz);
}

// ChunkManager::GetSDF (ChunkManager.cpp:476-499).  Note the reference only range-checks the linear voxel id,
// not the coordinates (Chunk::GetVoxelID Chunk.h:81-84): reproduced.
template <int N>
__device__ inline bool get_sdf(const MapView &M, const MeshParams &P, f3v posf, int hx, int hy, int hz, const int *nb, double &dist) {
    f3v origin;
    const int slot = chunk_at<N>(M, P, posf, hx, hy, hz, nb, origin);
    if (slot < 0) return false;
    const f3v rel = sub3(posf, origin);
    const int cx = (int)floorf(rel.x * P.rf_voxel), cy = (int)floorf(rel.y * P.rf_voxel), cz = (int)floorf(rel.z * P.rf_voxel);
    const int id = (cz * N + cy) * N + cx;
    if (id >= 0 && id < N * N * N) {
        const size_t off = (size_t)slot * (N * N * N) + id;
        if ((double)M.wgt[off] > 1e-12) {
            dist = (double)M.sdf[off];
            return true;
        }
    }
    return false;
}

// ChunkManager::GetSDFAndGradient (ChunkManager.cpp:449-474).  The reference makes the seven GetSDF calls one after
// the other and gives up at the first failure; here the seven voxel addresses are resolved first and their weights and
// distances requested together (14 independent loads instead of a chain of 14), then judged in the reference's order.
template <int N>
__device__ inline bool get_sdf_and_gradient(const MapView &M, const MeshParams &P, f3v pos, int hx, int hy, int hz, const int *nb,
                                            double &dist, f3v &grad) {
    const float r = P.res;
    const f3v posf = mk3(floorf(pos.x / r) * r + r / 2.0f, floorf(pos.y / r) * r + r / 2.0f, floorf(pos.z / r) * r + r / 2.0f);
    f3v q[7];
    q[0] = posf;
    q[1] = add3(posf, mk3(r, 0, 0));
    q[2] = add3(posf, mk3(0, r, 0));
    q[3] = add3(posf, mk3(0, 0, r));
    q[4] = sub3(posf, mk3(r, 0, 0));
    q[5] = sub3(posf, mk3(0, r, 0));
    q[6] = sub3(posf, mk3(0, 0, r));
    float w[7], d[7];
    bool ok[7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        f3v origin;
        const int slot = chunk_at<N>(M, P, q[i], hx, hy, hz, nb, origin);
        const f3v rel = sub3(q[i], origin);
        const int cx = (int)floorf(rel.x * P.rf_voxel), cy = (int)floorf(rel.y * P.rf_voxel), cz = (int)floorf(rel.z * P.rf_voxel);
        const int id = (cz * N + cy) * N + cx;
        ok[i] = slot >= 0 && id >= 0 && id < N * N * N;  // GetSDF: chunk present, linear voxel id in range (Chunk.h:81-84)
        const size_t off = ok[i] ? (size_t)slot * (N * N * N) + id : 0;
        w[i] = M.wgt[off];
        d[i] = M.sdf[off];
    }
#pragma unroll
    for (int i = 0; i < 7; i++)
        if (!(ok[i] && (double)w[i] > 1e-12)) return false;
    dist = (double)d[0];
    grad = normalized3(mk3((float)((double)d[1] - (double)d[4]), (float)((double)d[2] - (double)d[5]), (float)((double)d[3] - (double)d[6])));  // grad->normalize()
    return true;
}

// ChunkManager::GetColorVoxel (ChunkManager.cpp:588-607): packed RGBW of the voxel containing `pos`
template <int N>
__device__ inline bool get_color_voxel(const MapView &M, const MeshParams &P, f3v pos, int hx, int hy, int hz, const int *nb, uchar4 &out) {
    f3v origin;
    const int slot = chunk_at<N>(M, P, pos, hx, hy, hz, nb, origin);
    if (slot < 0) return false;
    const f3v rel = sub3(pos, origin);
    const int cx = (int)floorf(rel.x * P.rf_voxel), cy = (int)floorf(rel.y * P.rf_voxel), cz = (int)floorf(rel.z * P.rf_voxel);
    const int id = (cz * N + cy) * N + cx;
    if (id >= 0 && id < N * N * N) {
        out = M.rgbw[(size_t)slot * (N * N * N) + id];
        return true;
    }
    return false;
}

#ifndef MESH_COLOR_BATCH
#define MESH_COLOR_BATCH 0  // (round 4: the eight colour lookups located and requested together instead of one after the other -- two dependent round
                            // trips instead of sixteen, and SLOWER: mesh_triangle_kernel 18.2 -> 23.9 us on the default window, 21.8 -> 29.0 us on the
                            // driver's.  The sequential form stops at the first absent neighbour and touches one line at a time; kept.)
#endif
// ChunkManager::InterpolateColor (ChunkManager.cpp:501-573), including its use of integer voxel indices as metric
// positions for the 8 neighbour lookups (:506-520) and the nearest-voxel fallback Chunk::GetColorAt (Chunk.cpp:118-136)
template <int N>
__device__ inline f3v interpolate_color(const MapView &M, const MeshParams &P, f3v cp, int hx, int hy, int hz, const int *nb) {
    const float r = P.res;
    const int x0 = (int)floorf(cp.x / r), y0 = (int)floorf(cp.y / r), z0 = (int)floorf(cp.z / r);
    const int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    uchar4 v000, v001, v011, v111, v110, v100, v010, v101;
#if MESH_COLOR_BATCH
    // The reference makes the eight GetColorVoxel calls one after the other and stops at the first failure (:506-520); a lookup has no
    // side effect and the blend below needs all eight, so the eight voxels are located first (eight neighbour-table reads in flight),
    // their colours requested together (eight more), and the verdict taken afterwards: two dependent round trips instead of sixteen.
    bool all;
    {
        const int qx[8] = {x0, x0, x0, x1, x1, x1, x0, x1}, qy[8] = {y0, y0, y1, y1, y1, y0, y1, y0}, qz[8] = {z0, z1, z1, z1, z0, z0, z0, z1};
        int slot[8], id[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const f3v q = mk3((float)qx[i], (float)qy[i], (float)qz[i]);
            f3v origin;
            slot[i] = chunk_at<N>(M, P, q, hx, hy, hz, nb, origin);
            const f3v rel = sub3(q, origin);
            const int cx = (int)floorf(rel.x * P.rf_voxel), cy = (int)floorf(rel.y * P.rf_voxel), cz = (int)floorf(rel.z * P.rf_voxel);
            id[i] = (cz * N + cy) * N + cx;
        }
        uchar4 c[8];
        all = true;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const bool ok = slot[i] >= 0 && id[i] >= 0 && id[i] < N * N * N;  // GetColorVoxel: chunk present, linear voxel id in range (:588-607)
            c[i] = M.rgbw[ok ? (size_t)slot[i] * (N * N * N) + id[i] : 0];
            all = all && ok;
        }
        v000 = c[0]; v001 = c[1]; v011 = c[2]; v111 = c[3]; v110 = c[4]; v100 = c[5]; v010 = c[6]; v101 = c[7];
    }
#else
    bool all = get_color_voxel<N>(M, P, mk3((float)x0, (float)y0, (float)z0), hx, hy, hz, nb, v000);
    all = all && get_color_voxel<N>(M, P, mk3((float)x0, (float)y0, (float)z1), hx, hy, hz, nb, v001);
    all = all && get_color_voxel<N>(M, P, mk3((float)x0, (float)y1, (float)z1), hx, hy, hz, nb, v011);
    all = all && get_color_voxel<N>(M, P, mk3((float)x1, (float)y1, (float)z1), hx, hy, hz, nb, v111);
    all = all && get_color_voxel<N>(M, P, mk3((float)x1, (float)y1, (float)z0), hx, hy, hz, nb, v110);
    all = all && get_color_voxel<N>(M, P, mk3((float)x1, (float)y0, (float)z0), hx, hy, hz, nb, v100);
    all = all && get_color_voxel<N>(M, P, mk3((float)x0, (float)y1, (float)z0), hx, hy, hz, nb, v010);
    all = all && get_color_voxel<N>(M, P, mk3((float)x1, (float)y0, (float)z1), hx, hy, hz, nb, v101);
#endif
    if (!all) {
        f3v origin;
        const int slot = chunk_at<N>(M, P, cp, hx, hy, hz, nb, origin);
        if (slot < 0) return mk3(0.0f, 0.0f, 0.0f);
        // Chunk::GetColorAt: AABB::Contains, then (int)((pos - origin) / res)
        const float size = (float)N * P.res;
        const bool contains = cp.x >= origin.x && cp.y >= origin.y && cp.z >= origin.z && cp.x <= origin.x + size &&
                              cp.y <= origin.y + size && cp.z <= origin.z + size;
        if (contains) {
            const int cx = (int)((cp.x - origin.x) / P.res), cy = (int)((cp.y - origin.y) / P.res), cz = (int)((cp.z - origin.z) / P.res);
            if (cx >= 0 && cx < N && cy >= 0 && cy < N && cz >= 0 && cz < N) {
                const uchar4 c = M.rgbw[(size_t)slot * (N * N * N) + (cz * N + cy) * N + cx];
                return mk3((float)c.x / 255.0f, (float)c.y / 255.0f, (float)c.z / 255.0f);
            }
        }
        return mk3(0.0f, 0.0f, 0.0f);
    }
    const float xd = (cp.x - (float)x0) / (float)(x1 - x0);
    const float yd = (cp.y - (float)y0) / (float)(y1 - y0);
    const float zd = (cp.z - (float)z0) / (float)(z1 - z0);
    float out[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        auto g = [ch](const uchar4 &v) -> float { return (float)(ch == 0 ? v.x : (ch == 1 ? v.y : v.z)); };
        const float c00 = g(v000) * (1 - xd) + g(v100) * xd;
        const float c10 = g(v010) * (1 - xd) + g(v110) * xd;
        const float c01 = g(v001) * (1 - xd) + g(v101) * xd;
        const float c11 = g(v011) * (1 - xd) + g(v111) * xd;
        const float c0 = c00 * (1 - yd) + c10 * yd;
        const float c1 = c01 * (1 - yd) + c11 * yd;
        const float c = c0 * (1 - zd) + c1 * zd;
        out[ch] = c / 255.0f;
    }
    return mk3(out[0], out[1], out[2]);
}

// meshesToUpdate on the device (Chisel.h:175-189 marks the 27-neighbourhood of every updated chunk): one thread per
// (slot, neighbour offset); a resident neighbour of a dirty chunk becomes a job -- the thread whose exchange sets the
// neighbour's flag first appends its id to the job list (no compaction kernel behind this one: every launch on the map's
// stream costs >= 5 us there).  Ids of the neighbourhood that are not resident have no chunk to mesh (RecomputeMesh returns
// at once: ChunkManager.cpp:93-96).  The job counter `n_jobs` was zeroed by the previous recompute's launch of this kernel,
// which zeroes `n_jobs_next` for the next one (two counters alternate: zeroing the one in use here would race with the
// appends); flags and dirty bits are reset by the count kernel, job by job (every dirty slot is a job: offset 13 is itself).
// (rebuilds / completes the job list from the dirty flags: after point clouds, uploads, or when the list kept by the integration kernel
// has been given up)
__global__ void mesh_mark_kernel(MapView M, unsigned *mesh_flag, int *ids, int *n_jobs) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    // the dirty slots come from the list the integration kernels keep (mark_slot_dirty): the work here is proportional to what
    // changed, not to the size of the pool; a list that overflowed (slots dirtied, removed and dirtied again many times over
    // without a recompute in between) falls back to the flags of all slots
    const unsigned listed = M.slot_dirty[2 * (size_t)M.max_chunks];
    const bool scan = listed > (unsigned)M.max_chunks;
    const long long n = 27ll * (scan ? (long long)M.max_chunks : (long long)listed);
    for (long long t = gid; t < n; t += stride) {
        const int entry = (int)(t / 27), o = (int)(t % 27);
        const int slot = scan ? entry : (int)M.slot_dirty[(size_t)M.max_chunks + entry];
        if (!M.slot_dirty[slot]) continue;  // (cleared since it was listed: the chunk was removed)
        const uint64_t key = M.slot_key[slot];
        if (key == KEY_EMPTY) continue;
        int x, y, z;
        unpack_id(key, x, y, z);
        const int ns = (o == 13) ? slot : hash_find_quiescent(M, x + o % 3 - 1, y + (o / 3) % 3 - 1, z + o / 9 - 1);
        if (ns >= 0) mesh_append_job(M, mesh_flag, ns, ids, n_jobs);
    }
}
// (rare) resident chunks the host wants meshed as well: neighbourhoods of chunks that were removed while dirty
__global__ void mesh_append_kernel(MapView M, unsigned *mesh_flag, const int *slots, int n, int *ids, int *n_jobs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && slots[i] >= 0) mesh_append_job(M, mesh_flag, slots[i], ids, n_jobs);
}

#ifndef MESH_THREADS
#define MESH_THREADS 512
#endif
constexpr int MESH_BLOCK_THREADS = MESH_THREADS;  // per-chunk kernel: 8 cubes (16^3) per thread (256: 33 us, 512: 24 us, 1024: 35 us per recompute)

// corner voxel (cx, cy, cz), each in 0..N, of the cube grid of a job: (sdf, weight); absent chunk -> weight 0
template <int N>
__device__ inline float2 corner_voxel(const MapView &M, const int *nb, int cx, int cy, int cz) {
    const int slot = nb[NB_SELF + (cx == N ? 1 : 0) + (cy == N ? 3 : 0) + (cz == N ? 9 : 0)];
    if (slot < 0) return make_float2(0.0f, 0.0f);
    const int lx = (cx == N) ? 0 : cx, ly = (cy == N) ? 0 : cy, lz = (cz == N) ? 0 : cz;
    const size_t off = (size_t)slot * (N * N * N) + (lz * N + ly) * N + lx;
    return make_float2(M.sdf[off], M.wgt[off]);
}

// The (N+1)^3 corner voxels a chunk's cubes read -- the chunk plus one layer of its "+" neighbours -- staged once in LDS:
// every voxel is fetched from HBM once per kernel instead of once per cube corner (8x).  What the cubes need of a corner is its
// distance and whether it has been observed (weight > 0.5: ChunkManager.cpp:271 / :352), so one float per corner carries both:
// the distance, or NaN for "not observed" (19.6 KiB per 16^3 chunk instead of 39 KiB of (sdf, weight) pairs: four chunks per CU
// are in flight instead of two).  [A stored distance that is itself NaN under a weight above 0.5 would read as unobserved here
// and as a NaN vertex in the reference; integration cannot produce one: invalid depth pixels never update a voxel.]
// The chunk's own N^3 voxels arrive as 16-byte loads (x rows are contiguous in the pool), the 3 N^2 + 3 N + 1 border corners of
// the "+" neighbours one by one.  32^3 chunks (140 KiB) stay in L2 and fold on the fly.
template <int N>
struct CornerTile {
    static constexpr bool STAGED = (N <= 16);
    static constexpr int E = N + 1;
    static constexpr int SIZE = STAGED ? E * E * E : 1;
};
__device__ inline float fold_corner(float sdf, float weight) { return weight > 0.5f ? sdf : __builtin_nanf(""); }

template <int N>
__device__ inline void stage_corners(const MapView &M, const int *nb, float *s_vox) {
    if (!CornerTile<N>::STAGED) return;
    constexpr int E = N + 1, V = N * N * N, Q = V / 4, QU = (Q + MESH_BLOCK_THREADS - 1) / MESH_BLOCK_THREADS;
    constexpr int BORDER = E * E * E - V, BU = (BORDER + MESH_BLOCK_THREADS - 1) / MESH_BLOCK_THREADS;
    const int self = nb[NB_SELF];
    const float4 *sp = reinterpret_cast<const float4 *>(M.sdf + (size_t)self * V);
    const float4 *wp = reinterpret_cast<const float4 *>(M.wgt + (size_t)self * V);
    float4 vs[QU], vw[QU];
    float2 vb[BU];
    // every load of the thread is requested before the first is used: one round trip
#pragma unroll
    for (int u = 0; u < QU; u++) {
        const int q = min((int)threadIdx.x + u * MESH_BLOCK_THREADS, Q - 1);
        vs[u] = sp[q];
        vw[u] = wp[q];
    }
#pragma unroll
    for (int u = 0; u < BU; u++) {
        // border corners in three groups: the plane cz = N (E x E, its edges included), the plane cy = N below it (E x N), the plane
        // cx = N below both (N x N)
        const int b = min((int)threadIdx.x + u * MESH_BLOCK_THREADS, BORDER - 1);
        int cx, cy, cz;
        if (b < E * E) { cx = b % E; cy = b / E; cz = N; }
        else if (b < E * E + N * E) { const int t = b - E * E; cx = t % E; cy = N; cz = t / E; }
        else { const int t = b - E * E - N * E; cx = N; cy = t % N; cz = t / N; }
        vb[u] = corner_voxel<N>(M, nb, cx, cy, cz);
    }
#pragma unroll
    for (int u = 0; u < QU; u++) {
        const int q = (int)threadIdx.x + u * MESH_BLOCK_THREADS;
        if (q < Q) {
            const int i = 4 * q, x = i % N, y = (i / N) % N, z = i / (N * N);
            float *d = s_vox + (z * E + y) * E + x;
            d[0] = fold_corner(vs[u].x, vw[u].x);
            d[1] = fold_corner(vs[u].y, vw[u].y);
            d[2] = fold_corner(vs[u].z, vw[u].z);
            d[3] = fold_corner(vs[u].w, vw[u].w);
        }
    }
#pragma unroll
    for (int u = 0; u < BU; u++) {
        const int b = (int)threadIdx.x + u * MESH_BLOCK_THREADS;
        if (b < BORDER) {
            int cx, cy, cz;
            if (b < E * E) { cx = b % E; cy = b / E; cz = N; }
            else if (b < E * E + N * E) { const int t = b - E * E; cx = t % E; cy = N; cz = t / E; }
            else { const int t = b - E * E - N * E; cx = N; cy = t % N; cz = t / N; }
            s_vox[(cz * E + cy) * E + cx] = fold_corner(vb[u].x, vb[u].y);
        }
    }
    __syncthreads();
}

// cube (x, y, z): corner sdf values and the case index; false when a corner is unobserved (weight <= 0.5)
template <int N>
__device__ inline bool cube_config(const MapView &M, const int *nb, const float *s_vox, int x, int y, int z, float (&s)[8],
                                   int &index) {
    // cubeIndexOffsets (ChunkManager.cpp:67-69)
    const int ox[8] = {0, 1, 1, 0, 0, 1, 1, 0}, oy[8] = {0, 0, 1, 1, 0, 0, 1, 1}, oz[8] = {0, 0, 0, 0, 1, 1, 1, 1};
    constexpr int E = N + 1;
    index = 0;
    bool observed = true;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float v;
        if (CornerTile<N>::STAGED) {
            v = s_vox[((z + oz[i]) * E + (y + oy[i])) * E + (x + ox[i])];
        } else {
            const float2 c = corner_voxel<N>(M, nb, x + ox[i], y + oy[i], z + oz[i]);
            v = fold_corner(c.x, c.y);
        }
        observed = observed && (v == v);  // :271 / :352 "weight <= 0.5 -> not observed"
        s[i] = v;
        index |= (v < 0.0f) ? (1 << i) : 0;  // MarchingCubes::CalculateVertexConfiguration MarchingCubes.h:108-118
    }
    return observed;
}

// exclusive block scan of (a, b) pairs over BLOCK threads; returns this thread's offsets, totals in (ta, tb)
template <int BLOCK>
__device__ inline void block_scan2(int a, int b, int &oa, int &ob, int &ta, int &tb, int (*s_a)[2]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int ia = a, ib = b;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int na = __shfl_up(ia, o), nb = __shfl_up(ib, o);
        if (lane >= o) {
            ia += na;
            ib += nb;
        }
    }
    if (lane == 63) {
        s_a[wave][0] = ia;
        s_a[wave][1] = ib;
    }
    __syncthreads();
    int wa = 0, wb = 0;
    ta = tb = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) {
        if (w < wave) {
            wa += s_a[w][0];
            wb += s_a[w][1];
        }
        ta += s_a[w][0];
        tb += s_a[w][1];
    }
    oa = wa + ia - a;
    ob = wb + ib - b;
    __syncthreads();
}

constexpr int MESH_BLOCK = MESH_BLOCK_THREADS;
#ifndef MESH_COUNT_WAVES
#define MESH_COUNT_WAVES 5  // waves per SIMD the count kernel is compiled for: up to 96 registers, two 512-thread workgroups per CU.  (6 -- 80 registers,
                            // three workgroups per CU -- spills 32 bytes per lane and is slower although a third more jobs are resident: 22.5 against
                            // 21.4 us per recompute on the default window, 33.8 against 29.3 us on the driver's; 4 -- 128 registers -- 21.9 / 30.1)
#endif
constexpr int MESH_TRI_BLOCK = 256;  // per-triangle kernel

// What the host needs to know about a job after a recompute (one 32-byte record, fetched in one copy)
struct JobInfo {
    int x, y, z;       // chunk id
    int present;       // the chunk is resident (RecomputeMesh returns at once otherwise, ChunkManager.cpp:93-96)
    int n_vertices, n_grids;
    int tri_base;      // first triangle of the job in the batch's numbering (its vertices start at 3 * tri_base)
    int grid_base;     // first grid entry
};

// One triangle of the batch: which job, which cube (rank in the reference's traversal order), which case, which of the
// cube's triangles; gidx = position of the cube among the job's occupied cubes (its entry in Mesh::grids).
struct TriRec {
    unsigned job;
    unsigned code;   // rank << 11 | case index << 3 | triangle number
    unsigned gidx;   // the cube's entry in the batch's grid numbering (job's grid base + position among the job's occupied cubes)
};
// Beside the triangle list, per occupied cube (indexed like the grids): its eight corner distances (all observed: the cube carries
// triangles).  The count kernel has them in LDS; the triangle kernel would fetch each through two dependent loads (neighbour
// table, then voxel).
struct alignas(16) CubeCorners {
    float s[8];
};
constexpr int MESH_LIST_CAP = 1024;  // occupied cubes of a job listed in LDS at a time
// s[e] for a per-lane e (a register array cannot be indexed per lane: seven selects)
__device__ inline float pick8(const float (&s)[8], int e) {
    const float a0 = (e & 1) ? s[1] : s[0], a1 = (e & 1) ? s[3] : s[2], a2 = (e & 1) ? s[5] : s[4], a3 = (e & 1) ? s[7] : s[6];
    const float b0 = (e & 2) ? a1 : a0, b1 = (e & 2) ? a3 : a2;
    return (e & 4) ? b1 : b0;
}

// Per chunk (one workgroup): stage the corners, count (case table popcount) and scan the cubes in the reference's
// traversal order, reserve the chunk's range of the batch's triangle / grid numbering with one atomic each and list its
// triangles.  info[j] = what the host keeps of job j: sizes and its first triangle / grid in the batch (the chunks' ranges
// follow one another in completion order; within a chunk the order is the reference's).
// totals[0..1] = running totals (the atomics), totals[2] = set when the triangle list is too small (the host retries).
#ifdef CHISEL_PHASES
__device__ unsigned long long g_mesh_phase[8];  // diagnostic: 10 ns ticks per stage of mesh_count_kernel (thread 0 of every workgroup), [7] = jobs
#define MSTAMP(i) do { if (threadIdx.x == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); atomicAdd(&g_mesh_phase[i], n_ - mt_); mt_ = n_; } } while (0)
#else
#define MSTAMP(i) do { } while (0)
#endif
template <int N>
__global__ __launch_bounds__(MESH_BLOCK, (N > 16 ? 4 : MESH_COUNT_WAVES)) void mesh_count_kernel(MapView M, const int *__restrict__ ids, MeshJob *jobs, const int *__restrict__ n_jobs,
                                                                 JobInfo *info, int *totals, TriRec *tris, CubeCorners *corners, int tri_capacity, unsigned *mesh_flag, int keep_dirty) {
    __shared__ int s_scan[MESH_BLOCK / 64][2];
    __shared__ int s_nb[27];
    __shared__ unsigned s_sum[8];  // sign summaries of the chunk and its seven "+" neighbours (slot_summary)
    __shared__ int s_base[2];
    __shared__ float s_vox[CornerTile<N>::SIZE];
    __shared__ __attribute__((aligned(16))) unsigned char s_case[N * N * N];  // case index per cube, by traversal rank
    __shared__ unsigned s_list_a[MESH_LIST_CAP], s_list_t[MESH_LIST_CAP];  // occupied cubes of the job: rank | case << 16, first triangle (relative)
    __shared__ unsigned s_counts[64];  // the 256 vertex counts of the case table, four to a word (a per-lane index into constant memory is a global load)
    constexpr int V = N * N * N, CPT = (V + MESH_BLOCK - 1) / MESH_BLOCK;
    // (the first job's id is requested together with the job count: the id buffer holds at least 4096 entries -- more than the
    // grid has workgroups -- whatever the count; ensure_mesh_jobs)
    int jx0 = ids[3 * blockIdx.x], jy0 = ids[3 * blockIdx.x + 1], jz0 = ids[3 * blockIdx.x + 2];
    int n = *n_jobs;  // the job count stays on the device: the grid is persistent
    if (ids == M.mesh_jobs && n > M.mesh_jobs_capacity) n = M.mesh_jobs_capacity;  // (the kept list never gets there: the host gives it up first)
    if (blockIdx.x == 0 && threadIdx.x == 0) totals[3] = n;  // where the kernels behind this one (and a second emission) read it
    if (threadIdx.x < 64) s_counts[threadIdx.x] = reinterpret_cast<const unsigned *>(c_mc_counts)[threadIdx.x];  // (first barrier below)
    // the list of dirty slots has been consumed by the mark kernel -- also when none of them became a job (every listed chunk
    // removed since): reset outside the job loop
    if (blockIdx.x == 0 && threadIdx.x == 29 && !keep_dirty) M.slot_dirty[2 * (size_t)M.max_chunks] = 0u;
    for (int j = blockIdx.x; j < n; j += gridDim.x) {
#ifdef CHISEL_PHASES
    unsigned long long mt_ = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) atomicAdd(&g_mesh_phase[7], 1ull);
#endif
    __syncthreads();  // the previous job's LDS contents are no longer read
    // the job record: the chunk id and the pool slots of its 27-neighbourhood (27 hash lookups, one per thread); kept for
    // the triangle kernel and the host
    const bool first = j == (int)blockIdx.x;
    const int jx = first ? jx0 : ids[3 * j], jy = first ? jy0 : ids[3 * j + 1], jz = first ? jz0 : ids[3 * j + 2];
    if (threadIdx.x < 27) {
        const int o = threadIdx.x;
        const int dx = o % 3 - 1, dy = (o / 3) % 3 - 1, dz = o / 9 - 1;
        const int slot = hash_find_quiescent(M, jx + dx, jy + dy, jz + dz);
        s_nb[o] = slot;
        jobs[j].nb[o] = slot;
        if (dx >= 0 && dy >= 0 && dz >= 0) s_sum[dx + 2 * dy + 4 * dz] = slot >= 0 ? slot_summary(M)[slot] : 0u;  // the chunks that hold cube corners
    } else if (threadIdx.x == 27) {
        jobs[j].x = jx;
        jobs[j].y = jy;
        jobs[j].z = jz;
        jobs[j].pad[0] = jobs[j].pad[1] = 0;
    }
    __syncthreads();
    const bool present = s_nb[NB_SELF] >= 0;  // block-uniform
    // Can any cube of this chunk carry a triangle?  Corner 0 of every cube is a voxel of the chunk itself and must be observed
    // (ChunkManager.cpp:271 / :352), and a case other than 0 / 255 needs both signs among the cube's corners, which lie in the chunk
    // and its seven "+" neighbours.  About half of the jobs of a recompute -- chunks inside the band but away from the surface --
    // stop here: no corners staged, nothing classified.
    bool can = present;
    {
        const unsigned seen = (s_sum[0] | s_sum[1]) | (s_sum[2] | s_sum[3]) | (s_sum[4] | s_sum[5]) | (s_sum[6] | s_sum[7]);
        can = can && s_sum[0] != 0u && seen == SUM_ANY;
    }
    if (threadIdx.x == 28 && present && !keep_dirty) {
        // meshesToUpdate.clear() (Chisel.cpp:57) and the job flag of mesh_mark_kernel, which has finished: this job's own
        if (mesh_flag) mesh_flag[s_nb[NB_SELF]] = 0u;
        M.slot_dirty[s_nb[NB_SELF]] = 0u;
    }
    MSTAMP(0);
    if (can) stage_corners<N>(M, s_nb, s_vox);
    MSTAMP(1);
    int nv = 0, ng = 0;
    unsigned char cases[CPT];  // case index of the thread's cubes that carry triangles (0: none -- case 0 has no triangles either)
#pragma unroll
    for (int k = 0; k < CPT; k++) cases[k] = 0;
    if (can) {
        // Classification with neighbouring lanes on neighbouring cubes (rank k * MESH_BLOCK + thread: consecutive corners, no LDS
        // bank conflicts -- with a thread's own CPT consecutive cubes the lanes sit 8 corners apart, an 8-way conflict on every one
        // of the 64 corner reads); the case bytes go through LDS to the thread that owns the cube in the traversal order.
#ifndef MESH_CLASSIFY_UNROLL
#define MESH_CLASSIFY_UNROLL 2
#endif
#pragma unroll MESH_CLASSIFY_UNROLL  // (two cubes' sixteen corner reads in flight; all of them at once would cost a workgroup per CU in registers)
        for (int k = 0; k < CPT; k++) {
            const int r = k * MESH_BLOCK + (int)threadIdx.x;
            if (r < V) {
                int x, y, z, index;
                float sc[8];
                cube_of_rank<N>(r, x, y, z);
                unsigned char cs = 0;
                if (cube_config<N>(M, s_nb, s_vox, x, y, z, sc, index)) {
                    const int c = (int)((s_counts[index >> 2] >> ((index & 3) * 8)) & 0xffu);
                    cs = c ? (unsigned char)index : 0;
                }
                s_case[r] = cs;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CPT; k++) {
            const int r = threadIdx.x * CPT + k;
            if (r < V) {
                const int index = s_case[r];
                const int c = (int)((s_counts[index >> 2] >> ((index & 3) * 8)) & 0xffu);  // (case 0 has no vertices)
                nv += c;
                ng += (c != 0);  // IsOccupied (MarchingCubes.h:41-45)
                cases[k] = (unsigned char)index;
            }
        }
    }
    int ov, og, tv, tg;
    asm volatile("" ::"v"(nv));
    MSTAMP(2);
    block_scan2<MESH_BLOCK>(nv, ng, ov, og, tv, tg, s_scan);
    MSTAMP(3);
    if (threadIdx.x == 0) {
        const int tb = tv ? atomicAdd(&totals[0], tv / 3) : 0;
        const int gb = tg ? atomicAdd(&totals[1], tg) : 0;
        JobInfo ji;
        ji.x = jx; ji.y = jy; ji.z = jz;
        ji.present = present ? 1 : 0;
        ji.n_vertices = tv;
        ji.n_grids = tg;
        ji.tri_base = tb;
        ji.grid_base = gb;
        info[j] = ji;
        s_base[0] = tb;
        s_base[1] = gb;
        if (tb + tv / 3 > tri_capacity || gb + tg > tri_capacity) totals[2] = 1;  // (a cube has at least one triangle: the host grows both lists by the triangle total)
    }
    if (tv == 0) continue;  // block-uniform
    __syncthreads();
    MSTAMP(4);
    const int tb = s_base[0], gb = s_base[1];
    if (tb + tv / 3 > tri_capacity || gb + tg > tri_capacity) continue;  // block-uniform
    // The occupied cubes (a few hundred of the 4096, unevenly spread over the threads) are listed in LDS and emitted by all threads
    // together -- one cube each instead of up to CPT in a row on a few lanes --, MESH_LIST_CAP of them per round.
    for (int base = 0; base < tg; base += MESH_LIST_CAP) {
        if (base) __syncthreads();  // the previous round's list has been read
        int trel = ov / 3, gidx = og;
#pragma unroll
        for (int k = 0; k < CPT; k++) {
            const int index = cases[k];
            if (index == 0) continue;
            if ((unsigned)(gidx - base) < (unsigned)MESH_LIST_CAP) {
                s_list_a[gidx - base] = (unsigned)(threadIdx.x * CPT + k) | ((unsigned)index << 16);
                s_list_t[gidx - base] = (unsigned)trel;
            }
            trel += (int)((s_counts[index >> 2] >> ((index & 3) * 8)) & 0xffu) / 3;
            gidx++;
        }
        __syncthreads();
        const int n_round = min(tg - base, MESH_LIST_CAP);
        for (int e = threadIdx.x; e < n_round; e += MESH_BLOCK) {
            const unsigned a = s_list_a[e];
            const int r = (int)(a & 0xffffu), index = (int)(a >> 16), t0 = tb + (int)s_list_t[e];
            const int nt = (int)((s_counts[index >> 2] >> ((index & 3) * 8)) & 0xffu) / 3;
            CubeCorners cc;
            int x, y, z, idx2;
            cube_of_rank<N>(r, x, y, z);
            (void)cube_config<N>(M, s_nb, s_vox, x, y, z, cc.s, idx2);
            corners[gb + base + e] = cc;
            for (int t = 0; t < nt; t++) {
                TriRec rec;
                rec.job = (unsigned)j;
                rec.code = ((unsigned)r << 11) | ((unsigned)index << 3) | (unsigned)t;
                rec.gidx = (unsigned)(gb + base + e);
                tris[t0 + t] = rec;
            }
        }
    }
    MSTAMP(5);
    }
}

// MarchingCubes::InterpolateVertex (MarchingCubes.h:135-146), including "vertex1 + 0.5 * vertex2" (sic)
__device__ inline f3v interpolate_vertex(f3v v1, f3v v2, float sdf1, float sdf2) {
    const float minDiff = 1e-6;
    const float sdfDiff = sdf1 - sdf2;
    if (fabsf(sdfDiff) < minDiff) return add3(v1, scl3(v2, 0.5f));
    const float t = sdf1 / sdfDiff;
    return add3(v1, scl3(sub3(v2, v1), t));
}

// corner i of cube (x, y, z): cubeIndexOffsets (ChunkManager.cpp:67-69) = {0,1,1,0,0,1,1,0 / 0,0,1,1,0,0,1,1 / 0,0,0,0,1,1,1,1}
__device__ inline int corner_ox(int i) { return ((i + 1) >> 1) & 1; }
__device__ inline int corner_oy(int i) { return (i >> 1) & 1; }
__device__ inline int corner_oz(int i) { return i >> 2; }

// One thread per vertex of the batch's triangles (the cubes that carry triangles are few and unevenly spread over the
// chunks, so a per-chunk loop leaves most lanes idle): MeshCube (MarchingCubes.h:73-106) for its triangle -- vertices
// pushed in the order t+2, t+1, t, face normal -- then the second half of RecomputeMesh for the thread's own vertex:
// ComputeNormalsFromGradients (ChunkManager.cpp:609-626: the face normal stays when a lookup fails) and ColorizeMesh
// (:628-639), which read the map, not the mesh.  Triangle i owns vertices 3i .. 3i+2 of the arena; the first triangle of
// a cube also writes the cube's grid entry.
template <int N>
// The totals of the count kernel are read from the device (totals[0] triangles, totals[1] grids): the arena --
// vertices | normals | colours | grids -- was picked before they were known.  A batch that does not fit its arena
// writes nothing (the host, which reads the same totals, then runs the kernel again on a larger one).
// Housekeeping that rides along (an extra kernel, copy or event on the map's stream would sit on the critical path in front of the
// next integration): workgroup 0 writes the per-job records into pinned host memory (`host_info`, at most max_jobs of them; the
// host needs them at the next recompute) and then `seq` into host_flags[6].
__global__ __launch_bounds__(MESH_TRI_BLOCK) void mesh_triangle_kernel(MapView M, MeshParams P, const MeshJob *__restrict__ jobs,
                                                                    const JobInfo *__restrict__ info, const TriRec *__restrict__ tris,
                                                                    const CubeCorners *__restrict__ corners, const int *__restrict__ totals, float *arena, size_t arena_floats, int *host_info,
                                                                    volatile int *host_flags, int max_jobs, int seq, int publish) {
    const int n_tris = totals[0];
    const int n_jobs = totals[3];
    if ((publish & 1) && blockIdx.x == 0 && threadIdx.x == 0) {  // (publish: bit 0 = totals to the host, bit 1 = the kept job list was this recompute's input)
        // The recompute's totals, straight into pinned host memory as ONE 16-byte store -- {triangles, grids, jobs | overflow << 31,
        // sequence number} -- before anything else: the host polls word 3 for this recompute's sequence number when the caller next
        // touches the map.  No copy engine, no event, no stream wait and no kernel of its own in between (an event record on the
        // map's stream costs a barrier packet of 7-12 us in front of the next kernel, a one-thread kernel 5 us).
        uint4 v;
        v.x = (unsigned)n_tris;
        v.y = (unsigned)totals[1];
        v.z = (unsigned)n_jobs | (totals[2] ? 0x80000000u : 0u);
        v.w = (unsigned)seq;
        *reinterpret_cast<uint4 *>(const_cast<int *>(host_flags)) = v;
        // what the host polls is a second copy of the sequence number behind a system-scope fence: that the 16 bytes above arrive
        // as one piece is how the bus behaves, not a guarantee (this thread alone pays the few hundred nanoseconds)
        __threadfence_system();
        host_flags[5] = seq;
        // the job list the integration kernels keep has been consumed by the count kernel (its number is in totals[3]): empty again
        if ((publish & 2) && M.mesh_ctl) M.mesh_ctl[4] = 0;
    }
    const size_t nv3 = (size_t)n_tris * 9, ng3 = (size_t)totals[1] * 3;
    // a triangle list that overflowed (totals[2]) is incomplete: nothing is emitted, the host lists and emits again
    const bool fits = totals[2] == 0 && nv3 * (P.use_color ? 3 : 2) + ng3 <= arena_floats;
    if (blockIdx.x == 0) {
        const int *src = reinterpret_cast<const int *>(info);
        const int n = min(n_jobs, max_jobs) * 8;
        for (int i = threadIdx.x; i < n; i += MESH_TRI_BLOCK) host_info[i] = src[i];
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) host_flags[6] = seq;
    }
    if (fits) {
    float *vertices = arena, *normals = arena + nv3, *colors = P.use_color ? arena + 2 * nv3 : nullptr;
    float *grids = arena + nv3 * (P.use_color ? 3 : 2);
    // one thread per VERTEX: all three vertices of its triangle (the face normal needs them: three cheap interpolations),
    // then the costly part -- seven voxel lookups for the gradient, the colour lookups -- for its own vertex only
    for (int vi = blockIdx.x * MESH_TRI_BLOCK + threadIdx.x; vi < 3 * n_tris; vi += gridDim.x * MESH_TRI_BLOCK) {
    const int i = vi / 3, mine = vi - 3 * i;
    const TriRec rec = tris[i];
    const MeshJob &job = jobs[rec.job];  // stays in memory (L1 / L2): its neighbour table is indexed per lane
    const int *nb = job.nb;
    const int jx = job.x, jy = job.y, jz = job.z;
    const int r = (int)(rec.code >> 11), index = (int)((rec.code >> 3) & 0xffu), t = 3 * (int)(rec.code & 7u);
    int x, y, z;
    cube_of_rank<N>(r, x, y, z);
    const unsigned long long row = c_mc_cases[index];
    const f3v origin = mk3((float)(N * jx) * P.res, (float)(N * jy) * P.res, (float)(N * jz) * P.res);  // Chunk.cpp:43
    // cube origin = centroid of voxel (x, y, z) + chunk origin (ChunkManager.cpp:61, :404)
    const f3v coords = add3(mk3((float)x * P.res + P.half_res, (float)y * P.res + P.half_res, (float)z * P.res + P.half_res), origin);
    const CubeCorners cc = corners[rec.gidx];
    if (t == 0 && mine == 0) {
        const size_t g = (size_t)rec.gidx;
        grids[3 * g] = coords.x;
        grids[3 * g + 1] = coords.y;
        grids[3 * g + 2] = coords.z;
    }
    f3v p[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const int ed = (int)((row >> (4 * (t + 2 - a))) & 0xF);
        const int e0 = c_mc_edges[ed] & 0xF, e1 = c_mc_edges[ed] >> 4;
        // cornerCoords (:278-279) and corner sdf of the two ends of the edge
        const f3v c0 = add3(coords, mk3((float)corner_ox(e0) * P.res, (float)corner_oy(e0) * P.res, (float)corner_oz(e0) * P.res));
        const f3v c1 = add3(coords, mk3((float)corner_ox(e1) * P.res, (float)corner_oy(e1) * P.res, (float)corner_oz(e1) * P.res));
        const float s0 = pick8(cc.s, e0), s1 = pick8(cc.s, e1);  // (corner i of the cube: cubeIndexOffsets column i, as cube_config reads them)
        p[a] = interpolate_vertex(c0, c1, s0, s1);
    }
    const f3v fn = normalized3(cross3v(sub3(p[1], p[0]), sub3(p[2], p[0])));  // :95-101
    const f3v pv = mine == 0 ? p[0] : (mine == 1 ? p[1] : p[2]);
    float *vo = vertices + 3 * (size_t)vi;
    float *no = normals + 3 * (size_t)vi;
    vo[0] = pv.x;
    vo[1] = pv.y;
    vo[2] = pv.z;
    f3v nrm = fn;
    double dist;
    f3v grad;
    if ((P.stages & 1) && get_sdf_and_gradient<N>(M, P, pv, jx, jy, jz, nb, dist, grad)) {
        const float mag = sqrtf(sum3f(grad.x * grad.x, grad.y * grad.y, grad.z * grad.z));
        if ((double)mag > 1e-12) nrm = scl3(grad, 1.0f / mag);
    }
    no[0] = nrm.x;
    no[1] = nrm.y;
    no[2] = nrm.z;
    if (colors) {
        const f3v col = (P.stages & 2) ? interpolate_color<N>(M, P, pv, jx, jy, jz, nb) : mk3(0.0f, 0.0f, 0.0f);
        float *co = colors + 3 * (size_t)vi;
        co[0] = col.x;
        co[1] = col.y;
        co[2] = col.z;
    }
    }
    }
}

// ChunkManager::ExtractInsideVoxelMesh / ExtractBorderVoxelMesh (ChunkManager.cpp:259-379) for ONE cube of a resident chunk: the eight
// corners -- voxel `index` + cubeIndexOffsets, taken from the neighbour chunk where a coordinate leaves [0, N) on either side (:318-333) --,
// all of them observed (weight > 0.5) or nothing; then MarchingCubes::MeshCube (MarchingCubes.h:73-106) with the caller's cube
// coordinates.  One thread.  out: [0] vertices written (0, 3 .. 15), [1] IsOccupied (a grid entry follows), then 15 x 3 vertex floats and
// 15 x 3 normal floats (the face normal of each triangle, thrice).
template <int N>
__global__ void mesh_one_cube_kernel(MapView M, MeshParams P, int jx, int jy, int jz, int ix, int iy, int iz, float cx, float cy, float cz, float *out) {
    float sdf[8];
    bool observed = true;
    for (int i = 0; i < 8 && observed; i++) {
        int c[3] = {ix + corner_ox(i), iy + corner_oy(i), iz + corner_oz(i)};
        int off[3] = {0, 0, 0};
        for (int a = 0; a < 3; a++) {
            if (c[a] < 0) { off[a] = -1; c[a] = N - 1; }
            else if (c[a] >= N) { off[a] = 1; c[a] = 0; }
        }
        const int slot = hash_find_quiescent(M, jx + off[0], jy + off[1], jz + off[2]);
        if (slot < 0) { observed = false; break; }
        const size_t o = (size_t)slot * (N * N * N) + (c[2] * N + c[1]) * N + c[0];
        if (!(M.wgt[o] > 0.5f)) { observed = false; break; }
        sdf[i] = M.sdf[o];
    }
    out[0] = 0.0f;
    out[1] = 0.0f;
    if (!observed) return;
    int index = 0;
    for (int i = 0; i < 8; i++) index |= (sdf[i] < 0.0f) ? (1 << i) : 0;  // CalculateVertexConfiguration MarchingCubes.h:108-118
    const int nv = c_mc_counts[index];
    const unsigned long long row = c_mc_cases[index];
    const f3v coords = mk3(cx, cy, cz);
    for (int t = 0; t < nv; t += 3) {
        f3v p[3];
        for (int a = 0; a < 3; a++) {
            const int ed = (int)((row >> (4 * (t + 2 - a))) & 0xF);
            const int e0 = c_mc_edges[ed] & 0xF, e1 = c_mc_edges[ed] >> 4;
            const f3v c0 = add3(coords, mk3((float)corner_ox(e0) * P.res, (float)corner_oy(e0) * P.res, (float)corner_oz(e0) * P.res));
            const f3v c1 = add3(coords, mk3((float)corner_ox(e1) * P.res, (float)corner_oy(e1) * P.res, (float)corner_oz(e1) * P.res));
            p[a] = interpolate_vertex(c0, c1, sdf[e0], sdf[e1]);
        }
        const f3v fn = normalized3(cross3v(sub3(p[1], p[0]), sub3(p[2], p[0])));
        for (int a = 0; a < 3; a++) {
            float *v = out + 2 + 3 * (t + a), *n = out + 2 + 45 + 3 * (t + a);
            v[0] = p[a].x; v[1] = p[a].y; v[2] = p[a].z;
            n[0] = fn.x; n[1] = fn.y; n[2] = fn.z;
        }
    }
    out[0] = (float)nv;
    out[1] = nv ? 1.0f : 0.0f;  // IsOccupied (MarchingCubes.h:41-45): the case has a triangle
}

// MarchingCubes::MeshCube (MarchingCubes.h:73-106), ::InterpolateEdgeVertices (:120-132) and ::CalculateVertexConfiguration (:108-118) for a
// caller's own cube: eight corner coordinates (3 x 8, column-major as Eigen stores Matrix<float, 3, 8>) and eight distances, no map
// involved (the statics of the reference's class are public).  One thread.  out: [0] case index, [1] vertices written (0, 3 .. 15),
// then 12 x 3 edge coordinates (an edge without a sign change keeps zeros: the reference leaves that column unset), 15 x 3 vertex
// floats in the order MeshCube pushes them (t + 2, t + 1, t) and 15 x 3 normal floats (each triangle's face normal, thrice).
__global__ void mesh_cube_values_kernel(const float *coords, const float *sdf_in, float *out) {
    float sdf[8];
    f3v c[8];
    for (int i = 0; i < 8; i++) {
        sdf[i] = sdf_in[i];
        c[i] = mk3(coords[3 * i], coords[3 * i + 1], coords[3 * i + 2]);
    }
    int index = 0;
    for (int i = 0; i < 8; i++) index |= (sdf[i] < 0.0f) ? (1 << i) : 0;
    f3v edge[12];
    for (int e = 0; e < 12; e++) {
        const int e0 = c_mc_edges[e] & 0xF, e1 = c_mc_edges[e] >> 4;
        edge[e] = mk3(0.0f, 0.0f, 0.0f);
        if ((sdf[e0] < 0.0f && sdf[e1] >= 0.0f) || (sdf[e0] >= 0.0f && sdf[e1] < 0.0f)) edge[e] = interpolate_vertex(c[e0], c[e1], sdf[e0], sdf[e1]);
        out[2 + 3 * e] = edge[e].x;
        out[2 + 3 * e + 1] = edge[e].y;
        out[2 + 3 * e + 2] = edge[e].z;
    }
    const int nv = c_mc_counts[index];
    const unsigned long long row = c_mc_cases[index];
    float *v = out + 2 + 36, *n = out + 2 + 36 + 45;
    for (int t = 0; t < nv; t += 3) {
        f3v p[3];
        for (int a = 0; a < 3; a++) p[a] = edge[(int)((row >> (4 * (t + 2 - a))) & 0xF)];
        const f3v fn = normalized3(cross3v(sub3(p[1], p[0]), sub3(p[2], p[0])));
        for (int a = 0; a < 3; a++) {
            v[3 * (t + a)] = p[a].x; v[3 * (t + a) + 1] = p[a].y; v[3 * (t + a) + 2] = p[a].z;
            n[3 * (t + a)] = fn.x; n[3 * (t + a) + 1] = fn.y; n[3 * (t + a) + 2] = fn.z;
        }
    }
    out[0] = (float)index;
    out[1] = (float)nv;
}
// MarchingCubes::InterpolateVertex (MarchingCubes.h:135-146) for one pair; one thread
__global__ void interpolate_vertex_kernel(const float *in /* v1 xyz, v2 xyz, sdf1, sdf2 */, float *out) {
    const f3v r = interpolate_vertex(mk3(in[0], in[1], in[2]), mk3(in[3], in[4], in[5]), in[6], in[7]);
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
}

// ChunkManager::GetSDF / GetSDFAndGradient for one host-supplied position (chisel_hip_get_sdf*); one thread
template <int N>
__global__ void query_sdf_kernel(MapView M, MeshParams P, float x, float y, float z, int with_gradient, double *out /* dist, gx, gy, gz, found */) {
    double dist = 0.0;
    f3v grad = mk3(0, 0, 0);
    bool ok;
    if (with_gradient)
        ok = get_sdf_and_gradient<N>(M, P, mk3(x, y, z), 0, 0, 0, nullptr, dist, grad);
    else
        ok = get_sdf<N>(M, P, mk3(x, y, z), 0, 0, 0, nullptr, dist);
    out[0] = dist;
    out[1] = grad.x;
    out[2] = grad.y;
    out[3] = grad.z;
    out[4] = ok ? 1.0 : 0.0;
}

// The second half of ChunkManager::RecomputeMesh for a caller's own vertex list (ChunkManager::ComputeNormalsFromGradients
// ChunkManager.cpp:609-626 and ::ColorizeMesh :628-639 are public): one thread per vertex; a normal is replaced only where the
// gradient lookup succeeds, colours are written for every vertex.  stages: bit 0 normals, bit 1 colours.
template <int N>
__global__ void shade_vertices_kernel(MapView M, MeshParams P, const float *__restrict__ verts, long long n, float *normals, float *colors, int stages) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3v pv = mk3(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
    if ((stages & 1) && normals) {
        double dist;
        f3v grad;
        if (get_sdf_and_gradient<N>(M, P, pv, 0, 0, 0, nullptr, dist, grad)) {
            const float mag = sqrtf(sum3f(grad.x * grad.x, grad.y * grad.y, grad.z * grad.z));
            if ((double)mag > 1e-12) {
                const f3v nrm = scl3(grad, 1.0f / mag);
                normals[3 * i] = nrm.x;
                normals[3 * i + 1] = nrm.y;
                normals[3 * i + 2] = nrm.z;
            }
        }
    }
    if ((stages & 2) && colors && M.rgbw) {
        const f3v col = interpolate_color<N>(M, P, pv, 0, 0, 0, nullptr);
        colors[3 * i] = col.x;
        colors[3 * i + 1] = col.y;
        colors[3 * i + 2] = col.z;
    }
}

struct MeshBuffers {
    MeshJob *jobs = nullptr;
    int *ids = nullptr;
    JobInfo *info = nullptr; // [capacity] per-job results of a recompute
    int *totals = nullptr;   // [8] = MapView::mesh_ctl: triangles, grids, triangle-list overflow flag, jobs; [4]: entries of the job list the integration kernels keep
    int *n_jobs = nullptr;   // where the count kernel finds the number of jobs (the kept list's length, or null: totals[3] and the caller's ids)
    TriRec *tris = nullptr;  // triangle list of one recompute
    CubeCorners *corners = nullptr;  // [tri_capacity] corner distances of its occupied cubes
    int tri_capacity = 0;
    int capacity = 0;        // jobs
    unsigned *flags = nullptr;  // [max_chunks] "this slot is in the job list" = MapView::mesh_flag
    double *query = nullptr;
    float *cube = nullptr;      // result of mesh_one_cube_kernel
};
inline void free_mesh_buffers(MeshBuffers &b) {
    if (b.jobs) (void)hipFree(b.jobs);
    if (b.ids) (void)hipFree(b.ids);
    if (b.info) (void)hipFree(b.info);
    if (b.totals) (void)hipFree(b.totals);
    if (b.tris) (void)hipFree(b.tris);
    if (b.corners) (void)hipFree(b.corners);
    if (b.flags) (void)hipFree(b.flags);
    if (b.query) (void)hipFree(b.query);
    if (b.cube) (void)hipFree(b.cube);
    b = MeshBuffers();
}

}  // namespace chisel_hip
