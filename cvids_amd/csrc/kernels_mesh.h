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
//   mesh_jobs_kernel   : slots of the chunk and of its 7 "+" neighbours (the corners a border cube needs)
//   mesh_count_kernel  : vertices and grids per chunk (case table popcount + block scan)
//   mesh_emit_kernel   : same scan, then vertices and face normals into one arena
//   mesh_shade_kernel  : one thread per vertex: gradient normals and colours
// A cube is meshed only when all 8 corner voxels have weight > 0.5 and every chunk they live in exists
// (:271-276, :316-357); an absent neighbour simply reads as weight 0.
#pragma once
#include "chisel_device.h"
#include "kernels_map.h"
#include "mc_tables.h"

namespace chisel_hip {

__constant__ unsigned long long c_mc_cases[256] = CHISEL_MC_PACKED_CASES;
__constant__ unsigned char c_mc_counts[256] = CHISEL_MC_VERTEX_COUNTS;
__constant__ unsigned char c_mc_edges[12] = CHISEL_MC_EDGE_CORNERS;

struct MeshJob {
    int x, y, z;     // chunk id
    int nslot[8];    // pool slot of chunk id + (b&1, b>>1&1, b>>2&1); [0] = the chunk itself; -1 = absent
    int pad;
};

struct MeshParams {
    float res;        // voxelResolutionMeters
    float half_res;   // ChunkManager.cpp:52
    float rf_chunk;   // 1.0f / (chunkSize * res)  (ChunkManager::GetIDAt ChunkManager.h:138-140)
    float rf_voxel;   // 1.0f / res                (Chunk::GetVoxelCoords Chunk.cpp:74)
    int use_color;
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

// slot of the chunk containing `pos` (GetChunkAt ChunkManager.h:147-161); `hint`: a chunk whose slot is already known
template <int N>
__device__ inline int chunk_at(const MapView &M, const MeshParams &P, f3v pos, int hx, int hy, int hz, int hslot, f3v &origin) {
    int ix, iy, iz;
    id_at(P, pos, ix, iy, iz);
    // Chunk origin (Chunk.cpp:43): numVoxels * ID (int) * resolution
    origin = mk3((float)(N * ix) * P.res, (float)(N * iy) * P.res, (float)(N * iz) * P.res);
    if (ix == hx && iy == hy && iz == hz) return hslot;  // (callers without a hint pass hx = INT_MAX)
    if (ix < -ID_BIAS + 2 || ix > ID_BIAS - 2 || iy < -ID_BIAS + 2 || iy > ID_BIAS - 2 || iz < -ID_BIAS + 2 || iz > ID_BIAS - 2) return -1;
    return hash_find(M, ix, iy, iz);
}

// ChunkManager::GetSDF (ChunkManager.cpp:476-499).  Note the reference only range-checks the linear voxel id,
// not the coordinates (Chunk::GetVoxelID Chunk.h:81-84): reproduced.
template <int N>
__device__ inline bool get_sdf(const MapView &M, const MeshParams &P, f3v posf, int hx, int hy, int hz, int hslot, double &dist) {
    f3v origin;
    const int slot = chunk_at<N>(M, P, posf, hx, hy, hz, hslot, origin);
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

// ChunkManager::GetSDFAndGradient (ChunkManager.cpp:449-474)
template <int N>
__device__ inline bool get_sdf_and_gradient(const MapView &M, const MeshParams &P, f3v pos, int hx, int hy, int hz, int hslot,
                                            double &dist, f3v &grad) {
    const float r = P.res;
    const f3v posf = mk3(floorf(pos.x / r) * r + r / 2.0f, floorf(pos.y / r) * r + r / 2.0f, floorf(pos.z / r) * r + r / 2.0f);
    if (!get_sdf<N>(M, P, posf, hx, hy, hz, hslot, dist)) return false;
    double xp, yp, zp, xm, ym, zm;
    if (!get_sdf<N>(M, P, add3(posf, mk3(r, 0, 0)), hx, hy, hz, hslot, xp)) return false;
    if (!get_sdf<N>(M, P, add3(posf, mk3(0, r, 0)), hx, hy, hz, hslot, yp)) return false;
    if (!get_sdf<N>(M, P, add3(posf, mk3(0, 0, r)), hx, hy, hz, hslot, zp)) return false;
    if (!get_sdf<N>(M, P, sub3(posf, mk3(r, 0, 0)), hx, hy, hz, hslot, xm)) return false;
    if (!get_sdf<N>(M, P, sub3(posf, mk3(0, r, 0)), hx, hy, hz, hslot, ym)) return false;
    if (!get_sdf<N>(M, P, sub3(posf, mk3(0, 0, r)), hx, hy, hz, hslot, zm)) return false;
    grad = normalized3(mk3((float)(xp - xm), (float)(yp - ym), (float)(zp - zm)));  // grad->normalize()
    return true;
}

// ChunkManager::GetColorVoxel (ChunkManager.cpp:588-607): packed RGBW of the voxel containing `pos`
template <int N>
__device__ inline bool get_color_voxel(const MapView &M, const MeshParams &P, f3v pos, int hx, int hy, int hz, int hslot, uchar4 &out) {
    f3v origin;
    const int slot = chunk_at<N>(M, P, pos, hx, hy, hz, hslot, origin);
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

// ChunkManager::InterpolateColor (ChunkManager.cpp:501-573), including its use of integer voxel indices as metric
// positions for the 8 neighbour lookups (:506-520) and the nearest-voxel fallback Chunk::GetColorAt (Chunk.cpp:118-136)
template <int N>
__device__ inline f3v interpolate_color(const MapView &M, const MeshParams &P, f3v cp, int hx, int hy, int hz, int hslot) {
    const float r = P.res;
    const int x0 = (int)floorf(cp.x / r), y0 = (int)floorf(cp.y / r), z0 = (int)floorf(cp.z / r);
    const int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    uchar4 v000, v001, v011, v111, v110, v100, v010, v101;
    bool all = get_color_voxel<N>(M, P, mk3((float)x0, (float)y0, (float)z0), hx, hy, hz, hslot, v000);
    all = all && get_color_voxel<N>(M, P, mk3((float)x0, (float)y0, (float)z1), hx, hy, hz, hslot, v001);
    all = all && get_color_voxel<N>(M, P, mk3((float)x0, (float)y1, (float)z1), hx, hy, hz, hslot, v011);
    all = all && get_color_voxel<N>(M, P, mk3((float)x1, (float)y1, (float)z1), hx, hy, hz, hslot, v111);
    all = all && get_color_voxel<N>(M, P, mk3((float)x1, (float)y1, (float)z0), hx, hy, hz, hslot, v110);
    all = all && get_color_voxel<N>(M, P, mk3((float)x1, (float)y0, (float)z0), hx, hy, hz, hslot, v100);
    all = all && get_color_voxel<N>(M, P, mk3((float)x0, (float)y1, (float)z0), hx, hy, hz, hslot, v010);
    all = all && get_color_voxel<N>(M, P, mk3((float)x1, (float)y0, (float)z1), hx, hy, hz, hslot, v101);
    if (!all) {
        f3v origin;
        const int slot = chunk_at<N>(M, P, cp, hx, hy, hz, hslot, origin);
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
// (slot, neighbour offset); a resident neighbour of a dirty chunk gets its mesh flag set.  Ids of the neighbourhood
// that are not resident have no chunk to mesh (RecomputeMesh returns at once: ChunkManager.cpp:93-96).
__global__ void mesh_mark_kernel(MapView M, unsigned *mesh_flag) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int slot = t / 27, o = t % 27;
    if (slot >= M.max_chunks || !M.slot_dirty[slot]) return;
    const uint64_t key = M.slot_key[slot];
    if (key == KEY_EMPTY) return;
    int x, y, z;
    unpack_id(key, x, y, z);
    const int ns = (o == 13) ? slot : hash_find(M, x + o % 3 - 1, y + (o / 3) % 3 - 1, z + o / 9 - 1);
    if (ns >= 0) mesh_flag[ns] = 1u;
}

// compaction of the flagged slots into the id list of the jobs (ballot + prefix popcount); clears the flags
__global__ void mesh_collect_kernel(MapView M, unsigned *mesh_flag, int *ids, int *count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool keep = false;
    uint64_t key = KEY_EMPTY;
    if (i < M.max_chunks) {
        keep = mesh_flag[i] != 0u;
        if (keep) {
            mesh_flag[i] = 0u;
            key = M.slot_key[i];
            keep = key != KEY_EMPTY;
        }
    }
    const unsigned long long mask = __ballot(keep);
    if (!mask) return;
    const int lane = threadIdx.x & 63;
    const int leader = (int)__builtin_ctzll(mask);
    int base = 0;
    if (lane == leader) base = atomicAdd(count, __popcll(mask));
    base = __shfl(base, leader);
    if (keep) {
        const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
        int x, y, z;
        unpack_id(key, x, y, z);
        ids[3 * pos] = x;
        ids[3 * pos + 1] = y;
        ids[3 * pos + 2] = z;
    }
}

// slots of each listed chunk and its 7 "+" neighbours; one thread per (job, neighbour)
__global__ void mesh_jobs_kernel(MapView M, const int *ids, int n, MeshJob *jobs) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * 8) return;
    const int j = t >> 3, b = t & 7;
    const int x = ids[3 * j] + (b & 1), y = ids[3 * j + 1] + ((b >> 1) & 1), z = ids[3 * j + 2] + ((b >> 2) & 1);
    jobs[j].nslot[b] = hash_find(M, x, y, z);
    if (b == 0) {
        jobs[j].x = ids[3 * j];
        jobs[j].y = ids[3 * j + 1];
        jobs[j].z = ids[3 * j + 2];
        jobs[j].pad = 0;
    }
}

// corner voxel (cx, cy, cz), each in 0..N, of the cube grid of a job: (sdf, weight); absent chunk -> weight 0
template <int N>
__device__ inline float2 corner_voxel(const MapView &M, const int (&nslot)[8], int cx, int cy, int cz) {
    const int b = (cx == N ? 1 : 0) | (cy == N ? 2 : 0) | (cz == N ? 4 : 0);
    const int slot = nslot[b];
    if (slot < 0) return make_float2(0.0f, 0.0f);
    const int lx = (cx == N) ? 0 : cx, ly = (cy == N) ? 0 : cy, lz = (cz == N) ? 0 : cz;
    const size_t off = (size_t)slot * (N * N * N) + (lz * N + ly) * N + lx;
    return make_float2(M.sdf[off], M.wgt[off]);
}

// The (N+1)^3 corner voxels a chunk's cubes read -- the chunk plus one layer of its "+" neighbours -- staged once in
// LDS as (sdf, weight) pairs (39 KiB for 16^3 chunks): every voxel is fetched from HBM once per kernel instead of
// once per cube corner (8x).  32^3 chunks (287 KiB) do not fit and read their corners through L2.
template <int N>
struct CornerTile {
    static constexpr bool STAGED = (N <= 16);
    static constexpr int E = N + 1;
    static constexpr int SIZE = STAGED ? E * E * E : 1;
};

template <int N>
__device__ inline void stage_corners(const MapView &M, const int (&nslot)[8], float2 *s_vox) {
    if (!CornerTile<N>::STAGED) return;
    constexpr int E = N + 1;
    for (int i = threadIdx.x; i < E * E * E; i += blockDim.x) {
        const int cx = i % E, cy = (i / E) % E, cz = i / (E * E);
        s_vox[i] = corner_voxel<N>(M, nslot, cx, cy, cz);
    }
    __syncthreads();
}

// cube (x, y, z): corner sdf values and the case index; false when a corner is unobserved (weight <= 0.5)
template <int N>
__device__ inline bool cube_config(const MapView &M, const int (&nslot)[8], const float2 *s_vox, int x, int y, int z, float (&s)[8],
                                   int &index) {
    // cubeIndexOffsets (ChunkManager.cpp:67-69)
    const int ox[8] = {0, 1, 1, 0, 0, 1, 1, 0}, oy[8] = {0, 0, 1, 1, 0, 0, 1, 1}, oz[8] = {0, 0, 0, 0, 1, 1, 1, 1};
    constexpr int E = N + 1;
    index = 0;
    bool observed = true;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float2 v = CornerTile<N>::STAGED ? s_vox[((z + oz[i]) * E + (y + oy[i])) * E + (x + ox[i])]
                                               : corner_voxel<N>(M, nslot, x + ox[i], y + oy[i], z + oz[i]);
        observed = observed && (v.y > 0.5f);  // :271 / :352 "weight <= 0.5 -> not observed"
        s[i] = v.x;
        index |= (v.x < 0.0f) ? (1 << i) : 0;  // MarchingCubes::CalculateVertexConfiguration MarchingCubes.h:108-118
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

constexpr int MESH_BLOCK = 256;

// vertices / grids per job: counts[2*j], counts[2*j+1]
template <int N>
__global__ __launch_bounds__(MESH_BLOCK) void mesh_count_kernel(MapView M, const MeshJob *jobs, int *counts) {
    __shared__ int s_scan[MESH_BLOCK / 64][2];
    __shared__ int s_nslot[8];
    __shared__ float2 s_vox[CornerTile<N>::SIZE];
    constexpr int V = N * N * N, CPT = (V + MESH_BLOCK - 1) / MESH_BLOCK;
    const MeshJob &job = jobs[blockIdx.x];
    if (threadIdx.x < 8) s_nslot[threadIdx.x] = job.nslot[threadIdx.x];
    __syncthreads();
    int nslot[8];
#pragma unroll
    for (int i = 0; i < 8; i++) nslot[i] = s_nslot[i];
    if (nslot[0] >= 0) stage_corners<N>(M, nslot, s_vox);  // block-uniform
    int nv = 0, ng = 0;
    if (nslot[0] >= 0) {
        for (int k = 0; k < CPT; k++) {
            const int r = threadIdx.x * CPT + k;
            if (r >= V) break;
            int x, y, z, index;
            float s[8];
            cube_of_rank<N>(r, x, y, z);
            if (cube_config<N>(M, nslot, s_vox, x, y, z, s, index)) {
                const int c = c_mc_counts[index];
                nv += c;
                ng += (c != 0);  // IsOccupied (MarchingCubes.h:41-45)
            }
        }
    }
    int oa, ob, ta, tb;
    block_scan2<MESH_BLOCK>(nv, ng, oa, ob, ta, tb, s_scan);
    if (threadIdx.x == 0) {
        counts[2 * blockIdx.x] = ta;
        counts[2 * blockIdx.x + 1] = tb;
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

// offsets[2*j], offsets[2*j+1]: first vertex / grid of job j in the arena (3 floats per entry)
template <int N>
__global__ __launch_bounds__(MESH_BLOCK) void mesh_emit_kernel(MapView M, MeshParams P, const MeshJob *jobs, const int *offsets,
                                                                float *vertices, float *normals, float *grids) {
    __shared__ int s_scan[MESH_BLOCK / 64][2];
    __shared__ int s_nslot[8];
    __shared__ float2 s_vox[CornerTile<N>::SIZE];
    constexpr int V = N * N * N, CPT = (V + MESH_BLOCK - 1) / MESH_BLOCK;
    const MeshJob &job = jobs[blockIdx.x];
    if (threadIdx.x < 8) s_nslot[threadIdx.x] = job.nslot[threadIdx.x];
    __syncthreads();
    int nslot[8];
#pragma unroll
    for (int i = 0; i < 8; i++) nslot[i] = s_nslot[i];
    if (nslot[0] >= 0) stage_corners<N>(M, nslot, s_vox);  // block-uniform
    const int jx = job.x, jy = job.y, jz = job.z;
    // pass 1: this thread's share of the counts
    int nv = 0, ng = 0;
    if (nslot[0] >= 0) {
        for (int k = 0; k < CPT; k++) {
            const int r = threadIdx.x * CPT + k;
            if (r >= V) break;
            int x, y, z, index;
            float s[8];
            cube_of_rank<N>(r, x, y, z);
            if (cube_config<N>(M, nslot, s_vox, x, y, z, s, index)) {
                const int c = c_mc_counts[index];
                nv += c;
                ng += (c != 0);
            }
        }
    }
    int ov, og, tv, tg;
    block_scan2<MESH_BLOCK>(nv, ng, ov, og, tv, tg, s_scan);
    if (nslot[0] < 0 || nv == 0) return;
    size_t vpos = (size_t)offsets[2 * blockIdx.x] + ov;
    size_t gpos = (size_t)offsets[2 * blockIdx.x + 1] + og;
    const f3v origin = mk3((float)(N * jx) * P.res, (float)(N * jy) * P.res, (float)(N * jz) * P.res);  // Chunk.cpp:43
    const int ox[8] = {0, 1, 1, 0, 0, 1, 1, 0}, oy[8] = {0, 0, 1, 1, 0, 0, 1, 1}, oz[8] = {0, 0, 0, 0, 1, 1, 1, 1};
    for (int k = 0; k < CPT; k++) {
        const int r = threadIdx.x * CPT + k;
        if (r >= V) break;
        int x, y, z, index;
        float s[8];
        cube_of_rank<N>(r, x, y, z);
        if (!cube_config<N>(M, nslot, s_vox, x, y, z, s, index)) continue;
        const unsigned long long row = c_mc_cases[index];
        if ((row & 0xF) == 0xF) continue;
        // cube origin = centroid of voxel (x, y, z) + chunk origin (ChunkManager.cpp:61, :404)
        const f3v coords = add3(mk3((float)x * P.res + P.half_res, (float)y * P.res + P.half_res, (float)z * P.res + P.half_res), origin);
        grids[3 * gpos] = coords.x;
        grids[3 * gpos + 1] = coords.y;
        grids[3 * gpos + 2] = coords.z;
        gpos++;
        f3v cc[8];
#pragma unroll
        for (int i = 0; i < 8; i++)  // cornerCoords (:278-279)
            cc[i] = add3(coords, mk3((float)ox[i] * P.res, (float)oy[i] * P.res, (float)oz[i] * P.res));
        for (int t = 0; t < 15; t += 3) {
            if (((row >> (4 * t)) & 0xF) == 0xF) break;
            f3v p[3];
#pragma unroll
            for (int a = 0; a < 3; a++) {  // vertices pushed in the order t+2, t+1, t (MarchingCubes.h:86-88)
                const int e = (int)((row >> (4 * (t + 2 - a))) & 0xF);
                const int e0 = c_mc_edges[e] & 0xF, e1 = c_mc_edges[e] >> 4;
                p[a] = interpolate_vertex(cc[e0], cc[e1], s[e0], s[e1]);
            }
            const f3v fn = normalized3(cross3v(sub3(p[1], p[0]), sub3(p[2], p[0])));  // :95-101
#pragma unroll
            for (int a = 0; a < 3; a++) {
                vertices[3 * vpos] = p[a].x;
                vertices[3 * vpos + 1] = p[a].y;
                vertices[3 * vpos + 2] = p[a].z;
                normals[3 * vpos] = fn.x;
                normals[3 * vpos + 1] = fn.y;
                normals[3 * vpos + 2] = fn.z;
                vpos++;
            }
        }
    }
}

// Second half of RecomputeMesh, one thread per VERTEX (the per-cube loop above leaves most lanes idle: few cubes carry
// triangles): ComputeNormalsFromGradients (ChunkManager.cpp:609-626: the face normal stays when a lookup fails) and
// ColorizeMesh (:628-639).  Reads the vertices mesh_emit_kernel wrote (kernel boundary = visibility).
template <int N>
__global__ __launch_bounds__(MESH_BLOCK) void mesh_shade_kernel(MapView M, MeshParams P, const MeshJob *jobs, const int *offsets,
                                                                 const int *counts, const float *vertices, float *normals, float *colors) {
    const MeshJob &job = jobs[blockIdx.x];
    const int nv = counts[2 * blockIdx.x];
    if (nv == 0) return;
    const size_t base = (size_t)offsets[2 * blockIdx.x];
    const int jx = job.x, jy = job.y, jz = job.z, jslot = job.nslot[0];
    for (int v = threadIdx.x; v < nv; v += MESH_BLOCK) {
        const size_t i = base + v;
        const f3v p = mk3(vertices[3 * i], vertices[3 * i + 1], vertices[3 * i + 2]);
        double dist;
        f3v grad;
        if (get_sdf_and_gradient<N>(M, P, p, jx, jy, jz, jslot, dist, grad)) {
            const float mag = sqrtf(sum3f(grad.x * grad.x, grad.y * grad.y, grad.z * grad.z));
            if ((double)mag > 1e-12) {
                const f3v nrm = scl3(grad, 1.0f / mag);
                normals[3 * i] = nrm.x;
                normals[3 * i + 1] = nrm.y;
                normals[3 * i + 2] = nrm.z;
            }
        }
        if (P.use_color) {
            const f3v col = interpolate_color<N>(M, P, p, jx, jy, jz, jslot);
            colors[3 * i] = col.x;
            colors[3 * i + 1] = col.y;
            colors[3 * i + 2] = col.z;
        }
    }
}

// ChunkManager::GetSDF / GetSDFAndGradient for one host-supplied position (chisel_hip_get_sdf*); one thread
template <int N>
__global__ void query_sdf_kernel(MapView M, MeshParams P, float x, float y, float z, int with_gradient, double *out /* dist, gx, gy, gz, found */) {
    double dist = 0.0;
    f3v grad = mk3(0, 0, 0);
    bool ok;
    if (with_gradient)
        ok = get_sdf_and_gradient<N>(M, P, mk3(x, y, z), 0x7fffffff, 0, 0, -1, dist, grad);
    else
        ok = get_sdf<N>(M, P, mk3(x, y, z), 0x7fffffff, 0, 0, -1, dist);
    out[0] = dist;
    out[1] = grad.x;
    out[2] = grad.y;
    out[3] = grad.z;
    out[4] = ok ? 1.0 : 0.0;
}

struct MeshBuffers {
    MeshJob *jobs = nullptr;
    int *ids = nullptr;
    int *counts = nullptr;   // [2 * capacity] counts, then [2 * capacity] offsets
    int capacity = 0;        // jobs
    unsigned *flags = nullptr;  // [max_chunks] "mesh this slot"
    float *arena = nullptr;  // vertices | normals | colors | grids
    size_t arena_floats = 0;
    double *query = nullptr;
};
inline void free_mesh_buffers(MeshBuffers &b) {
    if (b.jobs) (void)hipFree(b.jobs);
    if (b.ids) (void)hipFree(b.ids);
    if (b.counts) (void)hipFree(b.counts);
    if (b.arena) (void)hipFree(b.arena);
    if (b.flags) (void)hipFree(b.flags);
    if (b.query) (void)hipFree(b.query);
    b = MeshBuffers();
}

}  // namespace chisel_hip
