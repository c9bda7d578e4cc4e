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
#ifndef MESH_GRAD_X3
#define MESH_GRAD_X3 1  // (round 6: triangle kernel 26.2 -> 24.6 us on the driver's window, 20.9 -> 20.4 on the default)
#endif
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
#if MESH_GRAD_X3
    // The three voxels along x (-x, centre, +x: q[4], q[0], q[1]) are neighbours in memory whenever the reference's own index
    // arithmetic puts them into one row of one chunk: then ONE 12-byte access per array fetches them (the kernel's time is the
    // number of scattered lane accesses its gathers make, not their bytes).  The indices are the reference's, computed as before;
    // only the access is merged.
    size_t offs[7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        f3v origin;
        const int slot = chunk_at<N>(M, P, q[i], hx, hy, hz, nb, origin);
        const f3v rel = sub3(q[i], origin);
        const int cx = (int)floorf(rel.x * P.rf_voxel), cy = (int)floorf(rel.y * P.rf_voxel), cz = (int)floorf(rel.z * P.rf_voxel);
        const int id = (cz * N + cy) * N + cx;
        ok[i] = slot >= 0 && id >= 0 && id < N * N * N;  // GetSDF: chunk present, linear voxel id in range (Chunk.h:81-84)
        offs[i] = ok[i] ? (size_t)slot * (N * N * N) + id : 0;
    }
    struct __attribute__((packed, aligned(4))) F3 { float a, b, c; };
    const bool row = ok[0] && ok[1] && ok[4] && offs[1] == offs[0] + 1 && offs[4] + 1 == offs[0];
    if (row) {
        const F3 w3 = *reinterpret_cast<const F3 *>(M.wgt + offs[4]), d3 = *reinterpret_cast<const F3 *>(M.sdf + offs[4]);
        w[4] = w3.a; w[0] = w3.b; w[1] = w3.c;
        d[4] = d3.a; d[0] = d3.b; d[1] = d3.c;
    } else {
        w[0] = M.wgt[offs[0]]; d[0] = M.sdf[offs[0]];
        w[1] = M.wgt[offs[1]]; d[1] = M.sdf[offs[1]];
        w[4] = M.wgt[offs[4]]; d[4] = M.sdf[offs[4]];
    }
    w[2] = M.wgt[offs[2]]; d[2] = M.sdf[offs[2]];
    w[3] = M.wgt[offs[3]]; d[3] = M.sdf[offs[3]];
    w[5] = M.wgt[offs[5]]; d[5] = M.sdf[offs[5]];
    w[6] = M.wgt[offs[6]]; d[6] = M.sdf[offs[6]];
#else
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
#endif
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

// ChunkManager::InterpolateColor (ChunkManager.cpp:501-573), including its use of integer voxel indices as metric
// positions for the 8 neighbour lookups (:506-520) and the nearest-voxel fallback Chunk::GetColorAt (Chunk.cpp:118-136)
template <int N>
__device__ inline f3v interpolate_color(const MapView &M, const MeshParams &P, f3v cp, int hx, int hy, int hz, const int *nb) {
    const float r = P.res;
    const int x0 = (int)floorf(cp.x / r), y0 = (int)floorf(cp.y / r), z0 = (int)floorf(cp.z / r);
    const int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    uchar4 v000, v001, v011, v111, v110, v100, v010, v101;
    // The first of the eight look-ups asks for the chunk at "position" (x0, y0, z0) -- voxel INDICES taken for metres (:506-508), i.e.
    // a chunk 1 / res chunks further out than the vertex -- and the reference stops at the first failure.  An id outside the box of
    // every id ever created is absent: no hash probe for (nearly) every vertex of a map that does not span hundreds of metres.
    bool first_may_exist = true;
    if (M.mesh_ctl) {
        int ix, iy, iz;
        id_at(P, mk3((float)x0, (float)y0, (float)z0), ix, iy, iz);
        const int *bb = M.mesh_ctl + MC_BBOX;
        first_may_exist = ix >= bb[0] && iy >= bb[1] && iz >= bb[2] && ix <= bb[3] && iy <= bb[4] && iz <= bb[5];
    }
    bool all = first_may_exist && get_color_voxel<N>(M, P, mk3((float)x0, (float)y0, (float)z0), hx, hy, hz, nb, v000);
    all = all && get_color_voxel<N>(M, P, mk3((float)x0, (float)y0, (float)z1), hx, hy, hz, nb, v001);
    all = all && get_color_voxel<N>(M, P, mk3((float)x0, (float)y1, (float)z1), hx, hy, hz, nb, v011);
    all = all && get_color_voxel<N>(M, P, mk3((float)x1, (float)y1, (float)z1), hx, hy, hz, nb, v111);
    all = all && get_color_voxel<N>(M, P, mk3((float)x1, (float)y1, (float)z0), hx, hy, hz, nb, v110);
    all = all && get_color_voxel<N>(M, P, mk3((float)x1, (float)y0, (float)z0), hx, hy, hz, nb, v100);
    all = all && get_color_voxel<N>(M, P, mk3((float)x0, (float)y1, (float)z0), hx, hy, hz, nb, v010);
    all = all && get_color_voxel<N>(M, P, mk3((float)x1, (float)y0, (float)z1), hx, hy, hz, nb, v101);
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

// corner voxel (cx, cy, cz), each in 0..N, of the cube grid of a job: (sdf, weight); absent chunk -> weight 0
template <int N>
__device__ inline float2 corner_voxel(const MapView &M, const int *nb, int cx, int cy, int cz) {
    const int slot = nb[NB_SELF + (cx == N ? 1 : 0) + (cy == N ? 3 : 0) + (cz == N ? 9 : 0)];
    if (slot < 0) return make_float2(0.0f, 0.0f);
    const int lx = (cx == N) ? 0 : cx, ly = (cy == N) ? 0 : cy, lz = (cz == N) ? 0 : cz;
    const size_t off = (size_t)slot * (N * N * N) + (lz * N + ly) * N + lx;
    return make_float2(M.sdf[off], M.wgt[off]);
}
// What the cubes need of a corner is its distance and whether it has been observed (weight > 0.5: ChunkManager.cpp:271 / :352), so
// one float per corner carries both: the distance, or NaN for "not observed".  [A stored distance that is itself NaN under a weight
// above 0.5 would read as unobserved here and as a NaN vertex in the reference; integration cannot produce one: invalid depth pixels
// never update a voxel.]
__device__ inline float fold_corner(float sdf, float weight) { return weight > 0.5f ? sdf : __builtin_nanf(""); }

// ---- mesh_count_kernel: one WAVE per sub-job ------------------------------------------------------------------------------------
// Rounds 1-4 gave a chunk to one 512-thread workgroup (look-ups -> stage (N+1)^3 corners -> classify -> block scan -> reserve -> list,
// five barriers): a recompute lasted as long as its longest job plus a second generation of workgroups (850 jobs on 512 resident), 0.10-0.14
// of the HBM roofline.  Round 5: a job is cut into SUB-JOBS along the reference's traversal order (ChunkManager.cpp:395-441: interior cubes
// x fastest, then y, then z; then the max-x, max-y and max-z planes), each a box of cubes whose own order "x fastest, then y, then z" IS
// the traversal order of its section:
//     sub-jobs 0 .. NI-1   interior cubes of ZL consecutive z-layers (16^3: 8 slabs of 2 layers = 450 cubes; their corners are voxels of
//                          the chunk itself and contiguous in the pool: staged as 16-byte loads, no neighbour needed)
//     NI, NI+1, NI+2       the max-x, max-y, max-z plane (240 / 225 / 256 cubes; corners in up to 4 / 2 / 8 chunks)
// One single-wave workgroup per sub-job (several thousand per recompute, all resident at once: a slot is refilled the moment its wave
// retires), no barrier between waves, 8 (16 for 32^3 chunks) cubes per lane.  What ties the sub-jobs of a job together travels through
// memory-side atomics only:
//   * cnt[job][s] = the sub-job's triangles | occupied cubes << 16 (a plain store: its readers are the NEXT kernel's threads, which add
//     up the entries in front of their own sub-job -- the position of a triangle in its chunk's arrays is its job's base + that prefix +
//     its position in the sub-job, so the arrays come out element for element in the reference's order whatever order the waves ran in);
//   * job_acc[job] += arrivals << 48 | cubes << 24 | triangles (one returning atomic): the wave that arrives last knows the job's totals,
//     reserves the job's range of the recompute's triangle / grid numbering (one atomic on the totals), writes the job's JobInfo and
//     leaves the accumulator at zero for the next recompute;
//   * the records the triangle kernel works from -- per occupied cube its eight corner distances, per triangle a 16-byte TriRec -- go
//     into UNORDERED lists: MESH_PARTS partitions with a cursor each (a single cursor would take ~ 88 atomics per us: 6 000 sub-jobs =
//     70 us), sub-job u appends to partition u mod MESH_PARTS with one returning 64-bit atomic issued beside the arrival (lane 0 / lane 1
//     of ONE instruction: one round trip).
template <int N>
struct MeshGeom {
    static constexpr int ZL = N == 16 ? 2 : (N == 8 ? 7 : 1);  // z-layers of interior cubes per sub-job
    static constexpr int NI = (N - 1 + ZL - 1) / ZL;           // interior sub-jobs
    static constexpr int S = NI + 3;                           // sub-jobs per job: 4 (8^3), 11 (16^3), 34 (32^3)
    static constexpr int CPL = N == 32 ? 16 : 8;               // cubes per lane
    static constexpr int MAXC = 64 * CPL;                      // >= cubes of any sub-job (343 / 450 / 1024)
    static constexpr int TILE = N == 32 ? 33 * 33 * 2 : N * N * (ZL + 1);  // >= corners of any sub-job's box
    static constexpr int ROW = (2 + S + 3) / 4 * 4;            // ints of a job's row of `cnt`: its triangle / grid base, then the sub-jobs' figures (16^3: one 64-byte line)
    static_assert((N - 1) * (N - 1) * ZL <= MAXC && N * N <= MAXC, "cubes per sub-job");
};
// cube box (BX x BY x bz cubes from (X0, Y0, z0)) and corner tile (TX x TY x (bz + 1)) of a sub-job type: 0 interior slab, 1 max-x, 2 max-y, 3 max-z
template <int N, int TYPE>
struct MeshBox {
    static constexpr int BX = TYPE == 1 ? 1 : (TYPE == 3 ? N : N - 1);
    static constexpr int BY = TYPE == 2 ? 1 : (TYPE == 0 ? N - 1 : N);
    static constexpr int TX = BX + 1, TY = BY + 1;
    static constexpr int X0 = TYPE == 1 ? N - 1 : 0, Y0 = TYPE == 2 ? N - 1 : 0;
    static constexpr unsigned PLUS = TYPE == 0 ? 0x01u : (TYPE == 1 ? 0x0Fu : (TYPE == 2 ? 0x05u : 0xFFu));  // "+" chunks (bit dx + 2 dy + 4 dz) that hold its corners
};
constexpr int MESH_PARTS = 64;  // partitions of the unordered record lists
// MapView::mesh_ctl / MeshBuffers::totals (ints): the recompute's totals -- [0] triangles and [1] grids are ONE 64-bit word for the
// last arrivers' atomic --, [2] record-list overflow, [3] jobs; [4] entries of the job list the integration kernels keep;
// from [8] on MESH_PARTS 64-bit cursors (triangles | cubes << 32).  [0..2] and the cursors start every recompute at zero.
enum { MC_TRIS = 0, MC_GRIDS = 1, MC_OVERFLOW = 2, MC_JOBS = 3, MC_KEPT = 4, MC_CURSORS = 8, MC_INTS = MC_BBOX + 8 };  // (MC_BBOX: chisel_device.h)
static_assert(MC_BBOX == MC_CURSORS + 2 * MESH_PARTS, "the created-id box sits behind the cursors");
static_assert(MC_LATCH > MC_KEPT && MC_LATCH < MC_CURSORS, "a free word in front of the cursors");
static_assert(MC_CURSORS == 8 && MESH_PARTS == 64 && MC_KEPT == 4, "kernels_integrate.h / kernels_map.h address these words by number");

#ifndef MESH_COUNT_WAVES
#define MESH_COUNT_WAVES 6  // waves per SIMD the count kernel is compiled for (its 6 KB of LDS per wave allow 27 per CU of 160 KB)
#endif
#ifndef MESH_COUNT_VGPRS
#define MESH_COUNT_VGPRS 80  // ... said in registers as well: the compiler prices the LDS against 64 KB per CU (three waves per SIMD) and then
                             // spends 151 registers
#endif
constexpr int MESH_TRI_BLOCK = 256;  // per-triangle kernel
#ifndef MESH_TRI_WAVES
#define MESH_TRI_WAVES 1  // waves per SIMD the triangle kernel is compiled for (1 = whatever its registers allow: six)
#endif

// What the host needs to know about a job after a recompute (one 32-byte record, fetched in one copy)
struct JobInfo {
    int x, y, z;       // chunk id
    int present;       // the chunk is resident (RecomputeMesh returns at once otherwise, ChunkManager.cpp:93-96)
    int n_vertices, n_grids;
    int tri_base;      // first triangle of the job in the batch's numbering (its vertices start at 3 * tri_base)
    int grid_base;     // first grid entry
};

// One triangle of the batch (an entry of the unordered list): which job and sub-job, which cube, which case, which of the cube's triangles,
// and where it goes: rel = position among its sub-job's triangles | position of its cube among the sub-job's occupied cubes << 16.
struct alignas(16) TriRec {
    unsigned job;
    unsigned code;   // cube x | y << 5 | z << 10 | case index << 15 | triangle number << 23 | sub-job << 26
    unsigned cube;   // the cube's CubeCorners entry
    unsigned rel;
};
// Per occupied cube: its eight corner distances (all observed: the cube carries triangles).  The count kernel has them in LDS; the
// triangle kernel would fetch each through two dependent loads (neighbour table, then voxel).
struct alignas(16) CubeCorners {
    float s[8];
};
// s[e] for a per-lane e (a register array cannot be indexed per lane: seven selects)
__device__ inline float pick8(const float (&s)[8], int e) {
    const float a0 = (e & 1) ? s[1] : s[0], a1 = (e & 1) ? s[3] : s[2], a2 = (e & 1) ? s[5] : s[4], a3 = (e & 1) ? s[7] : s[6];
    const float b0 = (e & 2) ? a1 : a0, b1 = (e & 2) ? a3 : a2;
    return (e & 4) ? b1 : b0;
}

#ifdef CHISEL_PHASES
__device__ unsigned long long g_mesh_phase[8];  // diagnostic: 10 ns ticks per stage of mesh_count_kernel (lane 0 of every wave), [7] = sub-jobs
#define MSTAMP(i) do { if (lane == 0 && (blockIdx.x & 31) == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); atomicAdd(&g_mesh_phase[i], n_ - mt_); mt_ = n_; } } while (0)  // (every 32nd wave: the atomics of all of them on six words would be what is measured)
#else
#define MSTAMP(i) do { } while (0)
#endif
#ifdef CHISEL_PHASES
__device__ unsigned long long g_tri_phase[8];  // diagnostic: 10 ns ticks per stage of mesh_triangle_kernel (lane 0 of every 32nd wave), [6] = kernel start -> this wave's start, [7] = waves
#define TSTAMP(i, dep) do { asm volatile("" ::"v"(dep)); if ((threadIdx.x & 63) == 0 && ((blockIdx.x * (MESH_TRI_BLOCK / 64) + (threadIdx.x >> 6)) & 31) == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); atomicAdd(&g_tri_phase[i], n_ - tt_); tt_ = n_; } } while (0)
#else
#define TSTAMP(i, dep) do { } while (0)
#endif

// info[j] = what the host keeps of job j: sizes and its first triangle / grid in the batch (the chunks' ranges follow one another in
// completion order; within a chunk the order is the reference's).  ctl: see MC_*.  `ids` holds ids_capacity entries.
template <int N>
__device__ __forceinline__ void mesh_count_body(const MapView &M, const int *__restrict__ ids, int ids_capacity, MeshJob *jobs, const int *__restrict__ n_jobs,
                                                JobInfo *info, int *ctl, unsigned *cnt, unsigned long long *job_acc, TriRec *tris, CubeCorners *corners,
                                                int part_capacity, unsigned *mesh_flag, int keep_dirty, int done_seq) {
    using G = MeshGeom<N>;
    __shared__ __attribute__((aligned(16))) float s_tile[G::TILE];
    __shared__ unsigned s_list[G::MAXC];  // occupied cubes of the sub-job in traversal order: cube | case << 10 | first triangle (relative) << 18
    __shared__ __attribute__((aligned(16))) unsigned char s_case[G::MAXC];  // case index per cube (0: no triangle)
    __shared__ int s_nb[27];
    __shared__ unsigned s_counts[64];  // the 256 vertex counts of the case table, four to a word (a per-lane index into constant memory is a global load)
    int lane = threadIdx.x;
    int n = *n_jobs;  // the job count stays on the device: the grid is persistent
    if (n > ids_capacity) n = ids_capacity;  // (the kept list never gets there: the host gives it up first)
    if (blockIdx.x == 0 && lane == 0) ctl[MC_JOBS] = n;  // where the kernels behind this one (and a second emission) read it
    // every integration launch queued before this recompute is over (pinned word [5]: the host's substitute for an event on the map's stream)
    if (blockIdx.x == 0 && lane == 2 && done_seq > 0) reinterpret_cast<volatile int *>(M.error_flag)[5] = done_seq;
    // the list of dirty slots has been consumed -- also when none of them became a job (every listed chunk removed since)
    if (blockIdx.x == 0 && lane == 1 && !keep_dirty) M.slot_dirty[2 * (size_t)M.max_chunks] = 0u;
    s_counts[lane] = reinterpret_cast<const unsigned *>(c_mc_counts)[lane];
    auto n_verts = [&](int index) -> int { return (int)((s_counts[index >> 2] >> ((index & 3) * 8)) & 0xffu); };
    unsigned long long *cursors = reinterpret_cast<unsigned long long *>(ctl + MC_CURSORS);
    // workgroups b and b + 8 share an XCD (observed dispatch order; speed only): the sub-jobs of a job run on one XCD, back to back --
    // the slabs' shared corner layers and the planes' rows meet in one L2
    const int xcd = (int)blockIdx.x & 7, stride = (int)gridDim.x >> 3;
    for (int q = (int)blockIdx.x >> 3;; q += stride) {
        const int jl = q / G::S, s = q - jl * G::S, j = jl * 8 + xcd;
        if (j >= n) break;
#ifdef CHISEL_PHASES
        unsigned long long mt_ = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && (blockIdx.x & 31) == 0) atomicAdd(&g_mesh_phase[7], 1ull);
#endif
        __syncthreads();  // the previous sub-job's LDS contents are no longer read
        // (a workgroup rarely takes a second unit; what the lanes derive from their number -- tile offsets of eight cubes for four box
        // shapes -- must not be hoisted in front of the loop, where it cost 150 registers and 250 bytes of scratch: opaque per iteration)
        asm volatile("" : "+v"(lane));
        const int jx = ids[3 * j], jy = ids[3 * j + 1], jz = ids[3 * j + 2];
        const int type = s < G::NI ? 0 : s - G::NI + 1;  // wave-uniform
        const unsigned plus_needed = type == 0 ? MeshBox<N, 0>::PLUS : (type == 1 ? MeshBox<N, 1>::PLUS : (type == 2 ? MeshBox<N, 2>::PLUS : MeshBox<N, 3>::PLUS));
        // The neighbourhood: sub-job 0 looks all 27 slots up (one lane each) and leaves the job record for the triangle kernel and the
        // host; the others only the chunks their corners lie in (an interior slab: the chunk itself).
        const int dx = lane % 3 - 1, dy = (lane / 3) % 3 - 1, dz = lane / 9 - 1;
        const bool plus = lane < 27 && dx >= 0 && dy >= 0 && dz >= 0;
        const bool counted = plus && ((plus_needed >> (dx + 2 * dy + 4 * dz)) & 1u);
        int slot = -1;
        unsigned sum = 0u;
        if (lane < 27 && (s == 0 || counted)) slot = hash_find_quiescent(M, jx + dx, jy + dy, jz + dz);
        if (counted && slot >= 0) sum = slot_summary(M)[slot];
        if (lane < 27) s_nb[lane] = slot;
        if (s == 0) {
            if (lane < 27) jobs[j].nb[lane] = slot;
            else if (lane == 27) {
                jobs[j].x = jx;
                jobs[j].y = jy;
                jobs[j].z = jz;
                jobs[j].pad[0] = jobs[j].pad[1] = 0;
            }
        }
        const int self = __shfl(slot, NB_SELF);
        const bool present = self >= 0;
        // Can any cube of this sub-job carry a triangle?  Corner 0 of every cube is a voxel of the chunk itself and must be observed
        // (ChunkManager.cpp:271 / :352), and a case other than 0 / 255 needs both signs among the cubes' corners, which lie in the chunks
        // counted above (slot_summary).  About half of the jobs of a recompute -- chunks inside the band but away from the surface --
        // stop here: no corners staged, nothing classified.
        const bool can = present && (unsigned)__shfl((int)sum, NB_SELF) != 0u && __ballot(counted && (sum & SUM_POS)) != 0ull && __ballot(counted && (sum & SUM_NEG)) != 0ull;
        if (s == 0 && lane == 28 && present && !keep_dirty) {
            // meshesToUpdate.clear() (Chisel.cpp:57) and the slot's "is in the job list" flag: this job's own
            if (mesh_flag) mesh_flag[self] = 0u;
            M.slot_dirty[self] = 0u;
        }
        MSTAMP(0);
        // lane 1 deposits the sub-job's figures: its entry of cnt, and the job's accumulator; the wave whose deposit is the job's last
        // allots the job its range.  (lane 0 takes the sub-job's stretch of the unordered lists in the same instruction.)
        auto deposit = [&](int my_nt, int my_ng, int part) -> unsigned long long {
            unsigned long long old = 0ull;
            if (lane == 1) cnt[(size_t)j * G::ROW + 2 + s] = (unsigned)my_nt | ((unsigned)my_ng << 16);
            if (lane == 1 || (lane == 0 && my_ng)) {
                unsigned long long *addr = lane == 0 ? cursors + part : job_acc + j;
                const unsigned long long inc = lane == 0 ? ((unsigned long long)my_ng << 32) | (unsigned long long)my_nt
                                                         : (1ull << 48) | ((unsigned long long)my_ng << 24) | (unsigned long long)my_nt;
                old = atomicAdd(addr, inc);
                if (lane == 1 && (int)(old >> 48) == G::S - 1) {
                    const unsigned long long tot = old + inc;
                    const int tnt = (int)(tot & 0xffffffull), tng = (int)((tot >> 24) & 0xffffffull);
                    unsigned long long base = 0ull;
                    if (tnt | tng) base = atomicAdd(reinterpret_cast<unsigned long long *>(ctl + MC_TRIS), ((unsigned long long)tng << 32) | (unsigned long long)tnt);
                    JobInfo ji;
                    ji.x = jx; ji.y = jy; ji.z = jz;
                    ji.present = present ? 1 : 0;
                    ji.n_vertices = 3 * tnt;
                    ji.n_grids = tng;
                    ji.tri_base = (int)(base & 0xffffffffull);
                    ji.grid_base = (int)(base >> 32);
                    info[j] = ji;
                    cnt[(size_t)j * G::ROW] = (unsigned)ji.tri_base;
                    cnt[(size_t)j * G::ROW + 1] = (unsigned)ji.grid_base;
                    job_acc[j] = 0ull;  // (every other sub-job of the job has been here: the accumulator is this wave's to reset)
                }
            }
            return old;
        };
        if (!can) {
            (void)deposit(0, 0, 0);
            continue;
        }
        auto run = [&](auto type_tag) {
            constexpr int T = decltype(type_tag)::value;
            using B = MeshBox<N, T>;
            const int z0 = T == 0 ? s * G::ZL : (T == 3 ? N - 1 : 0);
            const int bz = T == 0 ? min(G::ZL, N - 1 - z0) : (T == 3 ? 1 : N - 1);
            const int ncubes = B::BX * B::BY * bz;
            // ---- the box's corners into LDS, one float each (fold_corner), every load of a pass requested before the first is used
            if (T == 0) {
                // voxels [z0 N^2, (z0 + bz + 1) N^2) of the chunk itself: contiguous in the pool
                constexpr int PASS = N == 32 ? 4 : ((G::ZL + 1) * N * N / 4 + 63) / 64;  // 16^3: three 16-byte loads per lane and array
                const int n4 = (bz + 1) * N * N / 4;
                const float4 *sp = reinterpret_cast<const float4 *>(M.sdf + (size_t)self * (N * N * N) + (size_t)z0 * N * N);
                const float4 *wp = reinterpret_cast<const float4 *>(M.wgt + (size_t)self * (N * N * N) + (size_t)z0 * N * N);
                for (int base = 0; base < n4; base += 64 * PASS) {
                    float4 vs[PASS], vw[PASS];
#pragma unroll
                    for (int u = 0; u < PASS; u++) {
                        const int i = min(base + u * 64 + lane, n4 - 1);
                        vs[u] = sp[i];
                        vw[u] = wp[i];
                    }
#pragma unroll
                    for (int u = 0; u < PASS; u++) {
                        const int i = base + u * 64 + lane;
                        if (i < n4)
                            reinterpret_cast<float4 *>(s_tile)[i] = make_float4(fold_corner(vs[u].x, vw[u].x), fold_corner(vs[u].y, vw[u].y),
                                                                                fold_corner(vs[u].z, vw[u].z), fold_corner(vs[u].w, vw[u].w));
                    }
                }
            } else {
                constexpr int PASS = 5;
                const int nc = B::TX * B::TY * (bz + 1);
                for (int base = 0; base < nc; base += 64 * PASS) {
                    float2 v[PASS];
#pragma unroll
                    for (int u = 0; u < PASS; u++) {
                        const int i = min(base + u * 64 + lane, nc - 1);
                        v[u] = corner_voxel<N>(M, s_nb, B::X0 + i % B::TX, B::Y0 + (i / B::TX) % B::TY, z0 + i / (B::TX * B::TY));
                    }
#pragma unroll
                    for (int u = 0; u < PASS; u++) {
                        const int i = base + u * 64 + lane;
                        if (i < nc) s_tile[i] = fold_corner(v[u].x, v[u].y);
                    }
                }
            }
            __syncthreads();
            MSTAMP(1);
            // corner i of the cube whose corner 0 is tile entry t0: cubeIndexOffsets (ChunkManager.cpp:67-69)
            auto cube_corners = [&](int c, float (&sc)[8], int &index) -> bool {
                const int lx = c % B::BX, ly = (c / B::BX) % B::BY, lz = c / (B::BX * B::BY);
                const int t0 = (lz * B::TY + ly) * B::TX + lx;
                const int ox[8] = {0, 1, 1, 0, 0, 1, 1, 0}, oy[8] = {0, 0, 1, 1, 0, 0, 1, 1}, oz[8] = {0, 0, 0, 0, 1, 1, 1, 1};
                index = 0;
                bool observed = true;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float v = s_tile[t0 + ox[i] + oy[i] * B::TX + oz[i] * (B::TX * B::TY)];
                    observed = observed && (v == v);        // :271 / :352 "weight <= 0.5 -> not observed"
                    sc[i] = v;
                    index |= (v < 0.0f) ? (1 << i) : 0;     // MarchingCubes::CalculateVertexConfiguration MarchingCubes.h:108-118
                }
                return observed;
            };
            // Classification with neighbouring lanes on neighbouring cubes (cube k * 64 + lane: consecutive corners, no LDS bank
            // conflicts); the case bytes go through LDS to the lane that owns the cube in the traversal order (cubes lane * CPL ...).
#pragma unroll 2
            for (int k = 0; k < G::CPL; k++) {
                const int c = k * 64 + lane;
                if (c < ncubes) {
                    float sc[8];
                    int index;
                    unsigned char cs = 0;
                    if (cube_corners(c, sc, index)) cs = n_verts(index) ? (unsigned char)index : 0;
                    s_case[c] = cs;
                }
            }
            __syncthreads();
            int nv = 0, ng = 0;
            unsigned cw[G::CPL / 4];
#pragma unroll
            for (int w = 0; w < G::CPL / 4; w++) cw[w] = reinterpret_cast<const unsigned *>(s_case)[lane * (G::CPL / 4) + w];
#pragma unroll
            for (int k = 0; k < G::CPL; k++) {
                const int index = (lane * G::CPL + k < ncubes) ? (int)((cw[k >> 2] >> ((k & 3) * 8)) & 0xffu) : 0;
                const int c = n_verts(index);  // (case 0 has no vertices)
                nv += c;
                ng += (c != 0);  // IsOccupied (MarchingCubes.h:41-45)
            }
            MSTAMP(2);
            // wave scan of (vertices, occupied cubes) in traversal order
            int iv = nv, ig = ng;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int a = __shfl_up(iv, o), b = __shfl_up(ig, o);
                if (lane >= o) {
                    iv += a;
                    ig += b;
                }
            }
            const int tv = __shfl(iv, 63), tg = __shfl(ig, 63);
            MSTAMP(3);
            const int part = (j * G::S + s) & (MESH_PARTS - 1);
            const unsigned long long old = deposit(tv / 3, tg, part);
            if (tg == 0) return;
            const unsigned long long cur = (unsigned long long)__shfl((long long)old, 0);
            const int cur_t = (int)(cur & 0xffffffffull), cur_g = (int)(cur >> 32);
            MSTAMP(4);
            if (cur_t + tv / 3 > part_capacity || cur_g + tg > part_capacity) {  // wave-uniform
                if (lane == 0) ctl[MC_OVERFLOW] = 1;  // (the host grows both lists and runs the recompute's kernels again)
                return;
            }
            // The occupied cubes (a few dozen of the sub-job's, unevenly spread over the lanes) are listed in LDS and emitted by all
            // lanes together, one cube each.
            {
                int trel = (iv - nv) / 3, gidx = ig - ng;
#pragma unroll
                for (int k = 0; k < G::CPL; k++) {
                    const int c = lane * G::CPL + k;
                    const int index = (c < ncubes) ? (int)((cw[k >> 2] >> ((k & 3) * 8)) & 0xffu) : 0;
                    if (index == 0) continue;
                    s_list[gidx] = (unsigned)c | ((unsigned)index << 10) | ((unsigned)trel << 18);
                    trel += n_verts(index) / 3;
                    gidx++;
                }
            }
            __syncthreads();
            const size_t base_t = (size_t)part * part_capacity + cur_t, base_g = (size_t)part * part_capacity + cur_g;
            for (int e = lane; e < tg; e += 64) {
                const unsigned a = s_list[e];
                const int c = (int)(a & 1023u), index = (int)((a >> 10) & 255u), trel = (int)(a >> 18);
                CubeCorners cc;
                int idx2;
                (void)cube_corners(c, cc.s, idx2);
                corners[base_g + e] = cc;
                const int x = B::X0 + c % B::BX, y = B::Y0 + (c / B::BX) % B::BY, z = z0 + c / (B::BX * B::BY);
                const int nt = n_verts(index) / 3;
                TriRec rec;
                rec.job = (unsigned)j;
                rec.cube = (unsigned)(base_g + e);
                for (int t = 0; t < nt; t++) {
                    rec.code = (unsigned)x | ((unsigned)y << 5) | ((unsigned)z << 10) | ((unsigned)index << 15) | ((unsigned)t << 23) | ((unsigned)s << 26);
                    rec.rel = (unsigned)(trel + t) | ((unsigned)e << 16);
                    tris[base_t + trel + t] = rec;
                }
            }
            MSTAMP(5);
        };
        switch (type) {
            case 0: run(std::integral_constant<int, 0>()); break;
            case 1: run(std::integral_constant<int, 1>()); break;
            case 2: run(std::integral_constant<int, 2>()); break;
            default: run(std::integral_constant<int, 3>()); break;
        }
    }
}

// (the register cap is an attribute that takes a literal: one kernel per chunk size around the common body)
#define CHISEL_MESH_COUNT_KERNEL(NN, VGPRS, WAVES)                                                                                                    \
    __global__ __launch_bounds__(64, WAVES) __attribute__((amdgpu_num_vgpr(VGPRS))) void mesh_count_kernel_##NN(                                 \
        MapView M, const int *__restrict__ ids, int ids_capacity, MeshJob *jobs, const int *__restrict__ n_jobs, JobInfo *info, int *ctl, unsigned *cnt, \
        unsigned long long *job_acc, TriRec *tris, CubeCorners *corners, int part_capacity, unsigned *mesh_flag, int keep_dirty, int done_seq) { \
        mesh_count_body<NN>(M, ids, ids_capacity, jobs, n_jobs, info, ctl, cnt, job_acc, tris, corners, part_capacity, mesh_flag, keep_dirty, done_seq); \
    }
CHISEL_MESH_COUNT_KERNEL(8, MESH_COUNT_VGPRS, MESH_COUNT_WAVES)
CHISEL_MESH_COUNT_KERNEL(16, MESH_COUNT_VGPRS, MESH_COUNT_WAVES)
CHISEL_MESH_COUNT_KERNEL(32, 128, 3)
#undef CHISEL_MESH_COUNT_KERNEL

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
__global__ __launch_bounds__(MESH_TRI_BLOCK, MESH_TRI_WAVES) void mesh_triangle_kernel(MapView M, MeshParams P, const MeshJob *__restrict__ jobs,
                                                                    const JobInfo *__restrict__ info, const TriRec *__restrict__ tris,
                                                                    const CubeCorners *__restrict__ corners, const int *__restrict__ totals,
                                                                    const unsigned *__restrict__ cnt, int part_capacity, float *arena, size_t arena_floats, int *host_info,
                                                                    volatile int *host_flags, int max_jobs, int seq, int publish) {
    // The unordered triangle list: MESH_PARTS partitions of part_capacity entries, each filled up to its cursor.  Entry i of their
    // concatenation is this grid's i-th triangle: where the partitions begin in that numbering, once per workgroup.  (A partition per
    // workgroup -- no table, no search -- was 1-7 us slower: the workgroups that find nothing to do then sit between the others in dispatch order.)
    __shared__ int s_off[MESH_PARTS + 1];
#ifdef CHISEL_PHASES
    unsigned long long tt_ = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0 && ((blockIdx.x * (MESH_TRI_BLOCK / 64) + (threadIdx.x >> 6)) & 31) == 0) atomicAdd(&g_tri_phase[7], 1ull);
#endif
    if (threadIdx.x < MESH_PARTS) {
        int v = totals[MC_CURSORS + 2 * threadIdx.x];
#pragma unroll
        for (int o = 1; o < MESH_PARTS; o <<= 1) {
            const int a = __shfl_up(v, o);
            if ((int)threadIdx.x >= o) v += a;
        }
        s_off[threadIdx.x + 1] = v;
        if (threadIdx.x == 0) s_off[0] = 0;
    }
    __syncthreads();
    const int part_tris = s_off[MESH_PARTS];
    const int n_tris = totals[MC_TRIS];  // (= the sum of the cursors unless a partition overflowed)
    const int n_jobs = totals[MC_JOBS];
    if ((publish & 1) && blockIdx.x == 0 && threadIdx.x == 0) {  // (publish: bit 0 = totals to the host, bit 1 = the kept job list was this recompute's input)
        // The recompute's totals, straight into pinned host memory as ONE 16-byte store -- {triangles, grids, jobs | overflow << 31,
        // sequence number} -- before anything else: the host polls word 3 for this recompute's sequence number when the caller next
        // touches the map.  No copy engine, no event, no stream wait and no kernel of its own in between (an event record on the
        // map's stream costs a barrier packet of 7-12 us in front of the next kernel, a one-thread kernel 5 us).
        uint4 v;
        v.x = (unsigned)n_tris;
        v.y = (unsigned)totals[MC_GRIDS];
        v.z = (unsigned)n_jobs | (totals[MC_OVERFLOW] ? 0x80000000u : 0u);
        v.w = (unsigned)seq;
        *reinterpret_cast<uint4 *>(const_cast<int *>(host_flags)) = v;
        // what the host polls is a second copy of the sequence number behind a system-scope fence: that the 16 bytes above arrive
        // as one piece is how the bus behaves, not a guarantee (this thread alone pays the few hundred nanoseconds)
        host_flags[4] = (totals[MC_OVERFLOW] == 0 && (size_t)n_tris * 9 * (P.use_color ? 3 : 2) + (size_t)totals[MC_GRIDS] * 3 <= arena_floats) ? 0 : 1;  // (`fits`, below)
        __threadfence_system();
        host_flags[5] = seq;
        // the job list the integration kernels keep has been consumed by the count kernel (its number is in totals[3]): empty again
        if ((publish & 2) && M.mesh_ctl) M.mesh_ctl[MC_KEPT] = 0;
    }
    const size_t nv3 = (size_t)n_tris * 9, ng3 = (size_t)totals[MC_GRIDS] * 3;
    // a triangle list that overflowed (totals[2]) is incomplete: nothing is emitted, the host lists and emits again
    const bool fits = totals[MC_OVERFLOW] == 0 && nv3 * (P.use_color ? 3 : 2) + ng3 <= arena_floats;
    // ... and an integration launch the host has queued behind this recompute without waiting for its totals must not touch the map
    if ((publish & 1) && !fits && blockIdx.x == 0 && threadIdx.x == 0 && M.mesh_ctl) M.mesh_ctl[MC_LATCH] = 1;
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
    constexpr int ROW = MeshGeom<N>::ROW;
    for (int ti = (int)blockIdx.x * MESH_TRI_BLOCK + threadIdx.x; ti < 3 * part_tris; ti += (int)gridDim.x * MESH_TRI_BLOCK) {
    int i = ti / 3;
    const int mine = ti - 3 * i;
    int part = 0;
#pragma unroll
    for (int o = MESH_PARTS / 2; o > 0; o >>= 1)
        if (s_off[part + o] <= i) part += o;
    i -= s_off[part];
    TSTAMP(0, i);
    const TriRec rec = tris[(size_t)part * part_capacity + i];
    const MeshJob &job = jobs[rec.job];  // stays in memory (L1 / L2): its neighbour table is indexed per lane
    const int *nb = job.nb;
    const int jx = job.x, jy = job.y, jz = job.z;
    const int x = (int)(rec.code & 31u), y = (int)((rec.code >> 5) & 31u), z = (int)((rec.code >> 10) & 31u);
    const int index = (int)((rec.code >> 15) & 0xffu), t = 3 * (int)((rec.code >> 23) & 7u), sub = (int)(rec.code >> 26);
    // where the triangle goes: its job's base + the triangles of the sub-jobs in front of its own + its place in the sub-job
    // (mesh_count_kernel: the traversal order of ChunkManager.cpp:395-441); likewise the cube's grid entry.  The job's row of `cnt`
    // -- bases, then the sub-jobs' figures -- arrives as 16-byte loads issued together.
    int tri_pos = (int)(rec.rel & 0xffffu), grid_pos = (int)(rec.rel >> 16);
    {
        const uint4 *row4 = reinterpret_cast<const uint4 *>(cnt + (size_t)rec.job * ROW);
        uint4 r[ROW / 4];
#pragma unroll
        for (int q = 0; q < ROW / 4; q++) r[q] = row4[q];
        tri_pos += (int)r[0].x;
        grid_pos += (int)r[0].y;
#pragma unroll
        for (int q = 2; q < ROW; q++) {
            const unsigned c = (q & 3) == 0 ? r[q >> 2].x : ((q & 3) == 1 ? r[q >> 2].y : ((q & 3) == 2 ? r[q >> 2].z : r[q >> 2].w));
            const unsigned cc = (q - 2 < sub) ? c : 0u;
            tri_pos += (int)(cc & 0xffffu);
            grid_pos += (int)(cc >> 16);
        }
    }
    const int vi = 3 * tri_pos + mine;
    TSTAMP(1, vi);
    const unsigned long long row = c_mc_cases[index];
    const f3v origin = mk3((float)(N * jx) * P.res, (float)(N * jy) * P.res, (float)(N * jz) * P.res);  // Chunk.cpp:43
    // cube origin = centroid of voxel (x, y, z) + chunk origin (ChunkManager.cpp:61, :404)
    const f3v coords = add3(mk3((float)x * P.res + P.half_res, (float)y * P.res + P.half_res, (float)z * P.res + P.half_res), origin);
    const CubeCorners cc = corners[rec.cube];
    if (t == 0 && mine == 0) {
        const size_t g = (size_t)grid_pos;
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
    TSTAMP(2, pv.x);
    f3v nrm = fn;
    double dist;
    f3v grad;
    if ((P.stages & 1) && get_sdf_and_gradient<N>(M, P, pv, jx, jy, jz, nb, dist, grad)) {
        const float mag = sqrtf(sum3f(grad.x * grad.x, grad.y * grad.y, grad.z * grad.z));
        if ((double)mag > 1e-12) nrm = scl3(grad, 1.0f / mag);
    }
    TSTAMP(3, nrm.x);
    no[0] = nrm.x;
    no[1] = nrm.y;
    no[2] = nrm.z;
    if (colors) {
        const f3v col = (P.stages & 2) ? interpolate_color<N>(M, P, pv, jx, jy, jz, nb) : mk3(0.0f, 0.0f, 0.0f);
        float *co = colors + 3 * (size_t)vi;
        TSTAMP(4, col.x);
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
    int *totals = nullptr;   // = MapView::mesh_ctl (MC_*): triangles, grids, record-list overflow flag, jobs; the kept job list's length; the partition cursors
    int *n_jobs = nullptr;   // where the count kernel finds the number of jobs (the kept list's length, or null: totals[3] and the caller's ids)
    const int *ext_ids = nullptr;  // a job list on the device that is neither the kept one nor `ids` (the sharded recompute's plan: chisel_hip_update_meshes_planned) ...
    int *ext_n = nullptr;          // ... its length (device) ...
    int ext_capacity = 0;          // ... and its capacity in ids
    TriRec *tris = nullptr;  // [tri_capacity] triangle list of one recompute: MESH_PARTS partitions of tri_capacity / MESH_PARTS entries
    CubeCorners *corners = nullptr;  // [tri_capacity] corner distances of its occupied cubes, partitioned likewise
    int tri_capacity = 0;
    unsigned *cnt = nullptr;             // [capacity][S] triangles | occupied cubes << 16 of every sub-job (mesh_count_kernel)
    unsigned long long *job_acc = nullptr;  // [capacity] per-job accumulators of the count kernel's waves; zero between recomputes
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
    if (b.cnt) (void)hipFree(b.cnt);
    if (b.job_acc) (void)hipFree(b.job_acc);
    if (b.flags) (void)hipFree(b.flags);
    if (b.query) (void)hipFree(b.query);
    if (b.cube) (void)hipFree(b.cube);
    b = MeshBuffers();
}

}  // namespace chisel_hip
