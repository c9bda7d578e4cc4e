// kernels_map.h -- chunk-hash / pool maintenance and small query kernels.
//
// Device-side counterpart of ChunkManager's unordered_map bookkeeping (ChunkManager.h:67-122):
// CreateChunk happens inside integrate_kernel; everything else (Reset, RemoveChunk / GarbageCollect,
// HasChunk / GetChunk lookups, GetChunks enumeration) lives here.
#pragma once
#include "chisel_device.h"

namespace chisel_hip {

// default voxels of V-voxel chunk `slot` (DistVoxel() DistVoxel.cpp:27-31, ColorVoxel() ColorVoxel.cpp:27-31);
// called by every thread of a workgroup
__device__ inline void fill_default_chunk(const MapView &M, int slot, int V) {
    float4 *s4 = reinterpret_cast<float4 *>(M.sdf + (size_t)slot * V);
    float4 *w4 = reinterpret_cast<float4 *>(M.wgt + (size_t)slot * V);
    uint4 *c4 = M.rgbw ? reinterpret_cast<uint4 *>(M.rgbw + (size_t)slot * V) : nullptr;
    for (int i = threadIdx.x; i < V / 4; i += blockDim.x) {
        s4[i] = make_float4(99999.0f, 99999.0f, 99999.0f, 99999.0f);
        w4[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (c4) c4[i] = make_uint4(0u, 0u, 0u, 0u);
    }
}

// Chisel::Reset / ChunkManager::Reset (Chisel.cpp:44-48, ChunkManager.cpp:176-180) and the initial state: empty hash, full free list,
// default voxels in every slot (the pool invariant of chisel_device.h).  A reset restores only the slots that hold a chunk -- free slots
// hold default voxels already -- so its cost follows what is resident, not the size of the pool (all_slots = 0; 1 at creation, when
// the pool's memory is fresh: 1.04 ms and 6.3 GB written per /Chisel/Reset before, for 48 MB of resident chunks).  Slots are dealt to
// the workgroups in a stride (slot s belongs to workgroup s mod gridDim: the resident slots are the low-numbered ones, popped first),
// 256 at a time: the keys of a batch are read together, the occupied slots of the batch are then restored by the whole workgroup.
__global__ __launch_bounds__(256) void reset_map_kernel(MapView M, int V, int all_slots) {
    __shared__ int s_list[256];
    __shared__ int s_n;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = gid; i <= M.hash_mask; i += stride) M.hash_keys[i] = KEY_EMPTY;
    // (only the committed slots have voxel memory, hold chunks and sit on the free list: MapView::committed)
    for (uint64_t i = gid; i < (uint64_t)M.committed; i += stride) {
        M.slot_dirty[i] = 0;
        slot_summary(M)[i] = 0;
        if (M.mesh_flag) M.mesh_flag[i] = 0;
        M.free_list[i] = M.committed - 1 - (int)i;  // slot 0 is popped first
    }
    for (long long base = 0; base < (long long)M.committed; base += (long long)gridDim.x * 256) {
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        const long long slot = base + (long long)threadIdx.x * gridDim.x + blockIdx.x;
        if (slot < (long long)M.committed) {
            const bool used = all_slots || M.slot_key[slot] != KEY_EMPTY;
            M.slot_key[slot] = KEY_EMPTY;
            if (used) s_list[atomicAdd(&s_n, 1)] = (int)slot;
        }
        __syncthreads();
        const int n = s_n;
        for (int i = 0; i < n; i++) fill_default_chunk(M, s_list[i], V);
        __syncthreads();
    }
    if (gid == 0) {
        M.slot_dirty[2 * (size_t)M.max_chunks] = 0;  // the list of dirty slots is empty
        if (M.mesh_ctl) {
            M.mesh_ctl[4] = 0;                       // and so is the job list
            M.mesh_ctl[MC_LATCH] = 0;                // (a recompute that did not fit: void with the map)
            for (int a = 0; a < 3; a++) {            // the box of created ids: empty
                M.mesh_ctl[MC_BBOX + a] = INT32_MAX;
                M.mesh_ctl[MC_BBOX + 3 + a] = INT32_MIN;
            }
        }
        *M.free_top = M.committed;
        M.error_flag[0] = 0;
        M.error_flag[1] = 0;
    }
}

// A growable pool has just committed the slots [first, first + n) (chisel_hip.hip: grow_pool; M.committed is still `first`): default voxels
// into their fresh memory (the pool invariant), their per-slot state cleared, and onto the free list -- the highest on top, as after a reset.
// On the map's stream: no other kernel that pops or pushes slots runs beside it.
__global__ __launch_bounds__(256) void grow_pool_kernel(MapView M, int V, int first, int n) {
    for (int s = blockIdx.x; s < n; s += gridDim.x) {
        const int slot = first + s;
        fill_default_chunk(M, slot, V);
        if (threadIdx.x == 0) {
            M.slot_key[slot] = KEY_EMPTY;
            M.slot_dirty[slot] = 0;
            slot_summary(M)[slot] = 0;
            if (M.mesh_flag) M.mesh_flag[slot] = 0;
            M.free_list[*M.free_top + (n - 1 - s)] = slot;  // (free_top is raised by grow_commit_kernel, behind this kernel: stable here)
        }
    }
}
// ... and then, behind it on the stream, the free list's top (one thread)
__global__ void grow_commit_kernel(MapView M, int n) { *M.free_top += n; }

__device__ inline int hash_find(const MapView &M, int x, int y, int z, uint64_t *where = nullptr) {
    const uint64_t key = pack_id(x, y, z);
    const uint64_t h = chunk_hash(x, y, z) & M.hash_mask;
    for (uint64_t i = 0; i <= M.hash_mask; i++) {
        const uint64_t idx = (h + i) & M.hash_mask;
        const uint64_t k = M.hash_keys[idx];
        if (k == key) {
            if (where) *where = idx;
            return M.hash_vals[idx];
        }
        if (k == KEY_EMPTY) break;
    }
    return -1;
}

// The same for kernels that run while nothing inserts into the hash (the mesh kernels: the map's stream is theirs): key and
// slot of the home bucket are requested together -- one round trip instead of two for the chunks that sit in their home bucket.
__device__ inline int hash_find_quiescent(const MapView &M, int x, int y, int z) {
    const uint64_t key = pack_id(x, y, z);
    const uint64_t h = chunk_hash(x, y, z) & M.hash_mask;
    const uint64_t k0 = M.hash_keys[h];
    const int v0 = M.hash_vals[h];
    if (k0 == key) return v0;
    if (k0 == KEY_EMPTY) return -1;
    for (uint64_t i = 1; i <= M.hash_mask; i++) {
        const uint64_t idx = (h + i) & M.hash_mask;
        const uint64_t k = M.hash_keys[idx];
        if (k == key) return M.hash_vals[idx];
        if (k == KEY_EMPTY) break;
    }
    return -1;
}

// ---- meshesToUpdate on the device: the job list of the next mesh recompute (kernels_mesh.h) -----------------------------------------
// one resident chunk becomes a job: the caller whose exchange sets the slot's flag first appends its id
__device__ inline void mesh_append_job(const MapView &M, unsigned *mesh_flag, int slot, int *ids, int *n_jobs) {
    if (atomicExch(&mesh_flag[slot], 1u) != 0u) return;
    const uint64_t key = M.slot_key[slot];
    if (key == KEY_EMPTY) return;
    const int pos = atomicAdd(n_jobs, 1);
    if (pos >= M.mesh_jobs_capacity) return;  // (the host rebuilds the list from the dirty flags long before this: host_mesh.h)
    int x, y, z;
    unpack_id(key, x, y, z);
    ids[3 * pos] = x;
    ids[3 * pos + 1] = y;
    ids[3 * pos + 2] = z;
}
// The same expansion at the moment a slot is first dirtied, by the wave that dirtied it (integrate_kernel's epilogue: the wave has
// nothing else left to do): lanes 0-26 look the neighbourhood up and append what is resident.  The chunk hash may be receiving
// insertions from other waves of this launch: a hit only counts if slot_key confirms it (create_chunk writes key, value, slot_key in
// that order), and a neighbour that is missed because it is being created right now is dirty itself and lists itself.  Everything
// here is an accelerator for mesh_mark_kernel, which can rebuild the list from the dirty flags at any time.
// (out of line, the map view read from device memory: the integration kernel's registers are spoken for)
__device__ __attribute__((noinline)) void mesh_expand_dirty(const MapView *__restrict__ Mc, int slot, int x, int y, int z, int lane) {
    const MapView M = *Mc;
    if (!M.mesh_flag || lane >= 27) return;
    int ns = slot;
    if (lane != 13) {
        const int nx = x + lane % 3 - 1, ny = y + (lane / 3) % 3 - 1, nz = z + lane / 9 - 1;
        ns = hash_find(M, nx, ny, nz);
        if (ns >= 0 && (ns >= M.max_chunks || M.slot_key[ns] != pack_id(nx, ny, nz))) ns = -1;
    }
    if (ns >= 0) mesh_append_job(M, M.mesh_flag, ns, M.mesh_jobs, M.mesh_ctl + 4);
}

// HasChunk / GetChunk (ChunkManager.h:79-87): slot per id, -1 when absent
__global__ void lookup_kernel(MapView M, const int *ids, int n, int *slots) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) slots[i] = hash_find(M, ids[3 * i], ids[3 * i + 1], ids[3 * i + 2]);
}

// RemoveChunk(ChunkID) (ChunkManager.h:99-108) for a list: Chisel::GarbageCollect (Chisel.cpp:61-67).
// One workgroup per id: thread 0 unlinks the chunk, then the whole group restores default voxels in the
// freed slot (pool invariant).  Never runs concurrently with integrate_kernel (same stream).
__global__ __launch_bounds__(256) void remove_chunks_kernel(MapView M, const int *ids, int n, int *n_removed, int V) {
    __shared__ int s_slot;
    const int i = blockIdx.x;
    if (threadIdx.x == 0) {
        s_slot = -1;
        uint64_t where = 0;
        const int slot = hash_find(M, ids[3 * i], ids[3 * i + 1], ids[3 * i + 2], &where);
        if (slot >= 0) {
            // duplicates in the list: only the group that swaps the key out frees the slot
            const uint64_t key = pack_id(ids[3 * i], ids[3 * i + 1], ids[3 * i + 2]);
            if (atomicCAS((unsigned long long *)&M.hash_keys[where], (unsigned long long)key, (unsigned long long)KEY_TOMB) == key) {
                M.slot_key[slot] = KEY_EMPTY;
                M.slot_dirty[slot] = 0;
                slot_summary(M)[slot] = 0;
                if (M.mesh_flag) M.mesh_flag[slot] = 0;  // (its entry of the job list, if any, names an id that is now absent: skipped by the count kernel)
                s_slot = slot;
            }
        }
    }
    __syncthreads();
    const int slot = s_slot;
    if (slot < 0) return;
    fill_default_chunk(M, slot, V);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const int pos = atomicAdd(M.free_top, 1);
        M.free_list[pos] = slot;
        atomicAdd(n_removed, 1);
    }
}

// create-or-find for chisel_hip_upload_chunk (ChunkManager::AddChunk ChunkManager.h:89-92); one thread
__global__ void ensure_chunk_kernel(MapView M, int x, int y, int z, int *out_slot) {
    int slot = hash_find(M, x, y, z);
    if (slot < 0) {
        int top = atomicSub(M.free_top, 1) - 1;
        if (top < 0) {
            atomicAdd(M.free_top, 1);
            raise_error(M.error_flag, 1);
        } else {
            slot = M.free_list[top];
            const uint64_t key = pack_id(x, y, z);
            const uint64_t h = chunk_hash(x, y, z) & M.hash_mask;
            bool placed = false;
            for (uint64_t i = 0; i <= M.hash_mask && !placed; i++) {
                const uint64_t idx = (h + i) & M.hash_mask;
                const uint64_t cur = M.hash_keys[idx];
                if (cur == KEY_EMPTY || cur == KEY_TOMB) {
                    M.hash_keys[idx] = key;
                    M.hash_vals[idx] = slot;
                    placed = true;
                }
            }
            if (placed) {
                M.slot_key[slot] = key;
                bbox_include(M.mesh_ctl, x, y, z);
            } else {
                raise_error(M.error_flag, 2);
                slot = -1;
            }
        }
    }
    *out_slot = slot;
}

// Chunk::ComputeStatistics (Chunk.cpp:89-116) over every resident chunk, ChunkManager::PrintMemoryStatistics' sums (ChunkManager.cpp:641-678):
// one workgroup per slot (grid-stride), 16-byte loads, wave shuffles, one atomic per workgroup, slot and quantity.
// out: [0] unknown, [1] inside, [2] outside, [3] chunks (64-bit counts), then the weight sum (double), then id min[3] / max[3] (int).
struct CensusOut {
    unsigned long long unknown, inside, outside, chunks;
    double weight;
    int id_min[3], id_max[3];
};
__global__ __launch_bounds__(256) void census_kernel(MapView M, int V, CensusOut *out) {
    __shared__ unsigned s_cnt[4][3];
    __shared__ double s_w[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int slot = blockIdx.x; slot < M.committed; slot += gridDim.x) {
        const uint64_t key = M.slot_key[slot];  // block-uniform
        if (key == KEY_EMPTY) continue;
        unsigned unknown = 0, inside = 0, outside = 0;
        double wsum = 0.0;
        const float4 *sp = reinterpret_cast<const float4 *>(M.sdf + (size_t)slot * V), *wp = reinterpret_cast<const float4 *>(M.wgt + (size_t)slot * V);
        for (int q = threadIdx.x; q < V / 4; q += 256) {
            const float4 s4 = sp[q], w4 = wp[q];
            const float sv[4] = {s4.x, s4.y, s4.z, s4.w}, wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (wv[i] > 0) {
                    if (sv[i] < 0) inside++;
                    else outside++;
                } else {
                    unknown++;
                }
                wsum += (double)wv[i];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            unknown += __shfl_down(unknown, o);
            inside += __shfl_down(inside, o);
            outside += __shfl_down(outside, o);
            wsum += __shfl_down(wsum, o);
        }
        __syncthreads();
        if (lane == 0) {
            s_cnt[wave][0] = unknown; s_cnt[wave][1] = inside; s_cnt[wave][2] = outside;
            s_w[wave] = wsum;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(&out->unknown, (unsigned long long)(s_cnt[0][0] + s_cnt[1][0] + s_cnt[2][0] + s_cnt[3][0]));
            atomicAdd(&out->inside, (unsigned long long)(s_cnt[0][1] + s_cnt[1][1] + s_cnt[2][1] + s_cnt[3][1]));
            atomicAdd(&out->outside, (unsigned long long)(s_cnt[0][2] + s_cnt[1][2] + s_cnt[2][2] + s_cnt[3][2]));
            atomicAdd(&out->chunks, 1ull);
            atomicAdd(&out->weight, (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]));
            int x, y, z;
            unpack_id(key, x, y, z);
            atomicMin(&out->id_min[0], x); atomicMin(&out->id_min[1], y); atomicMin(&out->id_min[2], z);
            atomicMax(&out->id_max[0], x); atomicMax(&out->id_max[1], y); atomicMax(&out->id_max[2], z);
        }
    }
}

// enumerate resident chunks (GetChunks()) or the dirty ones: ballot compaction over the slot table
template <bool DIRTY_ONLY>
__global__ void list_slots_kernel(MapView M, int *ids, int *slots, int max_out, int *count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool keep = false;
    uint64_t key = KEY_EMPTY;
    if (i < M.committed) {  // (slots without voxel memory hold no chunk)
        key = M.slot_key[i];
        keep = key != KEY_EMPTY && (!DIRTY_ONLY || M.slot_dirty[i] != 0);
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
        if (pos < max_out) {
            int x, y, z;
            unpack_id(key, x, y, z);
            ids[3 * pos] = x;
            ids[3 * pos + 1] = y;
            ids[3 * pos + 2] = z;
            if (slots) slots[pos] = i;
        }
    }
}

// ---- whole-chunk export / import (halo exchange for meshing a sharded map; batched download / upload) ---------
// one workgroup per listed chunk: its voxel arrays -> row j of the output arrays (defaults when the chunk is absent)
__global__ __launch_bounds__(256) void export_chunks_kernel(MapView M, const int *ids, int V, float *sdf, float *wgt, uchar4 *rgbw, int *found) {
    __shared__ int s_slot;
    const int j = blockIdx.x;
    if (threadIdx.x == 0) {
        s_slot = hash_find(M, ids[3 * j], ids[3 * j + 1], ids[3 * j + 2]);
        found[j] = s_slot >= 0 ? 1 : 0;
    }
    __syncthreads();
    const int slot = s_slot;
    const size_t src = (size_t)(slot >= 0 ? slot : 0) * V, dst = (size_t)j * V;
    for (int v = threadIdx.x; v < V; v += 256) {
        sdf[dst + v] = slot >= 0 ? M.sdf[src + v] : 99999.0f;
        wgt[dst + v] = slot >= 0 ? M.wgt[src + v] : 0.0f;
        if (rgbw) rgbw[dst + v] = (slot >= 0 && M.rgbw) ? M.rgbw[src + v] : make_uchar4(0, 0, 0, 0);
    }
}
// one workgroup per listed chunk with take[j] != 0: create-or-find (thread 0; serialised over the list by the caller's
// choice of one launch per call -- the hash insert below is the single-writer form of ensure_chunk_kernel, so the
// kernel runs with ONE workgroup at a time per id; distinct ids never collide on a slot because free_top is atomic and
// the key is claimed with a CAS), then the row's voxels are written into the chunk
__global__ __launch_bounds__(256) void import_chunks_kernel(MapView M, const int *ids, const int *take, int V, const float *sdf, const float *wgt,
                                                             const uchar4 *rgbw) {
    __shared__ int s_slot;
    const int j = blockIdx.x;
    if (take && !take[j]) return;
    if (threadIdx.x == 0) {
        const int x = ids[3 * j], y = ids[3 * j + 1], z = ids[3 * j + 2];
        int slot = hash_find(M, x, y, z);
        if (slot < 0) {
            const int top = atomicSub(M.free_top, 1) - 1;
            if (top < 0) {
                atomicAdd(M.free_top, 1);
                raise_error(M.error_flag, 1);
            } else {
                slot = M.free_list[top];
                const uint64_t key = pack_id(x, y, z);
                const uint64_t h = chunk_hash(x, y, z) & M.hash_mask;
                bool placed = false;
                for (uint64_t i = 0; i <= M.hash_mask && !placed; i++) {
                    const uint64_t idx = (h + i) & M.hash_mask;
                    const uint64_t cur = M.hash_keys[idx];
                    if ((cur == KEY_EMPTY || cur == KEY_TOMB) &&
                        atomicCAS((unsigned long long *)&M.hash_keys[idx], (unsigned long long)cur, (unsigned long long)key) == cur) {
                        M.hash_vals[idx] = slot;
                        placed = true;
                    }
                }
                if (placed) {
                    M.slot_key[slot] = key;
                    bbox_include(M.mesh_ctl, x, y, z);
                } else {
                    raise_error(M.error_flag, 2);
                    slot = -1;
                }
            }
        }
        s_slot = slot;
    }
    __syncthreads();
    const int slot = s_slot;
    if (slot < 0) return;
    const size_t dst = (size_t)slot * V, src = (size_t)j * V;
    if (threadIdx.x == 0) slot_summary(M)[slot] = SUM_ANY;  // (voxels from outside: anything)
    for (int v = threadIdx.x; v < V; v += 256) {
        M.sdf[dst + v] = sdf[src + v];
        M.wgt[dst + v] = wgt[src + v];
        if (M.rgbw) M.rgbw[dst + v] = rgbw ? rgbw[src + v] : make_uchar4(0, 0, 0, 0);
    }
}
// ---- meshing a sharded map with shells instead of whole ghost chunks -------------------------------------------------------------
// What a job chunk J reads of a neighbour G = J + d (d in {-1, 0, 1}^3): cube corners one voxel into the "+" neighbours, gradients
// one voxel around the voxel that holds a vertex (vertices lie between the centres of voxel 0 and of the "+" neighbour's voxel 0),
// the nearest voxel's colour: along an axis with d = +1 the coordinates {0, 1} of G, with d = -1 the coordinate {N - 1}, with d = 0 all
// of them.  A box code holds that choice per axis (2 bits each: 0 = all, 1 = {0, 1}, 2 = {N - 1}, 3 = {0, 1, N - 1}); the requests
// of several jobs for one ghost merge per axis (both ends -> 3, anything with all -> all).  The payload of an item is its box in z, y, x order; the host computes the items' offsets
// from the codes alone, on both sides of the exchange.
__host__ __device__ inline int shell_len(int code, int N) { return code == 0 ? N : (code == 1 ? 2 : (code == 2 ? 1 : 3)); }
__host__ __device__ inline int shell_coord(int code, int i, int N) { return code == 2 ? N - 1 : ((code == 3 && i == 2) ? N - 1 : i); }
__host__ __device__ inline long long shell_volume(int box, int N) {
    return (long long)shell_len(box & 3, N) * shell_len((box >> 2) & 3, N) * shell_len((box >> 4) & 3, N);
}
// items: (x, y, z, box) per entry; offs: first payload voxel of every item
__global__ __launch_bounds__(256) void export_shells_kernel(MapView M, const int *items, const long long *offs, int N, float *sdf, float *wgt, uchar4 *rgbw,
                                                             int *found) {
    __shared__ int s_slot;
    const int j = blockIdx.x;
    if (threadIdx.x == 0) {
        s_slot = hash_find(M, items[4 * j], items[4 * j + 1], items[4 * j + 2]);
        found[j] = s_slot >= 0 ? 1 : 0;
    }
    __syncthreads();
    const int slot = s_slot, box = items[4 * j + 3];
    const int cx = box & 3, cy = (box >> 2) & 3, cz = (box >> 4) & 3;
    const int lx = shell_len(cx, N), ly = shell_len(cy, N), lz = shell_len(cz, N);
    const long long base = offs[j];
    const size_t src = (size_t)(slot >= 0 ? slot : 0) * N * N * N;
    for (int v = threadIdx.x; v < lx * ly * lz; v += 256) {
        const int x = shell_coord(cx, v % lx, N), y = shell_coord(cy, (v / lx) % ly, N), z = shell_coord(cz, v / (lx * ly), N);
        const size_t i = src + (size_t)(z * N + y) * N + x;
        sdf[base + v] = slot >= 0 ? M.sdf[i] : 99999.0f;
        wgt[base + v] = slot >= 0 ? M.wgt[i] : 0.0f;
        if (rgbw) rgbw[base + v] = (slot >= 0 && M.rgbw) ? M.rgbw[i] : make_uchar4(0, 0, 0, 0);
    }
}
// Installing shells, phase 1: one workgroup per DISTINCT ghost id (first[k] = an item of that ghost; its `found` decides): create the
// chunk unless it is there -- a fresh slot holds default voxels (pool invariant).  Distinct ids never collide on a slot: free_top is
// atomic and the hash key is claimed with a CAS.
__global__ void ensure_ghosts_kernel(MapView M, const int *items, const int *first, const int *found) {
    if (threadIdx.x) return;
    const int j = first[blockIdx.x];
    if (!found[j]) return;
    const int x = items[4 * j], y = items[4 * j + 1], z = items[4 * j + 2];
    if (hash_find(M, x, y, z) >= 0) return;
    const int top = atomicSub(M.free_top, 1) - 1;
    if (top < 0) {
        atomicAdd(M.free_top, 1);
        raise_error(M.error_flag, 1);
        return;
    }
    const int slot = M.free_list[top];
    const uint64_t key = pack_id(x, y, z);
    const uint64_t h = chunk_hash(x, y, z) & M.hash_mask;
    for (uint64_t i = 0; i <= M.hash_mask; i++) {
        const uint64_t idx = (h + i) & M.hash_mask;
        const uint64_t cur = M.hash_keys[idx];
        if ((cur == KEY_EMPTY || cur == KEY_TOMB) &&
            atomicCAS((unsigned long long *)&M.hash_keys[idx], (unsigned long long)cur, (unsigned long long)key) == cur) {
            M.hash_vals[idx] = slot;
            M.slot_key[slot] = key;
            bbox_include(M.mesh_ctl, x, y, z);
            return;
        }
    }
    raise_error(M.error_flag, 2);
}
// phase 2 (the next launch): one workgroup per item whose chunk exists now writes its box
__global__ __launch_bounds__(256) void import_shells_kernel(MapView M, const int *items, const long long *offs, const int *found, int N, const float *sdf,
                                                             const float *wgt, const uchar4 *rgbw) {
    __shared__ int s_slot;
    const int j = blockIdx.x;
    if (!found[j]) return;
    if (threadIdx.x == 0) s_slot = hash_find(M, items[4 * j], items[4 * j + 1], items[4 * j + 2]);
    __syncthreads();
    const int slot = s_slot, box = items[4 * j + 3];
    if (slot < 0) return;
    const int cx = box & 3, cy = (box >> 2) & 3, cz = (box >> 4) & 3;
    const int lx = shell_len(cx, N), ly = shell_len(cy, N), lz = shell_len(cz, N);
    const long long base = offs[j];
    const size_t dst = (size_t)slot * N * N * N;
    if (threadIdx.x == 0) slot_summary(M)[slot] = SUM_ANY;  // (voxels from outside: anything)
    for (int v = threadIdx.x; v < lx * ly * lz; v += 256) {
        const int x = shell_coord(cx, v % lx, N), y = shell_coord(cy, (v / lx) % ly, N), z = shell_coord(cz, v / (lx * ly), N);
        const size_t i = dst + (size_t)(z * N + y) * N + x;
        M.sdf[i] = sdf[base + v];
        M.wgt[i] = wgt[base + v];
        if (M.rgbw) M.rgbw[i] = rgbw ? rgbw[base + v] : make_uchar4(0, 0, 0, 0);
    }
}
// ---- the plan of a sharded recompute ON THE DEVICE (round 5) -----------------------------------------------------------------------
// Rounds 3-4 planned on the host (chisel_hip_mesh_shell_plan[_all]: 0.4-1.1 ms per recompute and rank, against ~0.1 ms of GPU work per
// step).  Here every rank derives, from the all-gathered list of updated chunks alone and without leaving the device:
//   * its own jobs (the 27-neighbourhoods of the updated chunks, de-duplicated in a hash set; those it owns);
//   * the shells it must SEND: for every job J of another rank r and every neighbour G = J + d that this rank owns, the box of G that J
//     reads (box code of d: shell_len above) -- one item per (J, d), less those whose box is part of another item's of the same (rank,
//     neighbour) (shell_item_covered below: a face box holds the edge and corner boxes the jobs beside it ask for; nobody has to agree on
//     a merge order, both sides apply the same test);
//   * how much it will RECEIVE from every owner (the same enumeration, counted from the other side).
// One small device-to-host copy -- per peer (items, voxels) to send and to receive, the job count -- is the only host wait of a sharded
// recompute.  What travels is a byte SEGMENT per (sender, receiver):
//     int32 head[4] = {items, voxels, 0, 0};  int32 item[items][8] = {x, y, z, box, found, first voxel, 0, 0};
//     float sdf[voxels];  float weight[voxels];  [uint32 rgbw[voxels]]
// The items sit in the order the sender's atomics produced: each one says where its voxels are, so the receiver needs no plan of its own.
struct ShellPlan {
    unsigned long long *jobset;    // [2 * jobset_capacity]: a hash set of the packed ids of all jobs (KEY_EMPTY = free), then the same ids as a list (ctl[4] of them)
    int jobset_capacity;           // power of two
    int *my_jobs;                  // [max_jobs][3] the jobs this rank owns
    int max_jobs;
    int *ctl;                      // [0] jobs of this rank, [1] overflow (job set / job list / item list), [2] largest per-rank entry count of the gathered list, [3] send items, [4] jobs of all ranks
    unsigned long long *send_cur;  // [n_shards] items | voxels << 32 this rank sends to each peer
    unsigned long long *recv_cnt;  // [n_shards] ... and receives from each
    int *send_items;               // [send_capacity][8]: x, y, z, box, destination, index within the destination's segment, first voxel, 0
    int send_capacity;
};
constexpr int SHELL_MAX_SHARDS = 64;
__host__ __device__ inline long long shell_segment_bytes(long long items, long long voxels, bool color) { return 16 + 32 * items + (color ? 12 : 8) * voxels; }
// box code of direction d = G - J (what job J reads of its neighbour G): per axis d > 0 -> {0, 1} (1), d < 0 -> {N - 1} (2), 0 -> all (0)
__host__ __device__ inline int shell_box_of(int dx, int dy, int dz) {
    return (dx > 0 ? 1 : (dx < 0 ? 2 : 0)) | ((dy > 0 ? 1 : (dy < 0 ? 2 : 0)) << 2) | ((dz > 0 ? 1 : (dz < 0 ? 2 : 0)) << 4);
}
// step 1: the job set.  gathered: per rank a block of 1 + 4 * cap ints -- count, then (x, y, z, flag) entries; flag 0 = an updated chunk
// (its 27-neighbourhood is meshed, Chisel.h:175-189), flag 1 = an id meshed as it is.  One thread per (entry, offset).
__global__ void shell_jobs_kernel(const int *__restrict__ gathered, int world, int cap, ShellPlan S, int n_shards, int shard_rank, int shard_block) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {
        int mx = 0;
        for (int r = 0; r < world; r++) mx = max(mx, gathered[(size_t)r * (1 + 4 * (size_t)cap)]);
        S.ctl[2] = mx;
    }
    const long long per_rank = 27ll * cap;
    if (t >= per_rank * world) return;
    const int r = (int)(t / per_rank), e = (int)((t % per_rank) / 27), o = (int)(t % 27);
    const int *blk = gathered + (size_t)r * (1 + 4 * (size_t)cap);
    if (e >= min(blk[0], cap)) return;
    const int *en = blk + 1 + 4 * e;
    if (en[3] != 0 && o != 13) return;
    const int x = en[0] + (en[3] ? 0 : o % 3 - 1), y = en[1] + (en[3] ? 0 : (o / 3) % 3 - 1), z = en[2] + (en[3] ? 0 : o / 9 - 1);
    const unsigned long long key = pack_id(x, y, z);
    const unsigned mask = (unsigned)S.jobset_capacity - 1u;
    unsigned h = (unsigned)(chunk_hash(x, y, z) * 0x9E3779B97F4A7C15ull >> 40) & mask;
    for (int probe = 0; probe < S.jobset_capacity; probe++, h = (h + 1u) & mask) {
        const unsigned long long cur = S.jobset[h];
        if (cur == key) return;
        if (cur == KEY_EMPTY) {
            const unsigned long long old = atomicCAS(&S.jobset[h], KEY_EMPTY, key);
            if (old == key) return;
            if (old != KEY_EMPTY) continue;  // (somebody else's id: next bucket)
            S.jobset[(size_t)S.jobset_capacity + atomicAdd(&S.ctl[4], 1)] = key;  // the list beside the set (as many entries as the set holds, at most)
            if (chunk_owner(x, y, z, n_shards, shard_block) == shard_rank) {
                const int pos = atomicAdd(&S.ctl[0], 1);
                if (pos < S.max_jobs) {
                    S.my_jobs[3 * pos] = x; S.my_jobs[3 * pos + 1] = y; S.my_jobs[3 * pos + 2] = z;
                } else {
                    S.ctl[1] = 1;
                }
            }
            return;
        }
    }
    S.ctl[1] = 1;  // the set is full
}
// (the job set is complete: shell_jobs_kernel is over)
__device__ inline bool shell_jobset_has(const ShellPlan &S, int x, int y, int z) {
    const unsigned long long key = pack_id(x, y, z);
    const unsigned mask = (unsigned)S.jobset_capacity - 1u;
    unsigned h = (unsigned)(chunk_hash(x, y, z) * 0x9E3779B97F4A7C15ull >> 40) & mask;
    for (int probe = 0; probe < S.jobset_capacity; probe++, h = (h + 1u) & mask) {
        const unsigned long long cur = S.jobset[h];
        if (cur == key) return true;
        if (cur == KEY_EMPTY) return false;
    }
    return false;
}
// Job J of rank r reads box(d) of its neighbour G = J + d.  Another job of r next to G may read a box of G that HOLDS this one -- the job
// at G - d', d' = d with some (not all) of its non-zero components zeroed: its box is "all" along those axes and the same along the others
// (the face box beside an edge box, the edge box beside a corner box).  Such an item is left out (round 6; both sides of a pair apply the
// same test, so what an owner sends is what the meshing rank counts on): 11 % of the voxels that travelled.
__device__ inline bool shell_item_covered(const ShellPlan &S, int gx, int gy, int gz, int dx, int dy, int dz, int r, int n_shards, int shard_block) {
    const int nz = (dx != 0) + (dy != 0) + (dz != 0);
    if (nz < 2) return false;
    for (int m = 1; m < 7; m++) {  // bit a of m: component a is zeroed
        const bool zx = m & 1, zy = (m >> 1) & 1, zz = (m >> 2) & 1;
        if ((zx && !dx) || (zy && !dy) || (zz && !dz)) continue;  // (only non-zero components can be zeroed)
        const int ex = zx ? 0 : dx, ey = zy ? 0 : dy, ez = zz ? 0 : dz;
        if (!(ex || ey || ez)) continue;                           // (... and not all of them)
        const int jx = gx - ex, jy = gy - ey, jz = gz - ez;
        if (chunk_owner(jx, jy, jz, n_shards, shard_block) == r && shell_jobset_has(S, jx, jy, jz)) return true;
    }
    return false;
}
// step 2: the items.  32 threads per job (26 directions), eight jobs per workgroup and round; the jobs come from the LIST shell_jobs_kernel
// keeps beside the set (a scan of the set's buckets -- 32 768 of them for a few thousand jobs -- was 17 us of a 25 us plan).  Counts and
// positions are reserved per workgroup in LDS, then once per (workgroup, peer) in memory.
__global__ __launch_bounds__(256) void shell_items_kernel(ShellPlan S, int N, int n_shards, int shard_rank, int shard_block) {
    __shared__ unsigned long long s_send[SHELL_MAX_SHARDS], s_recv[SHELL_MAX_SHARDS], s_base[SHELL_MAX_SHARDS];
    __shared__ int s_n, s_pos;
    const int n_all = min(S.ctl[4], S.jobset_capacity);
    const unsigned long long *joblist = S.jobset + S.jobset_capacity;
    for (int first = blockIdx.x * 8; first < n_all; first += gridDim.x * 8) {
        __syncthreads();  // (the round before has read its bases)
        if (threadIdx.x < SHELL_MAX_SHARDS) s_send[threadIdx.x] = s_recv[threadIdx.x] = 0ull;
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        const int job = first + (threadIdx.x >> 5), d = threadIdx.x & 31;
        bool sends = false;
        int gx = 0, gy = 0, gz = 0, box = 0, dest = 0, local = 0;
        unsigned long long mine = 0ull;
        if (job < n_all && d < 26) {
            int jx, jy, jz;
            unpack_id(joblist[job], jx, jy, jz);
            const int dd = d < 13 ? d : d + 1;  // (skip the job itself)
            const int dx = dd % 3 - 1, dy = (dd / 3) % 3 - 1, dz = dd / 9 - 1;
            gx = jx + dx; gy = jy + dy; gz = jz + dz;
            const int r = chunk_owner(jx, jy, jz, n_shards, shard_block), o = chunk_owner(gx, gy, gz, n_shards, shard_block);
            if (r != o && (o == shard_rank || r == shard_rank) && !shell_item_covered(S, gx, gy, gz, dx, dy, dz, r, n_shards, shard_block)) {
                box = shell_box_of(dx, dy, dz);
                const unsigned long long inc = 1ull | ((unsigned long long)shell_volume(box, N) << 32);
                if (o == shard_rank) {
                    sends = true;
                    dest = r;
                    mine = atomicAdd(&s_send[r], inc);  // this item's place among the workgroup's items for that peer
                    local = atomicAdd(&s_n, 1);
                } else if (r == shard_rank) {
                    atomicAdd(&s_recv[o], inc);
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < n_shards) {
            if (s_send[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&S.send_cur[threadIdx.x], s_send[threadIdx.x]);
            if (s_recv[threadIdx.x]) atomicAdd(&S.recv_cnt[threadIdx.x], s_recv[threadIdx.x]);
        }
        if (threadIdx.x == 0 && s_n) s_pos = atomicAdd(&S.ctl[3], s_n);
        __syncthreads();
        if (sends) {
            const unsigned long long at = s_base[dest] + mine;
            const int pos = s_pos + local;
            if (pos < S.send_capacity) {
                int *it = S.send_items + 8 * (size_t)pos;
                it[0] = gx; it[1] = gy; it[2] = gz; it[3] = box; it[4] = dest; it[5] = (int)(at & 0xffffffffull); it[6] = (int)(at >> 32); it[7] = 0;
            } else {
                S.ctl[1] = 1;
            }
        }
    }
}
// where segment `peer` begins in a buffer of consecutive segments, and its counts (cur: items | voxels << 32 per peer)
// (stride > 0: the wait-free form -- every segment has `stride` bytes to itself, whatever the others hold)
__device__ inline long long shell_segment_at(const unsigned long long *cur, int peer, bool color, int &items, long long &voxels, long long stride = 0) {
    long long off = stride * peer;
    for (int p = 0; p < peer && stride == 0; p++) off += shell_segment_bytes((long long)(cur[p] & 0xffffffffull), (long long)(cur[p] >> 32), color);
    items = (int)(cur[peer] & 0xffffffffull);
    voxels = (long long)(cur[peer] >> 32);
    return off;
}
// ---- the wait-free form of a sharded recompute (round 6) ------------------------------------------------------------------------------
// The host reads nothing of the plan: the segments have a FIXED size agreed on beforehand (from what the previous recompute needed), the
// exchange is an all-to-all of equal splits, and everything the steps behind the plan need to know they read on the device -- the export
// kernel the plan's cursors, the import / drop kernels the heads of the received segments.  What can go wrong is written into a STATUS
// vector by every rank right after its plan, all-reduced (MAX) in front of the exchange, and looked at
//   * by the kernels behind the exchange (`abort` = word 0 of the reduced vector): if any rank's list, plan or segment did not fit, NO rank
//     creates a ghost, meshes a job or clears a dirty flag -- the recompute has not happened and is made again the old way (with the host
//     reading the plan) from an untouched map;
//   * by the host when it next enters the map (ShardedChisel.Settle), which also takes the next recompute's sizes from it.
// status (written by the export kernel's first workgroup): [0] abort bits (1: a rank's dirty list was cut off, 2: job set / job list / item list overflow, 4: a segment exceeds the stride),
// [1] largest dirty count, [2] bytes of this rank's largest segment, [3] its jobs, [4] items it receives, [5] items it sends, [6] voxels it receives, [7] ghost chunks its earlier recomputes created (a running total, for the record)
constexpr int SHELL_STATUS_INTS = 8;
// (one thread, behind the plan's kernels: the first workgroup of the export kernel)
__device__ inline void shell_write_status(const ShellPlan &S, int cap, long long stride, bool color, int n_shards, int *status) {
    long long largest = 0, recv_items = 0, recv_voxels = 0;
    for (int p = 0; p < n_shards; p++) {
        largest = max(largest, shell_segment_bytes((long long)(S.send_cur[p] & 0xffffffffull), (long long)(S.send_cur[p] >> 32), color));
        recv_items += (long long)(S.recv_cnt[p] & 0xffffffffull);
        recv_voxels += (long long)(S.recv_cnt[p] >> 32);
    }
    const bool plan_over = S.ctl[1] != 0 || S.ctl[3] > S.send_capacity;
    status[0] = (S.ctl[2] > cap ? 1 : 0) | (plan_over ? 2 : 0) | ((stride > 0 && largest > stride) ? 4 : 0);
    status[1] = S.ctl[2];
    status[2] = (int)min(largest, 0x7fffffffll);
    status[3] = S.ctl[0];
    status[4] = (int)min(recv_items, 0x7fffffffll);
    status[5] = S.ctl[3];
    status[6] = (int)min(recv_voxels, 0x7fffffffll);
    status[7] = (int)min((long long)*reinterpret_cast<const unsigned long long *>(S.ctl + 16 + 4 * SHELL_MAX_SHARDS), 0x7fffffffll);  // ghost chunks this rank's earlier recomputes created (for the record)
}
// the all-reduce of the in-library group (host_group.h): every shard has the other shards' status vectors copied beside its own
__global__ void shell_status_max_kernel(const int *all, int world, int *reduced) {
    if (blockIdx.x != 0 || threadIdx.x >= SHELL_STATUS_INTS) return;
    int v = all[threadIdx.x];
    for (int r = 1; r < world; r++) v = max(v, all[r * SHELL_STATUS_INTS + threadIdx.x]);
    reduced[threadIdx.x] = v;
}
// behind the mesh step of a recompute that was called off: the count kernel has emptied the LIST of dirty slots (it does so whatever it meshed), their flags are all still up
// (clear_dirty_kernel left them) -- a list that says "overflowed" makes the next listing go by the flags (list_dirty_ids_kernel, mesh_mark_kernel)
__global__ void shell_abort_relist_kernel(MapView M, const int *abort) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && *abort) M.slot_dirty[2 * (size_t)M.max_chunks] = (unsigned)M.max_chunks + 1u;
}
// step 3 (owner): one workgroup per send item (or several items per workgroup: the wait-free form launches a fixed grid) packs its box
// into the destination's segment; workgroup 0 also writes the segment heads.  stride > 0: a segment that exceeds it carries a head that
// says so ({0, 0, 1, 0}) and nothing else.
__global__ __launch_bounds__(256) void shell_export_kernel(MapView M, ShellPlan S, int N, int n_shards, unsigned char *out, long long stride, int cap, int *status) {
    __shared__ int s_slot;
    const bool color = M.rgbw != nullptr;
    if (status && blockIdx.x == 0 && threadIdx.x == 64) shell_write_status(S, cap, stride, color, n_shards, status);  // (the wait-free form)
    if (blockIdx.x == 0 && threadIdx.x < n_shards) {
        int items;
        long long voxels;
        const long long off = shell_segment_at(S.send_cur, threadIdx.x, color, items, voxels, stride);
        const bool fits = stride == 0 || shell_segment_bytes(items, voxels, color) <= stride;
        int *head = reinterpret_cast<int *>(out + off);
        head[0] = fits ? items : 0; head[1] = fits ? (int)voxels : 0; head[2] = fits ? 0 : 1; head[3] = 0;
    }
    const int n = min(S.ctl[3], S.send_capacity);
    for (int j = blockIdx.x; j < n; j += gridDim.x) {
        const int *it = S.send_items + 8 * (size_t)j;
        int items;
        long long voxels;
        const long long off = shell_segment_at(S.send_cur, it[4], color, items, voxels, stride);
        if (stride > 0 && shell_segment_bytes(items, voxels, color) > stride) continue;  // (wave-uniform, workgroup-uniform)
        __syncthreads();  // (s_slot of the item before)
        if (threadIdx.x == 0) {
            s_slot = hash_find(M, it[0], it[1], it[2]);
            int *rec = reinterpret_cast<int *>(out + off + 16 + 32 * (long long)it[5]);
            rec[0] = it[0]; rec[1] = it[1]; rec[2] = it[2]; rec[3] = it[3]; rec[4] = s_slot >= 0 ? 1 : 0; rec[5] = it[6]; rec[6] = 0; rec[7] = 0;
        }
        __syncthreads();
        const int slot = s_slot, box = it[3];
        if (slot < 0) continue;  // (an absent chunk: the receiver skips the item, its voxels stay unwritten)
        const int cx = box & 3, cy = (box >> 2) & 3, cz = (box >> 4) & 3;
        const int lx = shell_len(cx, N), ly = shell_len(cy, N), lz = shell_len(cz, N);
        float *sdf = reinterpret_cast<float *>(out + off + 16 + 32 * (long long)items) + it[6];
        float *wgt = sdf + voxels;
        unsigned *col = reinterpret_cast<unsigned *>(wgt + voxels);
        const size_t src = (size_t)slot * N * N * N;
        for (int v = threadIdx.x; v < lx * ly * lz; v += 256) {
            const int x = shell_coord(cx, v % lx, N), y = shell_coord(cy, (v / lx) % ly, N), z = shell_coord(cz, v / (lx * ly), N);
            const size_t i = src + (size_t)(z * N + y) * N + x;
            sdf[v] = M.sdf[i];
            wgt[v] = M.wgt[i];
            if (color) col[v] = reinterpret_cast<const unsigned *>(M.rgbw)[i];
        }
    }
}
// The received buffer: consecutive segments, one per owner (empty for this rank itself), at byte offsets seg[peer].
struct ShellSegments {
    long long off[SHELL_MAX_SHARDS + 1];
    int first_item[SHELL_MAX_SHARDS + 1];  // items of the segments in front of each one (from the plan's receive counts)
};
__device__ inline const int *shell_received_item(const unsigned char *in, const ShellSegments &G, int n_shards, int j, int &peer) {
    peer = 0;
    while (peer + 1 < n_shards && G.first_item[peer + 1] <= j) peer++;
    return reinterpret_cast<const int *>(in + G.off[peer] + 16 + 32 * (long long)(j - G.first_item[peer]));
}
// step 4a (requester): a ghost chunk for every received item whose chunk was found at its owner -- several items may name one ghost, the
// thread whose compare-and-swap enters the key creates it.  (The ids are absent before: ghosts are dropped after every recompute.)
__device__ inline void shell_ensure_ghost(const MapView &M, const int *it, unsigned long long *ghosts_created) {
    if (!it[4]) return;
    const int x = it[0], y = it[1], z = it[2];
    const uint64_t key = pack_id(x, y, z);
    const uint64_t h = chunk_hash(x, y, z) & M.hash_mask;
    for (uint64_t i = 0; i <= M.hash_mask; i++) {
        const uint64_t idx = (h + i) & M.hash_mask;
        const uint64_t cur = M.hash_keys[idx];
        if (cur == key) return;
        if (cur != KEY_EMPTY && cur != KEY_TOMB) continue;
        const uint64_t old = atomicCAS((unsigned long long *)&M.hash_keys[idx], (unsigned long long)cur, (unsigned long long)key);
        if (old == key) return;
        if (old != cur) {  // the bucket went to another id meanwhile: look at it again
            i--;
            continue;
        }
        const int top = atomicSub(M.free_top, 1) - 1;
        if (top < 0) {
            atomicAdd(M.free_top, 1);
            M.hash_keys[idx] = KEY_TOMB;
            raise_error(M.error_flag, 1);
            return;
        }
        const int slot = M.free_list[top];
        M.hash_vals[idx] = slot;
        M.slot_key[slot] = key;
        bbox_include(M.mesh_ctl, x, y, z);
        if (ghosts_created) atomicAdd(ghosts_created, 1ull);
        return;
    }
    raise_error(M.error_flag, 2);
}
__global__ void shell_ensure_ghosts_kernel(MapView M, const unsigned char *__restrict__ in, ShellSegments G, int n_shards, int n_items, unsigned long long *ghosts_created) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_items) return;
    int peer;
    shell_ensure_ghost(M, shell_received_item(in, G, n_shards, j, peer), ghosts_created);
}
// step 4b (the next launch): one workgroup per received item writes its box into the ghost (s_slot: a word of LDS; the workgroup's threads
// arrive together and leave together)
__device__ inline void shell_import_item(const MapView &M, const unsigned char *__restrict__ in, long long seg_off, const int *it, int N, int *s_slot) {
    __syncthreads();  // (a workgroup that takes several items: s_slot of the one before)
    if (threadIdx.x == 0) *s_slot = it[4] ? hash_find(M, it[0], it[1], it[2]) : -1;
    __syncthreads();
    const int slot = *s_slot, box = it[3];
    if (slot < 0) return;
    const int *head = reinterpret_cast<const int *>(in + seg_off);
    const long long items = head[0], voxels = head[1];
    const float *sdf = reinterpret_cast<const float *>(in + seg_off + 16 + 32 * items) + it[5];
    const float *wgt = sdf + voxels;
    const unsigned *col = reinterpret_cast<const unsigned *>(wgt + voxels);
    const int cx = box & 3, cy = (box >> 2) & 3, cz = (box >> 4) & 3;
    const int lx = shell_len(cx, N), ly = shell_len(cy, N), lz = shell_len(cz, N);
    const size_t dst = (size_t)slot * N * N * N;
    if (threadIdx.x == 0) slot_summary(M)[slot] = SUM_ANY;  // (voxels from outside: anything)
    for (int v = threadIdx.x; v < lx * ly * lz; v += 256) {
        const int x = shell_coord(cx, v % lx, N), y = shell_coord(cy, (v / lx) % ly, N), z = shell_coord(cz, v / (lx * ly), N);
        const size_t i = dst + (size_t)(z * N + y) * N + x;
        M.sdf[i] = sdf[v];
        M.wgt[i] = wgt[v];
        if (M.rgbw) reinterpret_cast<unsigned *>(M.rgbw)[i] = col[v];
    }
}
__global__ __launch_bounds__(256) void shell_import_kernel(MapView M, const unsigned char *__restrict__ in, ShellSegments G, int n_shards, int N) {
    __shared__ int s_slot;
    int peer;
    const int *it = shell_received_item(in, G, n_shards, blockIdx.x, peer);
    shell_import_item(M, in, G.off[peer], it, N, &s_slot);
}
// step 6: the ghosts go again: shell_reset_boxes_kernel + shell_remove_ghosts_kernel below (both forms of the recompute)
// ---- the same three steps in the wait-free form: segments `stride` bytes apart, their item counts in their heads (a head that says
// "did not fit" counts as empty), a fixed grid whose workgroups take the items in turn, and nothing at all when the recompute was called off
// (`abort`: word 0 of the all-reduced status) -- or, for the drop (two kernels, below), while the mesh step in front of it is still to be emitted again
// (`latch`: mesh_ctl[MC_LATCH]: the second emission reads the ghosts; check_mesh_totals launches the drop again behind it) ---------------
__device__ inline void shell_segments_fixed(const unsigned char *__restrict__ in, long long stride, int n_shards, ShellSegments *G) {
    if (threadIdx.x == 0) {
        int items = 0;
        for (int p = 0; p < n_shards; p++) {
            const int *head = reinterpret_cast<const int *>(in + stride * p);
            G->off[p] = stride * p;
            G->first_item[p] = items;
            items += head[2] ? 0 : head[0];
        }
        G->off[n_shards] = stride * n_shards;
        G->first_item[n_shards] = items;
    }
    __syncthreads();
}
__global__ __launch_bounds__(256) void shell_ensure_ghosts_fixed_kernel(MapView M, const unsigned char *__restrict__ in, long long stride, int n_shards, const int *abort,
                                                                        unsigned long long *ghosts_created, int *plan_ctl) {
    __shared__ ShellSegments G;
    if (abort && *abort) {
        if (blockIdx.x == 0 && threadIdx.x == 0) plan_ctl[0] = 0;  // a recompute that was called off has no jobs (the mesh step behind this kernel reads the count)
        return;
    }
    shell_segments_fixed(in, stride, n_shards, &G);
    const int total = G.first_item[n_shards];
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        int peer;
        shell_ensure_ghost(M, shell_received_item(in, G, n_shards, j, peer), ghosts_created);
    }
}
__global__ __launch_bounds__(256) void shell_import_fixed_kernel(MapView M, const unsigned char *__restrict__ in, long long stride, int n_shards, int N, const int *abort) {
    __shared__ ShellSegments G;
    __shared__ int s_slot;
    if (abort && *abort) return;
    shell_segments_fixed(in, stride, n_shards, &G);
    const int total = G.first_item[n_shards];
    for (int j = blockIdx.x; j < total; j += gridDim.x) {
        int peer;
        const int *it = shell_received_item(in, G, n_shards, j, peer);
        shell_import_item(M, in, G.off[peer], it, N, &s_slot);
    }
}
// The drop in two passes.  A ghost holds default voxels everywhere but in the boxes that were written into it, so only those are restored
// (a sixth of the chunk on average; the one-pass kernel of round 5 restored all of it, from the one workgroup that won the ghost's key, while the
// workgroups of its other items waited for nothing: 40 us per recompute against 12 + 5): pass 1, one workgroup per item, puts the item's box
// back to default voxels; pass 2, one thread per item, takes the keys out and frees the slots.
// (stride == 0: the blocking form -- the segments lie back to back where `packed` says, as the host laid them out from the plan's counts)
__global__ __launch_bounds__(256) void shell_reset_boxes_kernel(MapView M, const unsigned char *__restrict__ in, long long stride, ShellSegments packed, int n_shards, int N,
                                                                const int *abort, const int *latch) {
    __shared__ ShellSegments G;
    __shared__ int s_slot;
    if ((abort && *abort) || (latch && *latch)) return;
    if (stride > 0) {
        shell_segments_fixed(in, stride, n_shards, &G);
    } else {
        if (threadIdx.x == 0) G = packed;
        __syncthreads();
    }
    const int total = G.first_item[n_shards];
    for (int j = blockIdx.x; j < total; j += gridDim.x) {
        int peer;
        const int *it = shell_received_item(in, G, n_shards, j, peer);
        __syncthreads();
        if (threadIdx.x == 0) s_slot = it[4] ? hash_find(M, it[0], it[1], it[2]) : -1;
        __syncthreads();
        const int slot = s_slot, box = it[3];
        if (slot < 0) continue;
        const int cx = box & 3, cy = (box >> 2) & 3, cz = (box >> 4) & 3;
        const int lx = shell_len(cx, N), ly = shell_len(cy, N), lz = shell_len(cz, N);
        const size_t dst = (size_t)slot * N * N * N;
        for (int v = threadIdx.x; v < lx * ly * lz; v += 256) {
            const int x = shell_coord(cx, v % lx, N), y = shell_coord(cy, (v / lx) % ly, N), z = shell_coord(cz, v / (lx * ly), N);
            const size_t i = dst + (size_t)(z * N + y) * N + x;
            M.sdf[i] = 99999.0f;  // (fill_default_chunk's values)
            M.wgt[i] = 0.0f;
            if (M.rgbw) reinterpret_cast<unsigned *>(M.rgbw)[i] = 0u;
        }
    }
}
__global__ __launch_bounds__(256) void shell_remove_ghosts_kernel(MapView M, const unsigned char *__restrict__ in, long long stride, ShellSegments packed, int n_shards,
                                                                  const int *abort, const int *latch) {
    __shared__ ShellSegments G;
    if ((abort && *abort) || (latch && *latch)) return;
    if (stride > 0) {
        shell_segments_fixed(in, stride, n_shards, &G);
    } else {
        if (threadIdx.x == 0) G = packed;
        __syncthreads();
    }
    const int total = G.first_item[n_shards];
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        int peer;
        const int *it = shell_received_item(in, G, n_shards, j, peer);
        if (!it[4]) continue;
        uint64_t where = 0;
        const int slot = hash_find(M, it[0], it[1], it[2], &where);
        if (slot < 0) continue;
        const uint64_t key = pack_id(it[0], it[1], it[2]);
        if (atomicCAS((unsigned long long *)&M.hash_keys[where], (unsigned long long)key, (unsigned long long)KEY_TOMB) != key) continue;  // (another item of the same ghost)
        M.slot_key[slot] = KEY_EMPTY;
        M.slot_dirty[slot] = 0;
        slot_summary(M)[slot] = 0;
        if (M.mesh_flag) M.mesh_flag[slot] = 0;
        __threadfence();
        const int pos = atomicAdd(M.free_top, 1);
        M.free_list[pos] = slot;
    }
}

// the chunks updated since the last recompute (Chisel.h:175-189 marks their 27-neighbourhoods; the expansion is the planner's), as
// (x, y, z, 0) entries behind a count, for the all-gather of a sharded recompute: out[0] = n, out[1 + 4 i ...] = entry i
__global__ void list_dirty_ids_kernel(MapView M, int *out, int capacity) {
    const unsigned listed = M.slot_dirty[2 * (size_t)M.max_chunks];
    const bool scan = listed > (unsigned)M.max_chunks;
    const long long n = scan ? (long long)M.max_chunks : (long long)listed;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const int slot = scan ? (int)t : (int)M.slot_dirty[(size_t)M.max_chunks + t];
        if (!M.slot_dirty[slot]) continue;
        const uint64_t key = M.slot_key[slot];
        if (key == KEY_EMPTY) continue;
        const int pos = atomicAdd(&out[0], 1);
        if (pos < capacity) {
            int x, y, z;
            unpack_id(key, x, y, z);
            out[1 + 4 * pos] = x; out[2 + 4 * pos] = y; out[3 + 4 * pos] = z; out[4 + 4 * pos] = 0;
        }
    }
}
// The tail of the dirty list: the chunks dirtied since entry `from` (chisel_hip_meshes_to_update_since), into page-locked host memory:
// out[0] = entries of the list now, out[1] = ids written, then (x, y, z) each.  One workgroup.
__global__ __launch_bounds__(256) void list_dirty_tail_kernel(MapView M, unsigned from, int *out, int capacity) {
    __shared__ int s_n;
    const unsigned listed = M.slot_dirty[2 * (size_t)M.max_chunks];
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    if (listed <= (unsigned)M.max_chunks) {
        for (unsigned t = from + threadIdx.x; t < listed; t += blockDim.x) {
            const int slot = (int)M.slot_dirty[(size_t)M.max_chunks + t];
            if (!M.slot_dirty[slot]) continue;  // (removed since it was listed: the host keeps those ids itself)
            const uint64_t key = M.slot_key[slot];
            if (key == KEY_EMPTY) continue;
            const int pos = atomicAdd(&s_n, 1);
            if (pos < capacity) {
                int x, y, z;
                unpack_id(key, x, y, z);
                out[2 + 3 * pos] = x; out[3 + 3 * pos] = y; out[4 + 3 * pos] = z;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[1] = s_n;
        __threadfence_system();
        out[0] = (int)listed;
    }
}
__global__ void clear_dirty_kernel(MapView M, const int *abort) {  // (abort: a sharded recompute that was called off keeps its dirty flags)
    if (abort && *abort) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M.max_chunks) {
        M.slot_dirty[i] = 0;
        if (M.mesh_flag) M.mesh_flag[i] = 0;
    }
    if (i == 0) {
        M.slot_dirty[2 * (size_t)M.max_chunks] = 0;  // and their list
        if (M.mesh_ctl) M.mesh_ctl[4] = 0;           // and the job list kept while integrating
    }
}

// ---- PublishDenseInfo's image conditioning (collaborative_server_system.cpp:213-214, :255-269) -------------------------
// cv::resize(..., INTER_LINEAR) restated (OpenCV imgproc/resize.cpp, 3.x / 4.x without IPP; parity unpinned -- no OpenCV in this
// image).  scale = 1. / ((double)dst / src).  Along x: tap position (dx + 0.5) * scale - 0.5 narrowed to float, sx = floor,
// fx -= sx; sx < 0 -> (0, fx = 0); sx >= width - 1 -> (width - 1, fx = 0) and the horizontal pass there is S[sx] * ONE.  Along y
// the fraction is NOT reset: the two source rows are clipped to the image instead.  Exactly halving both axes makes
// cv::resize substitute INTER_AREA (mean of the 2 x 2 block); equal sizes are copied.
struct ResizeTap {
    int s;      // first source sample
    float f;    // weight of the second one
    bool edge;  // second sample beyond the row: the horizontal pass is S[s] * ONE
};
__device__ inline double resize_scale(int dst, int src) { return 1.0 / ((double)dst / (double)src); }
__device__ inline ResizeTap resize_tap_x(int d, double scale, int ssize) {
    ResizeTap t;
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int si = (int)floorf(f);
    f -= (float)si;
    if (si < 0) {
        f = 0.0f;
        si = 0;
    }
    t.edge = si >= ssize - 1;
    if (t.edge) {
        f = 0.0f;
        si = ssize - 1;
    }
    t.s = si;
    t.f = f;
    return t;
}
__device__ inline void resize_tap_y(int d, double scale, int ssize, int &r0, int &r1, float &f) {
    f = (float)(((double)d + 0.5) * scale - 0.5);
    const int si = (int)floorf(f);
    f -= (float)si;
    r0 = min(max(si, 0), ssize - 1);
    r1 = min(max(si + 1, 0), ssize - 1);
}
__global__ void condition_depth_kernel(const double *__restrict__ src, int w0, int h0, float *__restrict__ dst, int w, int h) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    double v;
    if (w == w0 && h == h0) {
        v = src[(size_t)y * w0 + x];  // same size: cv::resize copies
    } else if (w0 == 2 * w && h0 == 2 * h) {
        // INTER_AREA, integer scale: sum of the block in row-major order (double), times the float 1 / area
        const double *s = src + (size_t)(2 * y) * w0 + 2 * x;
        v = (((s[0] + s[1]) + s[w0]) + s[w0 + 1]) * (double)0.25f;
    } else {
        const ResizeTap tx = resize_tap_x(x, resize_scale(w, w0), w0);
        int y0, y1;
        float fy;
        resize_tap_y(y, resize_scale(h, h0), h0, y0, y1, fy);
        const float a0 = 1.0f - tx.f, a1 = tx.f, b0 = 1.0f - fy, b1 = fy;
        // horizontal pass of both rows (double work type, float weights), then the vertical one
        const double *s0 = src + (size_t)y0 * w0 + tx.s, *s1 = src + (size_t)y1 * w0 + tx.s;
        const double r0 = tx.edge ? s0[0] * 1.0 : s0[0] * (double)a0 + s0[1] * (double)a1;
        const double r1 = tx.edge ? s1[0] * 1.0 : s1[0] * (double)a0 + s1[1] * (double)a1;
        v = r0 * (double)b0 + r1 * (double)b1;
    }
    float f = (float)v;                                            // convertTo(CV_32FC1)
    if (f < 0.1f || f > 20.0f) f = __builtin_nanf("");             // :262-265
    dst[(size_t)y * w + x] = f;
}
// 8-bit images (MONO8 / BGR8 / BGRA8, cn interleaved channels): OpenCV's fixed-point path.  Weights are shorts,
// round-to-nearest-even of the float weight * 2048; horizontal pass in int (S[sx] * 2048 at the edge); vertical pass
// uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2).  Halving: (a + b + c + d + 2) >> 2.
__device__ inline int resize_coef(float c) { return (int)rintf(c * 2048.0f); }
__global__ void condition_color_kernel(const uint8_t *__restrict__ src, int w0, int h0, int cn, uint8_t *__restrict__ dst, int w, int h) {
    const int xc = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;  // xc: pixel * cn + channel
    if (xc >= w * cn || y >= h) return;
    const int x = xc / cn, c = xc - x * cn;
    uint8_t out;
    if (w == w0 && h == h0) {
        out = src[(size_t)y * w0 * cn + xc];
    } else if (w0 == 2 * w && h0 == 2 * h) {
        const uint8_t *s = src + ((size_t)(2 * y) * w0 + 2 * x) * cn + c;
        out = (uint8_t)((s[0] + s[cn] + s[(size_t)w0 * cn] + s[(size_t)w0 * cn + cn] + 2) >> 2);
    } else {
        const ResizeTap tx = resize_tap_x(x, resize_scale(w, w0), w0);
        int y0, y1;
        float fy;
        resize_tap_y(y, resize_scale(h, h0), h0, y0, y1, fy);
        const int a0 = resize_coef(1.0f - tx.f), a1 = resize_coef(tx.f), b0 = resize_coef(1.0f - fy), b1 = resize_coef(fy);
        const uint8_t *s0 = src + ((size_t)y0 * w0 + tx.s) * cn + c, *s1 = src + ((size_t)y1 * w0 + tx.s) * cn + c;
        const int r0 = tx.edge ? s0[0] * 2048 : s0[0] * a0 + s0[cn] * a1;
        const int r1 = tx.edge ? s1[0] * 2048 : s1[0] * a0 + s1[cn] * a1;
        out = (uint8_t)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
    }
    dst[(size_t)y * w * cn + xc] = out;
}

// CollaborativeServer::SendPointCloud (collaborative_server_system.cpp:318-381): the organised "point cloud" PublishDenseInfo sends
// beside the images -- one 16-byte point per pixel: x = column, y = row (pixel coordinates, as floats), z = depth narrowed to float,
// rgb = the grey byte at (row, column) of the colour image replicated into 0x00gggggg; all four words NaN unless 0.1 < depth < 10.
// The colour byte is mColorImage.at<uint8_t>(u, v): byte v of row u whatever the channel count (color_step = bytes per row).
__global__ void publish_cloud_kernel(const double *__restrict__ depth, const uint8_t *__restrict__ color, int w, int h, int color_step,
                                     uint4 *__restrict__ points) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x, u = blockIdx.y;
    if (v >= w || u >= h) return;
    const float dep = (float)depth[(size_t)u * w + v];
    uint4 p;
    if (dep < 10.0f && dep > 0.1f) {
        const unsigned g = color[(size_t)u * color_step + v];
        p.x = __float_as_uint((float)v);
        p.y = __float_as_uint((float)u);
        p.z = __float_as_uint(dep);
        p.w = (g << 16) | (g << 8) | g;
    } else {
        p.x = p.y = p.z = p.w = 0x7fc00000u;  // std::numeric_limits<float>::quiet_NaN()
    }
    points[(size_t)u * w + v] = p;
}

// ---- known-answer kernels: the device arithmetic against the reference-built golden vectors ----------
__global__ void kat_truncation_kernel(int kind, float param, const float *depths, int n, float *trunc, float *weight1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float t = truncation_distance(kind, param, depths[i]);
        trunc[i] = t;
        weight1[i] = constant_weight(1.0f, t);
    }
}
// ops: n x (op, d, wu): op 0 = carve, 1 = integrate; out: n x (sdf, w)
__global__ void kat_dist_kernel(const float *ops, int n, float *out) {
    if (blockIdx.x || threadIdx.x) return;
    float sdf = 99999.0f, w = 0.0f;
    for (int i = 0; i < n; i++) {
        if (ops[3 * i] == 0.0f) {
            sdf = 99999.0f;
            w = 0.0f;
        } else {
            dist_integrate(sdf, w, ops[3 * i + 1], ops[3 * i + 2]);
        }
        out[2 * i] = sdf;
        out[2 * i + 1] = w;
    }
}
// ops: n x (r, g, b, wu) bytes; out: n x (r, g, b, w)
__global__ void kat_color_kernel(const uint8_t *ops, int n, uint8_t *out) {
    if (blockIdx.x || threadIdx.x) return;
    uchar4 c = make_uchar4(0, 0, 0, 0);
    for (int i = 0; i < n; i++) {
        c = color_integrate(c, ops[4 * i], ops[4 * i + 1], ops[4 * i + 2], ops[4 * i + 3]);
        out[4 * i] = c.x;
        out[4 * i + 1] = c.y;
        out[4 * i + 2] = c.z;
        out[4 * i + 3] = c.w;
    }
}
// exhaustive: color_integrate_fresh (division-free, packed) against color_integrate for every weight < 8, old and new channel
// value (the three channels carry old, 255 - old and old ^ 0x5a; new likewise); counts the words that differ
__global__ void kat_color_fresh_kernel(unsigned *mismatches) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;  // 8 * 256 * 256 cases
    const unsigned w = i >> 16, o = (i >> 8) & 0xffu, n = i & 0xffu;
    if (w >= 8u) return;
    const uchar4 c = make_uchar4((uint8_t)o, (uint8_t)(255u - o), (uint8_t)(o ^ 0x5au), (uint8_t)w);
    const uint8_t r = (uint8_t)n, g = (uint8_t)(255u - n), b = (uint8_t)(n ^ 0x5au);
    const uchar4 want = color_integrate(c, r, g, b, 1);
    const unsigned got = color_integrate_fresh((unsigned)c.x | ((unsigned)c.y << 8) | ((unsigned)c.z << 16) | ((unsigned)c.w << 24),
                                               (unsigned)r | ((unsigned)g << 8) | ((unsigned)b << 16));
    const unsigned want_bits = (unsigned)want.x | ((unsigned)want.y << 8) | ((unsigned)want.z << 16) | ((unsigned)want.w << 24);
    if (got != want_bits) atomicAdd(mismatches, 1u);
    // and the form the integration kernel uses: the new colour as the pixel's bytes lie in memory (blue, green, red[, alpha])
    const unsigned got2 = color_integrate_fresh_bgr((unsigned)c.x | ((unsigned)c.y << 8) | ((unsigned)c.z << 16) | ((unsigned)c.w << 24),
                                                    (unsigned)b | ((unsigned)g << 8) | ((unsigned)r << 16) | ((i * 2654435761u) & 0xff000000u));
    if (got2 != want_bits) atomicAdd(mismatches, 1u);
}
// exhaustive: reciprocal_in_range(z) against the IEEE division 1.0f / z for every float in [FASTZ_MIN, FASTZ_MAX]
// (81 binades x 2^23 mantissas); counts the mismatches and keeps one offending input
__global__ void kat_reciprocal_kernel(unsigned first_bits, unsigned long long n, unsigned long long *mismatches, unsigned *example) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float z = __uint_as_float(first_bits + (unsigned)i);
        const float want = 1.0f / z;
        const float got = reciprocal_in_range(z);
        if (__float_as_uint(want) != __float_as_uint(got)) {
            bad++;
            *example = first_bits + (unsigned)i;
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

// exhaustive over the 2^32 float bit patterns: floor_to_int(x) (v_cvt_flr_i32_f32) against (int)floorf(x) as the compiler emits
// it (v_floor_f32 + v_cvt_i32_f32) for every x that is not a NaN; for a NaN (the pair gives 0) the result must not be a possible
// pixel coordinate: the integration kernel's image test is (unsigned)floor_to_int(u) < W and has no separate NaN test
__global__ void kat_floor_kernel(unsigned long long *mismatches, unsigned *example) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((unsigned)i);
        const int got = floor_to_int(x);
        const bool ok = (x == x) ? (got == (int)floorf(x)) : ((unsigned)got >= (1u << 27));
        if (!ok) {
            bad++;
            *example = (unsigned)i;
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

}  // namespace chisel_hip
