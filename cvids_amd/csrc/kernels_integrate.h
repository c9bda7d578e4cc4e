// kernels_integrate.h -- projective SDF / weight / colour integration, one WAVE per 64 quads of a work-list chunk.
//
// Replaces ProjectionIntegrator::Integrate<float> (ProjectionIntegrator.h:51-99) and ::IntegrateColor<float,uint8_t>
// (:101-183) plus the allocate-everything / erase-untouched protocol of Chisel::IntegrateDepthScan[Color]
// (Chisel.h:77-108, 133-143, 202-207).
//
// Why waves and not workgroups.  A launch applies up to KMAX frames and every voxel must see them in frame order
// (DistVoxel::Integrate is a running average, DistVoxel.h:52-60), so a unit of work is a serial chain of K frames.  With a
// workgroup per chunk (4096 voxels, 8 voxels per thread, 128 registers, a barrier and a staged pixel tile per frame) the
// chip held 512 chains of K x 3-5 us and a launch lasted ceil(items / 512) such chains: 0.19 of the HBM roofline with the
// vector units 13-23 % busy.  Here the unit is a wave that owns 64 "quads" -- a lane's 4 (or 2) x-consecutive voxels; 256
// voxels are one z-layer of a 16^3 chunk --, 77 (57) registers -> six to eight waves per SIMD, thousands of chains in flight,
// each K x (one quad's work).  Waves never meet: no barrier, no LDS; the pixel records come from L2 / the Infinity Cache (a
// batch's frames stay there), and a wave whose layer lies outside the band and the carve region of a frame skips that frame
// after 16 instructions.
//
//   - work unit `wid` = (work item, 64-quad group), dealt statically so that the groups of one chunk sit on one XCD and every
//     XCD gets the same mix of expensive and cheap chunks; the host sizes the grid to about one unit per wave from the item
//     count a recent launch reported (chisel_hip.hip: launch_group), whatever exceeds the grid is pulled from up to 128 queue
//     heads (one returning atomic per unit; a head serves workgroups of one XCD, one neighbour is tried when it runs dry);
//   - a lane owns one quad: every voxel-plane access of a wave is 1 KiB (512 B) contiguous; the state is read once (at the
//     first frame that can touch the quad), lives in registers for the batch and is written once, only where it changed;
//   - a lane's verdicts are bit masks over its voxels and its counters per-lane adds, reduced when the wave retires; branches
//     are wave-uniform (__any): the scalar unit is shared by the four SIMDs of a CU and lane-mask arithmetic saturates it;
//   - chunk-level facts the reference derives per frame ("did any voxel integrate", "did anything change") travel through a
//     per-item record in HBM touched by device-scope atomics only (ItemSync): the first wave that integrates a voxel of a
//     chunk without a slot allocates it (free-list pop + hash CAS) and publishes the slot, its siblings take it from there;
//     the `probe` counter of such a chunk (carve tests of frames after the one that created it) is settled by the wave that
//     arrives last.  Resident chunks need one atomic OR per wave that changed something.
//   - per-voxel arithmetic follows the reference operation by operation in fp32 (compiled with -ffp-contract=off, IEEE
//     divide), 3-term sums in Eigen's a0 + (a1 + a2) order.
// Build parameters for experiments: INTEGRATE_WAVES / INTEGRATE_BLOCKS_PER_CU (occupancy),
// CHISEL_PHASES (in-kernel timers and utilisation counters), CHISEL_ABLATE_GATHER.
#pragma once
#include <type_traits>

#include "chisel_device.h"
#include "kernels_cull.h"  // COUNT_* (the batch counters the work-list builders leave behind)
#include "kernels_map.h"   // mesh_expand_dirty

namespace chisel_hip {

#ifndef INTEGRATE_WAVES
#define INTEGRATE_WAVES 6      // waves per SIMD the register allocator must leave room for (<= 80 VGPRs)
#endif
#ifndef INTEGRATE_BLOCKS_PER_CU
#define INTEGRATE_BLOCKS_PER_CU 6
#endif
// Waves per workgroup.  The waves of a workgroup never meet (no barrier, no LDS), but the hardware takes a workgroup's place back
// only when its LAST wave has ended: with four waves per workgroup a place idles for the difference between the longest and the
// mean of four chains.
// Measured on the driver's window (640x480 / 1 cm, 10 frames per launch, 1 257 items; round 3): 4 voxels per lane 113 us with four
// waves per workgroup, 98-100 us with one; 2 voxels per lane 110 -> 100 us with eight instead of six waves per SIMD.
#ifndef INTEGRATE_WPB
#define INTEGRATE_WPB 1
#endif
// Voxels per lane (x-consecutive), a template parameter of the kernel: 4 for launches that fill the chip many times over (least
// arithmetic per voxel: the quad's y / z terms are shared), 2 for small launches (twice as many units of half the length: what
// such a launch takes is the length of its longest chains, not its arithmetic).  The host picks per launch: measured at 640x480 /
// 1 cm, 10 frames per launch, 2 voxels per lane win by 10 % at 540 work items (61 against 68 us) and by 3 % at 1 240, lose 17 % at
// 1 500+ items (318 against 272 us) and 20 % on one-frame launches (nothing to shorten: 21 against 17 us); and 9 % on a 16-frame
// launch of the 4-agent stream, where a chunk is seen by 4-5 of the frames: the choice also asks for >= 6 frames per item on average.
#ifndef INTEGRATE_FINE_BELOW
#define INTEGRATE_FINE_BELOW 600   // work items (16^3 chunks; scaled by voxels per chunk) below which a launch of >= 4 frames runs with 2 voxels per lane
                                   // (round 3, one wave per workgroup: at 1 257 items 4 voxels per lane take 98 us, 2 take 100-106; at 540 items 63.2 against 62.6;
                                   // round 4, bricks + cell masks: default window 48.0 us with 600, 50.3 with 900, 53.5 with 1 300; driver's window 73.7 / 73.6 / 86.6)
#endif

// Round-3 instruction-count work, each switchable for A/B builds (all on by default):
#ifndef INTEGRATE_PIPE
#define INTEGRATE_PIPE 0  // records of frame k + 1 requested before frame k is applied (see run_unit)
#endif
#ifndef OPT_INSIDE
#define OPT_INSIDE 1     // chunks whose voxels all project onto the image (WI_INSIDE): no per-voxel image tests, offset by one mad
#endif
#ifndef OPT_ENDZ
#define OPT_ENDZ 1       // camera-z extrema of a quad from its two end voxels (z is monotone along x)
#endif
#ifndef OPT_CARVESKIP
#define OPT_CARVESKIP 1  // no carve (band) verdicts in wave-frames where no lane's z interval reaches the carve (band) region
#endif
#ifndef OPT_COLOR2
#define OPT_COLOR2 1     // colour sample: bytes converted where they lie (no swizzle), one fma less per channel, clamp by min
#endif

constexpr int QUEUE_STRIDE = 32;  // ints between two queue heads (one 128-byte line each)
constexpr int QUEUE_HEADS = 128;  // power of two, multiple of 8 (a head's waves share an XCD)

template <int N, int VPL>
struct Geom {
    static_assert(VPL == 4 || VPL == 2, "voxels per lane");
    static constexpr int V = N * N * N;
    static constexpr int QX = N / VPL;               // quads (a lane's VPL voxels) per x-row
    static constexpr int QUADS = V / VPL;
    static constexpr int LAYER_QUADS = QX * N;       // quads per z-layer
    static constexpr int WPC = QUADS / 64;           // wave units per chunk at 4 voxels per lane: 2 (8^3), 16 (16^3), 128 (32^3)
    static_assert(QUADS % 64 == 0, "whole waves");
    // A unit is a BRICK of 2 quads x 8 rows x 4 layers (8 x 8 x 4 voxels at 4 per lane, 4 x 8 x 4 at 2): compact in every direction, so
    // that the band shell cuts few of them whatever the viewing direction (a z-layer of 16 x 16 x 1 is cut by every frame that looks
    // along x or y), and the unit of brick_kernel's depth test (kernels_cull.h: BRICK_X / _Y / _Z).
    // Lanes 2i, 2i + 1 hold 2 * VPL * 4 contiguous bytes of one x-row per array.
    static constexpr int BX = 2 * VPL, BY = 8, BZ = 4;
    static constexpr int NBX = N / BX, NBY = N / BY, NBZ = N / BZ;
    static_assert(NBX * NBY * NBZ == WPC && NBX >= 1 && NBY >= 1 && NBZ >= 1, "bricks tile the chunk");
    static constexpr int CELL = N / 4;               // voxels per cell edge
    static_assert(BZ <= 2 * CELL, "a brick's cells lie in one 32-bit half of the 64-bit cell mask");
    static constexpr int WPB = INTEGRATE_WPB;         // waves per workgroup
    static constexpr int BLOCK = 64 * WPB;
    static constexpr int GRID = 256 * INTEGRATE_BLOCKS_PER_CU * (4 / WPB);  // persistent grid: what is resident at once
    static constexpr int GRID_STEP = (8 * WPC / WPB > 32) ? 8 * WPC / WPB : 32;  // blocks: every XCD's share of the first round is whole chunks
    static_assert(GRID % GRID_STEP == 0, "the statically dealt units are whole chunks");
    static_assert(WPB == 1 || WPB == 2 || WPB == 4, "waves per workgroup");
    static_assert(INTEGRATE_GRID_CAP % GRID_STEP == 0, "largest grid");
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

// a lane's voxels: VPL floats / packed colours, moved as one 16- or 8-byte access
template <int VPL>
struct alignas(4 * VPL) QuadFT { float v[VPL]; };
template <int VPL>
struct alignas(4 * VPL) QuadUT { unsigned v[VPL]; };
template <int VPL>
__device__ inline float &f4(QuadFT<VPL> &q, int j) { return q.v[j]; }
template <int VPL>
__device__ inline unsigned &u4(QuadUT<VPL> &q, int j) { return q.v[j]; }
__device__ inline unsigned wave_count(bool p) { return (unsigned)__popcll(__ballot(p)); }  // active lanes with p, wave-uniform

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
                bbox_include(M.mesh_ctl, x, y, z);
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

// The slot of a chunk that had none when the work-list was built, for a wave that integrated one of its voxels: the first
// such wave allocates (ChunkManager::CreateChunk), the others wait for its verdict.  The wait is on a wave that is running.
// Lane 0 only; returns the slot or -1; *created = 1 for the allocating wave.
// -> the slot (or -1), with CLAIM_CREATED set when THIS call allocated it (a flag in the result, not an out-parameter: the address of a
// local handed to a function that is not inlined puts that local into scratch memory -- a store, a wait and a load per call)
constexpr int CLAIM_CREATED = 1 << 30;
__device__ __attribute__((noinline)) int claim_slot(const MapView *__restrict__ Mc, ItemSync *sy, int x, int y, int z) {
    int s = atomicCAS(&sy->slot, 0, 1);
    if (s == 0) {
        const int slot = create_chunk(Mc, x, y, z);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // hash entry written before the slot can be seen
        atomicExch(&sy->slot, slot >= 0 ? slot + 2 : -1);
        return slot >= 0 ? (slot | CLAIM_CREATED) : slot;
    }
    for (int spin = 0; s == 1 && spin < (1 << 22); spin++) {
        __builtin_amdgcn_s_sleep(8);
        s = atomicOr(&sy->slot, 0);
    }
    if (s == 1) raise_error(Mc->error_flag, 1);  // the allocating wave never published (a second, pre-empted?): the map is incomplete, say so
    return s >= 2 ? s - 2 : -1;
}

template <int N, bool COLOR, bool SAMECAM, int VPL0>
#ifndef INTEGRATE_WAVES2
#define INTEGRATE_WAVES2 8  // the instantiations with 2 voxels per lane need 56 vector registers: eight waves per SIMD once the compiler also keeps
                            // to the 80 scalar registers that go with them (it derives that cap from this bound; 14 cold values go to lanes of a vector register)
#endif
#ifdef INTEGRATE_SGPRS
#define INTEGRATE_SGPR_ATTR __attribute__((amdgpu_num_sgpr(INTEGRATE_SGPRS)))
#else
#define INTEGRATE_SGPR_ATTR
#endif
__global__ __launch_bounds__(64 * INTEGRATE_WPB, (VPL0 == 2 ? INTEGRATE_WAVES2 : INTEGRATE_WAVES)) INTEGRATE_SGPR_ATTR void integrate_kernel(IntegrateParams P, MapView M, const MapView *__restrict__ Mc,
                                                                          const WorkItem *__restrict__ items,
                                                                          const FrameBox *__restrict__ boxes, ItemSync *sync,
                                                                          const int *__restrict__ work_count, int *queues,
                                                                          int max_items, int split, int lseq, const unsigned short *__restrict__ brick_masks) {
    using G = Geom<N, VPL0>;   // the launch's own granularity: what the grid and the queue heads are laid out for
    using GF = Geom<N, 2>;     // the fine one (2 voxels per lane) of the items behind `split`
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Blocks b and b + 8 share an XCD (observed dispatch order; speed only).  The first round is dealt statically: XCD x takes
    // chunks x, x + 8, ... of the (cost-ordered) work-list, all units of a chunk on one XCD, whose L2 then holds the pixel
    // footprint of "its" chunks only -- and every XCD gets the same mix of expensive and cheap chunks.
    const int nb = (int)gridDim.x;  // multiple of 8 and of 2 * WPC
    const int xcd = (int)blockIdx.x & 7;
    const int local_unit = ((int)blockIdx.x >> 3) * G::WPB + wave;  // this wave among its XCD's
    // The recompute in front of this launch will be emitted again (chisel_device.h: MC_LATCH): the launch leaves the map alone and the host
    // replays it afterwards.  (Requested together with the item count: one scalar wait for both.)
#ifdef CHISEL_NO_LATCH  // (A/B build only: what the word costs; CHISEL_HIP_DEFER_TOTALS=0 must go with it)
    const int latch = 0;
#else
    const int latch = M.mesh_ctl ? M.mesh_ctl[MC_LATCH] : 0;
#endif
    int n_items = *work_count;
    // this launch has started: everything queued in front of it on the map's stream is over (the host's substitute for events on that
    // stream, chisel_hip.hip: launch_seq; pinned memory, one thread -- also of a launch that leaves at once)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        reinterpret_cast<volatile int *>(M.error_flag)[4] = lseq;
        reinterpret_cast<volatile int *>(M.error_flag)[7] = M.committed - *M.free_top;  // slots in use (what a growable pool's host looks at)
    }
    if (latch) return;
    if (n_items > max_items) n_items = max_items;
    const int total = n_items * G::WPC;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // what the host sizes a later launch from: the number of work items, and the number of (item, frame) pairs as the order
        // kernel's cost classes give it (0 when the list was not ordered): chains of few frames are not worth 2 voxels per lane
        volatile int *report = reinterpret_cast<volatile int *>(M.error_flag);
        const int *cls = work_count - COUNT_ITEMS + COUNT_CLASS0;
        int pairs = 0;
        for (int c = 0; c < 8; c++) pairs += cls[c] * (KMAX - 2 * c);
        report[2] = n_items;
        report[3] = pairs;
        // the totals of the next mesh recompute start from zero (no recompute is in flight while this kernel runs: same stream)
        if (M.mesh_ctl) {
            M.mesh_ctl[0] = 0;
            M.mesh_ctl[1] = 0;
            M.mesh_ctl[2] = 0;
        }
    }
    // ... and so do the cursors of its record lists (kernels_mesh.h: MC_CURSORS = 8 ints in, MESH_PARTS = 64 64-bit words)
    if (blockIdx.x == 0 && threadIdx.x < 64 && M.mesh_ctl) reinterpret_cast<unsigned long long *>(M.mesh_ctl + 8)[threadIdx.x] = 0ull;
    const int grid_waves = nb * G::WPB;
#if defined(CHISEL_PHASES) && defined(PHASE0_AT)
#define PHASE0(at, dep) do { if (PHASE0_AT == at) { asm volatile("" ::"s"(dep)); PHASE(0); } } while (0)  // diagnostic: where inside the "item" stage the first stamp sits
#else
#define PHASE0(at, dep) do { } while (0)
#endif
    // Two granularities in one launch (VPL0 == 4): the items behind `split` -- the tail of the cost-ordered list -- run with 2 voxels per
    // lane: twice as many units of half the length, so that the stretch in which the chip drains (one unit long) is half as long.  Only
    // when the grid covers every unit statically (sized by the host from a recent launch's item count); a launch that turns out
    // larger than that takes the one granularity and the queue heads, as before.
    int n_coarse_x = 0, units_x = 0;  // this XCD's chunks x, x + 8, ...: how many of them are coarse; its units in all
    bool mixed = false;
    if (VPL0 == 4 && split >= 0 && split < n_items) {
        const int u0 = ((split + 7) / 8) * G::WPC + ((n_items + 7) / 8 - (split + 7) / 8) * GF::WPC;  // XCD 0 holds the most
        mixed = u0 <= (nb / 8) * G::WPB;
        if (mixed) {
            n_coarse_x = split > xcd ? (split - xcd + 7) / 8 : 0;
            const int n_x = n_items > xcd ? (n_items - xcd + 7) / 8 : 0;
            units_x = n_coarse_x * G::WPC + (n_x - n_coarse_x) * GF::WPC;
        }
    }
    const int rem_chunks = mixed ? 0 : n_items - grid_waves / G::WPC;  // chunks behind the statically dealt ones
    const IntegratorParams &ip = P.ip;
    unsigned t_sdf = 0, t_col = 0, t_colsat = 0, t_probe = 0, t_carved = 0;  // per lane; summed over the wave when it retires
    unsigned n_new = 0, n_updated = 0;                                       // wave-uniform
    int shard_try = 0;
#ifdef CHISEL_PHASES
    unsigned long long ph_last_start = 0, ph_max_unit = 0, ph_exec0 = 0;
    int ph_wid = 0;
    unsigned long long ph_visit = 0, ph_exec = 0, ph_exec_t = 0, ph_units = 0, ph_band = 0, ph_fr_t = 0;
    unsigned long long ph_f[5] = {0, 0, 0, 0, 0}, ph_ft = 0;
#define FSTAMP(i, dep) do { asm volatile("" ::"v"(dep)); const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); ph_f[i] += n_ - ph_ft; ph_ft = n_; } while (0)
    unsigned long long ph_need_l = 0, ph_vox = 0, ph_band_l = 0, ph_carve_w = 0, ph_carve_l = 0, ph_bandcarve_v = 0;
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, ph_t = __builtin_amdgcn_s_memrealtime(), ph_t0 = ph_t;
#define PHASE(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); ph[i] += n_ - ph_t; ph_t = n_; } while (0)
#else
#define PHASE(i) do { } while (0)
#define FSTAMP(i, dep) do { } while (0)
#endif

    // one unit = (work item, 64-quad group) at VPL voxels per lane; `return` = done with it
    auto run_unit = [&](auto vpl_tag, const int it, const int wq, const int wid) {
            constexpr int VPL = decltype(vpl_tag)::value;
            using G = Geom<N, VPL>;
            using QuadF = QuadFT<VPL>;
            using QuadU = QuadUT<VPL>;
            PHASE0(1, n_items);
            const WorkItem wi = items[it];
            // lane k: the cull kernel's flags of (chunk, frame k), requested together with the work item
            int cr_flags = 0;
            if (lane < P.n_frames) cr_flags = boxes[(size_t)it * P.n_frames + lane].flags;
            const int cxi = __builtin_amdgcn_readfirstlane(wi.x), cyi = __builtin_amdgcn_readfirstlane(wi.y),
                      czi = __builtin_amdgcn_readfirstlane(wi.z);
            int slot = __builtin_amdgcn_readfirstlane(wi.slot);
            unsigned mask = (unsigned)__builtin_amdgcn_readfirstlane((int)wi.frame_mask);
            // the brick of this unit and the lane's quad in it
            const int bx = wq % G::NBX, by = (wq / G::NBX) % G::NBY, bz = wq / (G::NBX * G::NBY);  // wave-uniform
            const int vx0 = bx * G::BX + (lane & 1) * VPL, vy = by * G::BY + ((lane >> 1) & 7), vz = bz * G::BZ + (lane >> 4);
            // the frames that can touch THIS brick: the cull kernel's depth test at brick scale (kernels_cull.h, "the brick phase"), requested
            // together with the work item.  At 2 voxels per lane a unit is one half of such a brick and takes its mask.
            constexpr int B4X = (VPL == 4) ? 1 : 2;  // units per brick along x
            const int brick = (bz * G::NBY + by) * (G::NBX / B4X) + bx / B4X;
            const unsigned unit_frames = (unsigned)__builtin_amdgcn_readfirstlane((int)brick_masks[(size_t)it * BrickGrid<N>::PER_CHUNK + brick]);
            PHASE0(2, (int)unit_frames);
            if (slot >= 0 && (mask & unit_frames) == 0u) return;  // a resident chunk, and no frame of the launch can touch this brick
            if (slot == SLOT_LOOKUP) {
                // the previous batch may have created this chunk while the work-list was built: it has finished now
                // (a sibling wave of THIS launch may have allocated it meanwhile -- its claim on the item precedes its hash entry,
                // so an entry seen together with a claim means "absent when the batch began")
                int s = 0;
                if (lane == 0) {
                    s = find_chunk(Mc, cxi, cyi, czi);
                    if (s >= 0 && atomicOr(&sync[it].slot, 0) != 0) s = -1;
                }
                slot = __builtin_amdgcn_readfirstlane(s);
                if (slot < 0) {
                    // not resident: frames before the first one that may integrate could only carve, i.e. do nothing
                    const unsigned inband = (unsigned)__builtin_amdgcn_readfirstlane((int)wi.inband_mask);
                    if (inband == 0u) return;
                    mask &= ~((1u << __builtin_ctz(inband)) - 1u);
                }
            }
            if (mask == 0u) return;  // (the item's mask: every unit of the item takes this exit, or none)
            const bool existed = slot >= 0;  // memory of `slot` holds this chunk's voxels
            ItemSync *sy = sync + it;
            // from here on the unit's own frames.  A unit of a chunk without a slot that has none still counts itself in below: the
            // unit that arrives last settles the chunk's `probe` figure
            mask &= unit_frames;
            if (mask == 0u && existed) return;

            // voxelCenter = centroids[i] + origin (ChunkManager.cpp:61: Vec3(x,y,z)*res + half; ProjectionIntegrator.h:63),
            // origin = numVoxels * ID (int) * resolution (Chunk.cpp:43)
            float wx[VPL], wy, wz;
            {
                const float ox = (float)(N * cxi) * ip.res, oy = (float)(N * cyi) * ip.res, oz = (float)(N * czi) * ip.res;
                wy = ((float)vy * ip.res + ip.half_res) + oy;
                wz = ((float)vz * ip.res + ip.half_res) + oz;
#pragma unroll
                for (int j = 0; j < VPL; j++) wx[j] = ((float)(vx0 + j) * ip.res + ip.half_res) + ox;
            }
#if !defined(PHASE0_AT)
            PHASE(0);
#endif
#ifdef CHISEL_PHASES
            ph_units++;
            ph_last_start = ph_t;
            ph_exec0 = ph_exec;
            ph_wid = wid;
#endif
            // default voxels: DistVoxel() DistVoxel.cpp:27-31, ColorVoxel() ColorVoxel.cpp:27-31
            QuadF s4, w4;
            QuadU c4;
#pragma unroll
            for (int j = 0; j < VPL; j++) {
                s4.v[j] = 99999.0f;
                w4.v[j] = 0.0f;
                c4.v[j] = 0u;
            }
            // per-lane flags, kept in one vector register (as lane masks they would cost eight scalar registers):
            // HAVE / HAVEC: sdf+weight / colour registers hold the chunk's values (a chunk without a slot has default voxels:
            // nothing to read); DCHG / CCHG: they differ from memory
            constexpr unsigned HAVE = 1u, HAVEC = 2u, DCHG = 4u, CCHG = 8u;
            unsigned st = existed ? 0u : (HAVE | HAVEC);
            // voxel addresses as a wave-uniform base (the slot: scalar registers) plus a 32-bit lane offset (one vector register
            // instead of three 64-bit addresses)
            const size_t slot_base = (size_t)(existed ? slot : 0) * G::V;
            const unsigned lane_off = (unsigned)((vz * N + vy) * N + vx0);
            unsigned bm = 0u, cm = 0u;  // frames in which this wave integrated / changed a voxel (wave-uniform)
            int carve_v = 0;            // lane k: this wave's carve tests of frame k (items without a slot)

            // Frames of the mask: each can touch this brick (brick_kernel's conservative test).  A lane whose own cell
            // it cannot touch reads the all-NaN record instead of a pixel and fails every test.
            // A frame has two halves.  project(): the lane's voxels in the frame's camera, their pixel records requested -- geometry only,
            // nothing of it depends on the voxels' state.  apply(): verdicts and updates, in frame order.  INTEGRATE_PIPE: the records of
            // frame k + 1 are requested before frame k is applied, so that their round trip runs beside frame k's arithmetic (for launches
            // that do not fill the chip: what such a launch takes is the length of its units' chains, and registers are not scarce there).
            struct Proj {
                int k, flags;         // wave-uniform
                bool need;            // this lane's cell can be touched by the frame
                float pcz[VPL];       // camera z of the lane's voxels
                unsigned off[VPL];    // byte offsets of their records (0: the all-NaN record)
                PixelRec r[VPL];
            };
            auto project = [&](const int k) -> Proj {
                Proj pj;
                pj.k = k;
#ifdef CHISEL_PHASES
                ph_ft = __builtin_amdgcn_s_memrealtime();
                ph_fr_t = ph_ft;
                ph_visit++;
#endif
                const int flags = __builtin_amdgcn_readlane(cr_flags, k);
                pj.flags = flags;
                const FrameCam &F = P.f[k];
                const CameraParams &C = F.cam;
                // inCamera = R^T * (voxelCenter - t) (ProjectionIntegrator.h:64), row i of R^T summed as a0 + (a1 + a2)
                const float dy = wy - C.t[1], dz = wz - C.t[2];
                const float s2 = C.R[5] * dy + C.R[8] * dz;
                float dx[VPL];
                float (&pcz)[VPL] = pj.pcz;
#pragma unroll
                for (int j = 0; j < VPL; j++) {
                    dx[j] = wx[j] - C.t[0];
                    pcz[j] = C.R[2] * dx[j] + s2;
                }
                const bool need = true;  // (every lane of a unit that visits a frame looks its pixels up: the brick is the unit of the depth test)
                pj.need = need;
                FSTAMP(0, (int)need);
                // the quad's state, at the first frame that can touch it (straight into the tuples: no use, no wait)
                if (need && !(st & HAVE)) {
                    s4 = *reinterpret_cast<const QuadF *>((M.sdf + slot_base) + lane_off);
                    w4 = *reinterpret_cast<const QuadF *>((M.wgt + slot_base) + lane_off);
                    st |= HAVE;
                }
                if (COLOR && need && !(st & HAVEC)) {
                    c4 = *reinterpret_cast<const QuadU *>((M.rgbw + slot_base) + lane_off);
                    st |= HAVEC;
                }
                unsigned (&off)[VPL] = pj.off;
                PixelRec (&r)[VPL] = pj.r;
                const float s0 = C.R[3] * dy + C.R[6] * dz;
                const float s1 = C.R[4] * dy + C.R[7] * dz;
                // PinholeCamera::ProjectPoint (PinholeCamera.cpp:38-45): invZ = 1.0f / z
                float inv_z[VPL];
                if (flags & WI_FASTZ) {  // wave-uniform
#pragma unroll
                    for (int j = 0; j < VPL; j++) inv_z[j] = reciprocal_in_range(pcz[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < VPL; j++) inv_z[j] = 1.0f / pcz[j];
                }
                // ---- geometry + projection -> record of every voxel of the quad, the four gathers in flight together --------
                // Record offsets are 32 bits against a scalar base; the base is the all-NaN record in front of the frame's image,
                // which a voxel that is off the image (or not wanted) reads: it fails the band and the carve test like a skipped
                // pixel.  IsPointOnImage (PinholeCamera.cpp:61-64) is 0 <= u < W && 0 <= v < H, and the integrator skips z < 0
                // (ProjectionIntegrator.h:68 / :126).  With iu = floor(u) (as int, saturating; INT_MIN for NaN) the image test is
                // (unsigned)iu < W && (unsigned)iv < H -- there floor(u) == (int)u (:72 / :131) -- and z == +-0 or NaN gives
                // u, v = +-inf / NaN, which fail it, so "z > 0" is the remaining predicate.  The three tests are chained through
                // selects (no lane-mask arithmetic).
                const char *rec_base = reinterpret_cast<const char *>(F.rec - 1);
                const unsigned row_bytes = (unsigned)C.W * (unsigned)sizeof(PixelRec);
                float zn[VPL];  // camera z, or -1 for a quad this frame cannot touch
                if (OPT_INSIDE && (flags & WI_INSIDE)) {  // wave-uniform
                    // Every voxel of the chunk lies in front of the camera and projects onto the image (cull_chunk_frame): the three
                    // tests hold, the offset is row * row_bytes + (col + 1) * 8.  A lane this frame cannot touch (`need`, from the cull
                    // kernel's conservative bounds: its verdicts would fail anyway) reads the NaN record instead of a pixel.
                    const unsigned nm = need ? 0xffffffffu : 0u;
#pragma unroll
                    for (int j = 0; j < VPL; j++) {
                        const float pcx = C.R[0] * dx[j] + s0, pcy = C.R[1] * dx[j] + s1;
                        const float u = C.fx * pcx * inv_z[j] + C.cx;
                        const float v = C.fy * pcy * inv_z[j] + C.cy;
                        const unsigned iu = (unsigned)floor_to_int(u), iv = (unsigned)floor_to_int(v);
                        off[j] = (__umul24(iv, row_bytes) + ((iu << 3) + 8u)) & nm;
                        r[j] = *reinterpret_cast<const PixelRec *>(rec_base + off[j]);
                    }
                } else {
#pragma unroll
                for (int j = 0; j < VPL; j++) zn[j] = need ? pcz[j] : -1.0f;
#pragma unroll
                for (int j = 0; j < VPL; j++) {
                    const float pcx = C.R[0] * dx[j] + s0, pcy = C.R[1] * dx[j] + s1;
                    const float u = C.fx * pcx * inv_z[j] + C.cx;
                    const float v = C.fy * pcy * inv_z[j] + C.cy;
                    const int iu = floor_to_int(u), iv = floor_to_int(v);
                    const int iv_ok = ((unsigned)iu < (unsigned)C.W) ? iv : -1;
                    const float z_ok = ((unsigned)iv_ok < (unsigned)C.H) ? zn[j] : -1.0f;
                    // DepthAt(row, col) DepthImage.h:72-76
                    off[j] = (z_ok > 0.0f) ? __umul24((unsigned)iv, row_bytes) + ((unsigned)iu + 1u) * (unsigned)sizeof(PixelRec) : 0u;
#ifdef CHISEL_ABLATE_GATHER  // diagnostic (wrong results): what would the kernel cost if the record gathers were coalesced?
                    r[j] = *reinterpret_cast<const PixelRec *>(rec_base + (off[j] ? (unsigned)((lane * VPL + j + 1) * sizeof(PixelRec)) : 0u));
#else
                    r[j] = *reinterpret_cast<const PixelRec *>(rec_base + off[j]);
#endif
                }
                }
                FSTAMP(1, off[VPL - 1]);
                return pj;
            };
            // Branches are wave-uniform (__any) and the lanes predicated; a lane's verdicts are bit masks over its voxels and its counters
            // per-lane adds, summed over the wave once, when the wave retires: the scalar unit serves all four SIMDs of a CU at about half the
            // vector rate per SIMD, and lane-mask logic (one scalar AND / OR / popcount per predicate) made it as busy as the vector units.
            auto apply = [&](const Proj &pj) {
                const int k = pj.k, flags = pj.flags;
                const FrameCam &F = P.f[k];
                const float (&pcz)[VPL] = pj.pcz;
                const unsigned (&off)[VPL] = pj.off;
                const PixelRec (&r)[VPL] = pj.r;
                // ---- band tests: bit j of bandm / carvem = voxel j takes the in-band / the carve branch -------------------------
                // r.x is NaN for the pixels the reference skips (:74 depth > 50 / :134 isnan / :141 depth > 100): both tests fail.
                float sd[VPL];
                unsigned bandm = 0u, carvem = 0u;
#if OPT_CARVESKIP
#pragma unroll
                for (int j = 0; j < VPL; j++) sd[j] = r[j].x - pcz[j];                              // surfaceDist :79 / :139
                {
#pragma unroll
                    for (int j = 0; j < VPL; j++) bandm |= (fabsf(sd[j]) < r[j].y + ip.diag) ? (1u << j) : 0u;  // :81 / :144
                }
                if (ip.carving) {
#pragma unroll
                    for (int j = 0; j < VPL; j++) carvem |= (sd[j] > r[j].y + ip.carving_dist) ? (1u << j) : 0u;  // :86 / :164 (else branch)
                    carvem &= ~bandm;
                }
#else
#pragma unroll
                for (int j = 0; j < VPL; j++) {
                    sd[j] = r[j].x - pcz[j];                                                        // surfaceDist :79 / :139
                    bandm |= (fabsf(sd[j]) < r[j].y + ip.diag) ? (1u << j) : 0u;                    // :81 / :144
                    carvem |= (sd[j] > r[j].y + ip.carving_dist) ? (1u << j) : 0u;                  // :86 / :164 (else branch)
                }
                carvem = ip.carving ? (carvem & ~bandm) : 0u;
#endif
                FSTAMP(2, bandm | carvem);
                t_sdf += (unsigned)__popc(bandm);
#ifdef CHISEL_PHASES
                ph_need_l += __builtin_popcountll(__ballot(pj.need));
                for (int j = 0; j < VPL; j++) ph_vox += __builtin_popcountll(__ballot(off[j] != 0u));
                ph_band_l += __builtin_popcountll(__ballot(bandm != 0u));
                ph_carve_l += __builtin_popcountll(__ballot(carvem != 0u));
                for (int j = 0; j < VPL; j++) ph_bandcarve_v += __builtin_popcountll(__ballot(((bandm | carvem) >> j) & 1u));
                if (__any(carvem != 0u)) ph_carve_w++;
#endif
                // `probe`: carve tests on a chunk the reference's map holds before this frame
                if (existed) {
                    t_probe += (unsigned)__popc(carvem);
                } else {
                    unsigned frame_carve = 0u;  // wave-uniform
#pragma unroll
                    for (int j = 0; j < VPL; j++) frame_carve += wave_count((carvem >> j) & 1u);
                    carve_v = (lane == k) ? (int)frame_carve : carve_v;  // (a frame is visited once)
                }
                if (__any(bandm != 0u)) {
#ifdef CHISEL_PHASES
                    ph_band++;
#endif
                    bm |= 1u << k;
                    cm |= 1u << k;
                    // colour first: the pixels of the in-band voxels whose colour weight is below 8 are requested now (one 4-byte
                    // gather per voxel, all in flight) and consumed after the sdf arithmetic
                    unsigned cw[VPL], csh[VPL];
                    unsigned lastm = 0u;  // OPT_COLOR2: bit j = the word of voxel j was read one byte low (last pixel of a 3-channel image)
                    int cpix[VPL];
                    unsigned freshm = 0u;  // bit j: voxel j takes a colour sample
                    const bool word_gather = COLOR && F.color_channels >= 3;  // wave-uniform
                    if (COLOR) {
                        const unsigned image_bytes = (unsigned)(F.ccam.W * F.ccam.H * F.color_channels);
                        unsigned hasm = 0u;
#pragma unroll
                        for (int j = 0; j < VPL; j++) {
                            // one camera: the colour pixel is the depth pixel (its index back from the record offset; in band => on the image)
                            cpix[j] = SAMECAM ? (int)(off[j] / (unsigned)sizeof(PixelRec)) - 1
                                              : (((bandm >> j) & 1u) ? color_pixel(F.ccam, wx[j], wy, wz) : -1);
                            if (!SAMECAM) hasm |= (cpix[j] >= 0) ? (1u << j) : 0u;
                            freshm |= ((u4(c4, j) >> 24) < 8u) ? (1u << j) : 0u;  // colorVoxel.GetWeight() < 8, ProjectionIntegrator.h:152
                        }
                        if (SAMECAM) hasm = bandm;
                        else hasm &= bandm;
                        freshm &= hasm;
                        t_col += (unsigned)__popc(freshm);
                        if (!SAMECAM) t_colsat += (unsigned)__popc(hasm & ~freshm);  // one camera: colsat = sdf - col
                        if (OPT_COLOR2 && word_gather && __any(freshm != 0u)) {
                            // (the shift of the image's last pixel is kept as one bit per voxel: the words stay in registers
                            // through the sdf arithmetic, four shift counts beside them cost the kernel a wave per SIMD)
                            const unsigned last_word = image_bytes - 4u;
                            lastm = 0u;
#pragma unroll
                            for (int j = 0; j < VPL; j++) {
                                unsigned sh;
                                cw[j] = color_gather2(F.color, ((freshm >> j) & 1u) ? (unsigned)cpix[j] : 0u, (unsigned)F.color_channels, last_word, sh);
                                lastm |= sh ? (1u << j) : 0u;
                            }
                        } else if (word_gather && __any(freshm != 0u)) {
#pragma unroll
                            for (int j = 0; j < VPL; j++)
#ifdef CHISEL_ABLATE_GATHER
                                cw[j] = color_gather(F.color, ((freshm >> j) & 1u) ? lane * VPL + j : 0, F.color_channels, image_bytes, csh[j]);
#else
                                cw[j] = color_gather(F.color, ((freshm >> j) & 1u) ? cpix[j] : 0, F.color_channels, image_bytes, csh[j]);
#endif
                        }
                    }
                    // Integrate: voxel.Integrate(surfaceDist, 1.0f) :84; IntegrateColor: weighter->GetWeight(.., truncation) :161-162 --
                    // weight / (5 * truncation), an IEEE division (11 instructions) unless the cull kernel has established that the
                    // weight is 1 and 5 * truncation of every pixel under the chunk is in the range where the 4-instruction reciprocal
                    // is exact (WI_FASTWU; wave-uniform)
                    float wu[VPL];
                    if (COLOR && (flags & WI_FASTWU)) {
#pragma unroll
                        for (int j = 0; j < VPL; j++) wu[j] = reciprocal_in_range(5.0f * r[j].y);
                    } else {
#pragma unroll
                        for (int j = 0; j < VPL; j++) wu[j] = COLOR ? constant_weight(ip.weight, r[j].y) : 1.0f;
                    }
#pragma unroll
                    for (int j = 0; j < VPL; j++) {
                        float ns = f4(s4, j), nw = f4(w4, j);
                        dist_integrate(ns, nw, sd[j], wu[j]);
                        const bool in_band = ((bandm >> j) & 1u) != 0u;
                        f4(s4, j) = in_band ? ns : f4(s4, j);
                        f4(w4, j) = in_band ? nw : f4(w4, j);
                    }
                    st |= bandm ? DCHG : 0u;
                    if (COLOR && __any(freshm != 0u)) {
                        if (word_gather) {
#pragma unroll
                            for (int j = 0; j < VPL; j++) {
                                const unsigned nc = OPT_COLOR2 ? color_integrate_fresh_bgr(u4(c4, j), cw[j] >> (((lastm >> j) & 1u) * 8u))
                                                               : color_integrate_fresh(u4(c4, j), color_word(cw[j], csh[j]));
                                u4(c4, j) = ((freshm >> j) & 1u) ? nc : u4(c4, j);
                            }
                        } else {  // 1 / 2 channel images
#pragma unroll
                            for (int j = 0; j < VPL; j++) {
                                if ((freshm >> j) & 1u) {
                                    unsigned cbits = u4(c4, j);
                                    uchar4 cv = *reinterpret_cast<uchar4 *>(&cbits);
                                    uint8_t cr, cg, cb;
                                    color_at(F.color, cpix[j], F.color_channels, cr, cg, cb);
                                    cv = color_integrate(cv, cr, cg, cb, 1);
                                    u4(c4, j) = *reinterpret_cast<unsigned *>(&cv);
                                }
                            }
                        }
                        st |= freshm ? CCHG : 0u;
                    }
                }
                FSTAMP(3, f4(s4, 0));
                // (a voxel that takes the carve branch is only touched if it holds a weight: most do not -- free space in front of the band
                // was never integrated -- and then nothing below can hit)
                bool may_hit = false;  // wave-uniform
                if (__any(carvem != 0u)) {
                    unsigned heldm = 0u;
#pragma unroll
                    for (int j = 0; j < VPL; j++) heldm |= (f4(w4, j) > 0.0f) ? (1u << j) : 0u;
                    may_hit = __any((carvem & heldm) != 0u);
                }
                if (may_hit) {
                    unsigned hitm = 0u;
#pragma unroll
                    for (int j = 0; j < VPL; j++) {
                        const bool hit = ((carvem >> j) & 1u) && (f4(w4, j) > 0.0f) && sdf_below_carve_threshold(f4(s4, j));
                        hitm |= hit ? (1u << j) : 0u;
                        const bool decay = COLOR && !(f4(w4, j) < 5.0f);      // :166-177: decay
                        const float cw_ = decay ? f4(w4, j) - 1.0f : 0.0f;    // else :88-95 / :170 Carve() == Reset()
                        const float cs_ = decay ? f4(s4, j) : 99999.0f;
                        f4(w4, j) = hit ? cw_ : f4(w4, j);
                        f4(s4, j) = hit ? cs_ : f4(s4, j);
                    }
                    t_carved += (unsigned)__popc(hitm);
                    st |= hitm ? DCHG : 0u;
                    if (__any(hitm != 0u)) cm |= 1u << k;
                }
#ifdef CHISEL_PHASES
                FSTAMP(4, f4(s4, 0));
                ph_exec++;
                ph_exec_t += __builtin_amdgcn_s_memrealtime() - ph_fr_t;
#endif
            };
#if INTEGRATE_PIPE
            if (mask) {  // (a unit of a chunk without a slot may have no frame of its own: it only counts itself in below)
                Proj cur = project(__builtin_ctz(mask));
                mask &= mask - 1u;
                while (true) {
                    const bool more = mask != 0u;  // wave-uniform
                    Proj nxt = cur;
                    if (more) {
                        nxt = project(__builtin_ctz(mask));
                        mask &= mask - 1u;
                    }
                    apply(cur);
                    if (!more) break;
                    cur = nxt;
                }
            }
#else
            while (mask) {
                const int k = __builtin_ctz(mask);
                mask &= mask - 1u;
                apply(project(k));
            }
#endif

            PHASE(1);
            // ---- the unit's results -------------------------------------------------------------------------------------
            if (!existed) {
                // Carving cannot change a default voxel, so a wave that never integrated has nothing to write; one that did
                // needs the chunk's slot.  The reference creates the chunk before and erases it after every frame in which
                // no voxel integrates (Chisel.h:133-143, 202-207): the outcome is "exists from the first such frame on".
                if (bm) {
                    int s = 0;
                    if (lane == 0) s = claim_slot(Mc, sy, cxi, cyi, czi);
                    s = __builtin_amdgcn_readfirstlane(s);
                    const bool created = s >= 0 && (s & CLAIM_CREATED) != 0;
                    slot = s >= 0 ? (s & ~CLAIM_CREATED) : s;
                    n_new += created ? 1u : 0u;
                }
                // deposit what the chunk-level `probe` figure needs, then count this wave in; the last one settles it.
                // Returning atomics: the arrival is issued only after the deposits have been performed.
                unsigned seen = 0u;
                if (lane == 0 && bm) seen += atomicOr(&sy->band, bm);
                if (lane < KMAX && carve_v) seen += atomicAdd(&sy->carve[lane], (unsigned)carve_v);
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(seen)::"memory");
                unsigned last = 0u;
                if (lane == 0) last = (atomicAdd(&sy->arrived, 1u) == (unsigned)(G::WPC - 1)) ? 1u : 0u;
                if (__builtin_amdgcn_readfirstlane((int)last)) {
                    unsigned band_all = 0u, cnt = 0u;
                    if (lane == 0) band_all = atomicOr(&sy->band, 0u);
                    if (lane < KMAX) cnt = atomicAdd(&sy->carve[lane], 0u);
                    band_all = (unsigned)__builtin_amdgcn_readfirstlane((int)band_all);
                    // resident before frame k  <=>  some voxel integrated in a frame < k
                    const bool counts = band_all != 0u && lane < KMAX && lane > __builtin_ctz(band_all | 0x80000000u);
                    unsigned v = counts ? cnt : 0u;
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) v += __shfl_down(v, o);
                    if (lane == 0) t_probe += v;  // (the counters are per lane)
                }
                PHASE(2);
                if (slot < 0) return;  // never integrated here, or no slot left (error raised)
            }
            unsigned signs = 0u;  // SUM_POS | SUM_NEG over the observed voxels this wave writes (wave-uniform)
            {
                const size_t out_base = (size_t)slot * G::V;  // wave-uniform
                if (st & DCHG) {
                    *reinterpret_cast<QuadF *>((M.sdf + out_base) + lane_off) = s4;
                    *reinterpret_cast<QuadF *>((M.wgt + out_base) + lane_off) = w4;
                }
                if (COLOR && (st & CCHG)) *reinterpret_cast<QuadU *>((M.rgbw + out_base) + lane_off) = c4;
                bool pos = false, neg = false;
#pragma unroll
                for (int j = 0; j < VPL; j++) {
                    const bool seen = (st & DCHG) && f4(w4, j) > 0.5f;
                    neg = neg || (seen && f4(s4, j) < 0.0f);
                    pos = pos || (seen && !(f4(s4, j) < 0.0f));
                }
                signs = (__any(pos) ? SUM_POS : 0u) | (__any(neg) ? SUM_NEG : 0u);
            }
            if (cm) {
                // "needsUpdate" of the chunk per frame (Chisel.h:85 / :167): each frame counts once per chunk -- by the wave
                // whose OR sets its bit first -- and the slot is marked for the mesher (Chisel.h:175-189)
                unsigned old = 0u;
                int newly_dirty = 0;
                if (lane == 0) {
                    old = atomicOr(&sy->changed, cm);
                    if (old == 0u) newly_dirty = mark_slot_dirty(M, slot) ? 1 : 0;  // (the first unit of the chunk to report a change in this launch: its siblings need not repeat it)
                    if (signs) atomicOr(&slot_summary(M)[slot], signs);  // (a wave that wrote a voxel changed one: cm != 0)
                }
                old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
                n_updated += (unsigned)__popc(cm & ~old);
                // first update of this chunk since the last mesh recompute: its 27-neighbourhood joins the recompute's job list now
                // (Chisel.h:175-189), so that the recompute needs no pass over the dirty slots of its own
                if (__builtin_amdgcn_readfirstlane(newly_dirty)) mesh_expand_dirty(Mc, slot, cxi, cyi, czi, lane);
            }
    };
    if (mixed) {  // one unit per wave, all of them dealt statically
        if (local_unit < units_x) {
            const int u4 = n_coarse_x * G::WPC;
            if (local_unit < u4) run_unit(std::integral_constant<int, VPL0>{}, (local_unit / G::WPC) * 8 + xcd, local_unit % G::WPC, local_unit);
            else run_unit(std::integral_constant<int, 2>{}, (n_coarse_x + (local_unit - u4) / GF::WPC) * 8 + xcd, (local_unit - u4) % GF::WPC, local_unit);
        }
        PHASE(3);
    }
    int wid = mixed ? total : ((local_unit / G::WPC) * 8 + xcd) * G::WPC + local_unit % G::WPC;
    while (wid < total) {
        run_unit(std::integral_constant<int, VPL0>{}, wid / G::WPC, wid % G::WPC, wid);
        PHASE(3);
#ifdef CHISEL_PHASES
        if (ph_last_start && ((ph_t - ph_last_start) << 32) > ph_max_unit) ph_max_unit = ((ph_t - ph_last_start) << 32) | ((unsigned long long)(ph_wid & 0xffffff) << 8) | (ph_exec - ph_exec0);
#endif

        // ---- next unit: this block's home queue head, then its neighbour's.  A head serves the waves of 1 / QUEUE_HEADS of the
        // blocks (same XCD); one returning atomic per unit, <= 64 waves per head: a single word saturates near 90 atomics / us,
        // and eight heads for 8192 waves cost every wave 10-20 us per unit.
        wid = total;
        while (rem_chunks > 0 && shard_try < 2) {
            const int heads = nb < QUEUE_HEADS ? nb : QUEUE_HEADS;  // every head is some workgroup's home (nb is a multiple of 8)
            const int h = ((int)blockIdx.x + shard_try * 8) % heads;
            const int units = rem_chunks > h ? ((rem_chunks - h + heads - 1) / heads) * G::WPC : 0;  // chunks h, h + heads, ... of the remainder
            int t = units;
            if (lane == 0 && units > 0) t = atomicAdd(&queues[h * QUEUE_STRIDE], 1);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t < units) {
                wid = grid_waves + ((t / G::WPC) * heads + h) * G::WPC + (t % G::WPC);
                break;
            }
            shard_try++;
        }
        PHASE(4);
    }

#ifdef CHISEL_PHASES
    if (lane == 0) {
        unsigned long long *row = M.block_counters + (size_t)(blockIdx.x & (INTEGRATE_MAX_GRID - 1)) * 16;
        for (int i = 0; i < 5; i++) atomicAdd(&row[9 + i], ph[i]);
        atomicAdd(&row[14], ph_t - ph_t0);
        atomicAdd(&row[15], 1ull);
        unsigned long long *row2 = M.block_counters + (size_t)INTEGRATE_MAX_GRID * 16 + (size_t)(blockIdx.x & (INTEGRATE_MAX_GRID - 1)) * 16;
        atomicMax(&row2[5], ~ph_t0); atomicMax(&row2[6], ph_t); atomicMax(&row2[7], ph_last_start); atomicMax(&row2[8], ph_max_unit);
        if (wave == 0) { row2[9] = ph_t0; row2[10] = ph_t; row2[11] = ph_units; row2[12] = ph_exec; }
        atomicAdd(&row2[0], ph_visit); atomicAdd(&row2[1], ph_exec); atomicAdd(&row2[2], ph_exec_t); atomicAdd(&row2[3], ph_units); atomicAdd(&row2[4], ph_band);
        unsigned long long *row3 = M.block_counters + (size_t)INTEGRATE_MAX_GRID * 16 + (size_t)(blockIdx.x & 63) * 16;  // rows 0-63, columns 13-15 (three packed pairs)
        atomicAdd(&row3[13], ph_need_l); atomicAdd(&row3[14], ph_vox); atomicAdd(&row3[15], ph_bandcarve_v);
        unsigned long long *row4 = M.block_counters + (size_t)INTEGRATE_MAX_GRID * 16 + (size_t)(64 + (blockIdx.x & 63)) * 16;
        atomicAdd(&row4[13], ph_band_l); atomicAdd(&row4[14], ph_carve_l); atomicAdd(&row4[15], ph_carve_w);
        unsigned long long *row5 = M.block_counters + (size_t)INTEGRATE_MAX_GRID * 16 + (size_t)(128 + (blockIdx.x & 63)) * 16;
        atomicAdd(&row5[13], ph_f[0]); atomicAdd(&row5[14], ph_f[1]); atomicAdd(&row5[15], ph_f[2]);
        unsigned long long *row6 = M.block_counters + (size_t)INTEGRATE_MAX_GRID * 16 + (size_t)(192 + (blockIdx.x & 63)) * 16;
        atomicAdd(&row6[13], ph_f[3]); atomicAdd(&row6[14], ph_f[4]);
    }
#endif
    // ---- counters: one no-return atomic per wave and counter into this block's private row (rows are summed lazily by
    // reduce_counters_kernel).  Nothing waits for the atomics.
    unsigned sums[5] = {t_sdf, t_col, (COLOR && SAMECAM) ? (t_sdf - t_col) : t_colsat, t_probe, t_carved};
#pragma unroll
    for (int c = 0; c < 5; c++)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sums[c] += __shfl_down(sums[c], o);
    if (lane == 0) {
        unsigned long long *row = M.block_counters + (size_t)(blockIdx.x & (INTEGRATE_MAX_GRID - 1)) * 16;
        const unsigned vals[8] = {sums[0], sums[1], sums[2], sums[3], sums[4], 0u, n_new, n_updated};
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (vals[k]) atomicAdd(&row[k], (unsigned long long)vals[k]);
        if (blockIdx.x == 0 && wave == 0) {
            atomicAdd(&row[5], (unsigned long long)n_items);
            atomicAdd(&row[8], (unsigned long long)P.n_frames);
        }
    }
}

// sums the per-block rows into counters[] (CHISEL_HIP_NUM_COUNTERS = 9 entries); one block of 256 threads
__global__ void reduce_counters_kernel(MapView M, int n_rows) {
    __shared__ unsigned long long s[256];
#ifdef CHISEL_PHASES
    constexpr int NK = 16;
#else
    constexpr int NK = 9;
#endif
    for (int k = 0; k < NK; k++) {
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
#ifdef CHISEL_PHASES
    if (threadIdx.x < 9) {
        unsigned long long v = 0ull;
        for (int r = 0; r < n_rows; r++) {
            const unsigned long long x = M.block_counters[(size_t)INTEGRATE_MAX_GRID * 16 + (size_t)r * 16 + threadIdx.x];
            if (threadIdx.x >= 5) v = x > v ? x : v;
            else v += x;
        }
        M.counters[16 + threadIdx.x] = v;
    }
#endif
}

}  // namespace chisel_hip
