// host_mesh.h -- mesh-side entry points of the C ABI (included by chisel_hip.hip).
//
// Chisel::UpdateMeshes (Chisel.cpp:50-59) -> ChunkManager::RecomputeMeshes (ChunkManager.cpp:130-169): the ids flagged
// since the last recompute are meshed on the GPU (kernels_mesh.h: mark -> collect -> count, which also allots each job its range
// of the triangle list -> emit into one arena).  The arena stays in HBM (MeshArena in chisel_hip.hip); only the per-chunk sizes travel to the host, the vertex
// data follows when a caller asks for a mesh.
namespace {

MeshParams mesh_params(const chisel_hip_map *m) {
    MeshParams P;
    P.res = m->cfg.voxel_resolution;
    P.half_res = m->cfg.voxel_resolution * 0.5f;                    // ChunkManager.cpp:52
    P.rf_chunk = 1.0f / ((float)m->N * m->cfg.voxel_resolution);    // ChunkManager.h:138-140
    P.rf_voxel = 1.0f / m->cfg.voxel_resolution;                    // Chunk.cpp:74
    P.use_color = m->cfg.use_color ? 1 : 0;
    P.stages = m->mesh_stages & m->tune.mesh_stage_mask;
    return P;
}

// publish: bit 0 = the totals go to the host (the first emission of a recompute), bit 1 = the job list kept by the integration kernels was
// this recompute's input and is empty from here on
void launch_mesh_triangles(chisel_hip_map *m, const MeshParams &P, float *arena, size_t arena_floats, int publish = 0) {
    MeshBuffers &B = m->mesh_buf;
    const JobInfo *bases = B.info;
    const int *totals = B.totals;
    int *host_info = m->mesh_info_dev;
    volatile int *host_flags = (volatile int *)m->mesh_totals_dev;
    const int max_jobs = std::min(MESH_INFO_PREFETCH, B.capacity), seq = m->mesh_seq, part = B.tri_capacity / MESH_PARTS;
    const dim3 grid(4096), block(MESH_TRI_BLOCK);  // persistent: the number of triangles (per partition of the list) is read on the device
    switch (m->N) {
        case 8: hipLaunchKernelGGL(mesh_triangle_kernel<8>, grid, block, 0, m->stream, m->view, P, B.jobs, bases, B.tris, B.corners, totals, B.cnt, part, arena, arena_floats, host_info, host_flags, max_jobs, seq, publish); break;
        case 16: hipLaunchKernelGGL(mesh_triangle_kernel<16>, grid, block, 0, m->stream, m->view, P, B.jobs, bases, B.tris, B.corners, totals, B.cnt, part, arena, arena_floats, host_info, host_flags, max_jobs, seq, publish); break;
        case 32: hipLaunchKernelGGL(mesh_triangle_kernel<32>, grid, block, 0, m->stream, m->view, P, B.jobs, bases, B.tris, B.corners, totals, B.cnt, part, arena, arena_floats, host_info, host_flags, max_jobs, seq, publish); break;
    }
}

int mesh_subjobs(const chisel_hip_map *m) { return m->N == 8 ? MeshGeom<8>::S : (m->N == 16 ? MeshGeom<16>::S : MeshGeom<32>::S); }
int mesh_row_ints(const chisel_hip_map *m) { return m->N == 8 ? MeshGeom<8>::ROW : (m->N == 16 ? MeshGeom<16>::ROW : MeshGeom<32>::ROW); }
int ensure_mesh_jobs(chisel_hip_map *m, int n) {
    MeshBuffers &B = m->mesh_buf;
    if (n <= B.capacity) return CHISEL_HIP_OK;
    HIP_TRY(hipStreamSynchronize(m->stream));
    for (void *p : {(void *)B.jobs, (void *)B.ids, (void *)B.info, (void *)B.cnt, (void *)B.job_acc})
        if (p) HIP_TRY(hipFree(p));
    B.jobs = nullptr; B.ids = nullptr; B.info = nullptr; B.cnt = nullptr; B.job_acc = nullptr;
    int cap = std::max(4096, B.capacity);
    while (cap < n) cap *= 2;
    HIP_TRY(hipMalloc(&B.jobs, (size_t)cap * sizeof(MeshJob)));
    HIP_TRY(hipMalloc(&B.ids, (size_t)cap * 3 * sizeof(int)));
    HIP_TRY(hipMalloc(&B.info, (size_t)cap * sizeof(JobInfo)));
    HIP_TRY(hipMalloc(&B.cnt, (size_t)cap * mesh_row_ints(m) * sizeof(unsigned)));
    HIP_TRY(hipMalloc(&B.job_acc, (size_t)cap * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(B.job_acc, 0, (size_t)cap * sizeof(unsigned long long), m->stream));  // (the count kernel's last arrivers keep them at zero)
    B.capacity = cap;
    return CHISEL_HIP_OK;
}

// device counters of a recompute (mesh_buf.totals)
enum { MT_TRIS = MC_TRIS, MT_GRIDS = MC_GRIDS, MT_OVERFLOW = MC_OVERFLOW, MT_JOBS = MC_JOBS };
int *mesh_totals(chisel_hip_map *m) { return m->mesh_buf.totals; }
// a recompute's totals and the cursors of its record lists start from zero: normally the first wave of the integration launch in
// front of it has seen to that (kernels_integrate.h); this is for the recomputes that have no such launch in front of them
hipError_t zero_mesh_counters(chisel_hip_map *m) {
    hipError_t e = hipMemsetAsync(mesh_totals(m), 0, 3 * sizeof(int), m->stream);  // keeps MT_JOBS and the kept list's length
    if (e == hipSuccess) e = hipMemsetAsync(mesh_totals(m) + MC_CURSORS, 0, 2 * MESH_PARTS * sizeof(int), m->stream);
    return e;
}

// The job list of a recompute: the resident chunks of the 27-neighbourhoods of the slots dirtied since the last one, de-duplicated
// through one flag per slot.  The integration kernels keep it as they go (mesh_expand_dirty: the wave that first dirties a slot
// appends its neighbourhood), so a recompute normally starts with its count kernel; mesh_mark_kernel builds the same entries from the
// dirty flags when something else has dirtied slots since (point clouds), when the kept list has been given up (`rebuild`), or
// completes it with `extra` host-side ids (neighbourhoods of chunks that were removed while dirty).  The number of entries stays on
// the device (mesh_ctl[4], copied to the totals by the count kernel); nothing here waits for the stream unless `extra` is used.
int collect_mesh_ids(chisel_hip_map *m, const std::vector<int> &extra) {
    MeshBuffers &B = m->mesh_buf;
    const int C = m->view.committed;
    int rc = ensure_mesh_jobs(m, C);  // (MeshJob / JobInfo records: worst case every resident chunk -- every slot that has memory)
    if (rc) return rc;
    int *n_jobs = mesh_totals(m) + MC_KEPT;
    B.n_jobs = n_jobs;
    B.ext_ids = nullptr;
    if (m->mesh_mark_needed) {
        hipLaunchKernelGGL(mesh_mark_kernel, dim3(1024), dim3(256), 0, m->stream, m->view, B.flags, m->view.mesh_jobs, n_jobs);
        m->mesh_mark_needed = false;
    }
    if (!m->mesh_totals_clean) {
        // (two recomputes without an integration launch in between: its first thread is what zeroes the totals otherwise)
        HIP_TRY(zero_mesh_counters(m));
    }
    m->mesh_totals_clean = false;
    if (!extra.empty()) {
        // (rare) ids kept on the host: the ones that are resident join the job list
        const int ne = (int)(extra.size() / 3);
        std::vector<int> slots;
        rc = lookup_slots(m, extra.data(), ne, slots);
        if (rc) return rc;
        int *d_slots = nullptr;
        HIP_TRY(hipMalloc(&d_slots, (size_t)ne * sizeof(int)));
        HIP_TRY(hipMemcpy(d_slots, slots.data(), (size_t)ne * sizeof(int), hipMemcpyHostToDevice));  // (pageable source, rare path: a blocking copy; lookup_slots has waited for the stream already)
        hipLaunchKernelGGL(mesh_append_kernel, dim3((ne + 255) / 256), dim3(256), 0, m->stream, m->view, B.flags, (const int *)d_slots, ne, m->view.mesh_jobs, n_jobs);
        HIP_TRY(hipStreamSynchronize(m->stream));
        HIP_TRY(hipFree(d_slots));
    }
    HIP_TRY(hipGetLastError());
    return CHISEL_HIP_OK;
}

// Arena buffers are recycled: a released one goes to a small pool (no hipFree, which would wait for the device) and
// the next recompute takes the tightest fit.  Safe without synchronisation: every kernel that touched a released
// arena ran before the recompute that replaced its last mesh, whose results the host has already waited for.
void free_arena(chisel_hip_map *m, MeshArena &A) {
    if (A.dev) {
        m->arena_pool.emplace_back(A.dev, A.capacity);
        if (m->arena_pool.size() > 32) {  // drop the smallest (take_arena_buffer accepts any buffer that is large enough)
            size_t k = 0;
            for (size_t i = 1; i < m->arena_pool.size(); i++)
                if (m->arena_pool[i].second < m->arena_pool[k].second) k = i;
            (void)hipFree(m->arena_pool[k].first);
            m->arena_pool.erase(m->arena_pool.begin() + (long)k);
        }
    }
    A = MeshArena();
}
int take_arena_buffer(chisel_hip_map *m, size_t floats, float **dev, size_t *capacity) {
    long best = -1;
    for (size_t i = 0; i < m->arena_pool.size(); i++)
        // the tightest fit; no upper bound -- a pool of buffers that are all "too large" made every recompute allocate a new one and
        // free_arena drop it again (the smallest), about 0.4 ms of hipMalloc / hipFree per recompute for the rest of the process
        if (m->arena_pool[i].second >= floats && (best < 0 || m->arena_pool[i].second < m->arena_pool[(size_t)best].second))
            best = (long)i;
    if (best >= 0) {
        *dev = m->arena_pool[(size_t)best].first;
        *capacity = m->arena_pool[(size_t)best].second;
        m->arena_pool.erase(m->arena_pool.begin() + best);
        return CHISEL_HIP_OK;
    }
    const size_t cap = floats + floats / 4 + 1024;  // headroom: consecutive recomputes are of similar size
    HIP_TRY(hipMalloc(dev, cap * sizeof(float)));
    *capacity = cap;
    return CHISEL_HIP_OK;
}
void release_mesh_ref(chisel_hip_map *m, MeshRef &ref) {
    if (ref.arena >= 0) {
        MeshArena &A = m->arenas[ref.arena];
        if (--A.live == 0) free_arena(m, A);
    }
    ref = MeshRef();
}
void clear_meshes(chisel_hip_map *m) {
    m->pending_meshes.active = false;
    m->pending_meshes.unchecked = false;
    m->deferred_set = -1;  // (whatever was queued behind an unseen recompute is void with the map; reset_map_kernel clears MC_LATCH)
    m->ghost_packed = nullptr;  // (... and so are the ghosts of a sharded recompute and what its wait-free form left open)
    m->ghost_packed_items = 0;
    m->shell_fixed_ghosts = m->shell_redrop = m->shell_uncommitted = false;
    m->shell_abort_dev = nullptr;
    for (MeshArena &A : m->arenas) free_arena(m, A);
    m->arenas.clear();
    m->meshes.clear();
}
void release_arena_pool(chisel_hip_map *m) {
    for (auto &b : m->arena_pool) (void)hipFree(b.first);
    m->arena_pool.clear();
}

// the job list kept by the integration kernels is dropped; the next recompute rebuilds it from the dirty flags (mesh_mark_kernel)
void give_up_job_list(chisel_hip_map *m) {
    if (m->mesh_buf.flags) (void)hipMemsetAsync(m->mesh_buf.flags, 0, (size_t)m->view.max_chunks * sizeof(unsigned), m->stream);
    if (m->mesh_buf.totals) (void)hipMemsetAsync(m->mesh_buf.totals + MC_KEPT, 0, sizeof(int), m->stream);
    if (m->mesh_buf.job_acc) (void)hipMemsetAsync(m->mesh_buf.job_acc, 0, (size_t)m->mesh_buf.capacity * sizeof(unsigned long long), m->stream);  // (a count kernel that never ran to its end)
    m->mesh_mark_needed = true;
    m->removed_since_recompute = 0;
}

void launch_mesh_count(chisel_hip_map *m) {
    MeshBuffers &B = m->mesh_buf;
    int *d_totals = mesh_totals(m);
    int *n_jobs = B.ext_ids ? B.ext_n : (B.n_jobs ? B.n_jobs : d_totals + MT_JOBS);  // a plan's job count, the kept job list's counter, or the count a caller put into the totals
    const int *ids = B.ext_ids ? B.ext_ids : (B.n_jobs ? m->view.mesh_jobs : B.ids);   // ... and the entries that go with it
    const int ids_capacity = B.ext_ids ? B.ext_capacity : (B.n_jobs ? m->view.mesh_jobs_capacity : B.capacity);
    ProfScope ps(m, CHISEL_HIP_KERNEL_MESH);
    // one single-wave workgroup per (job, sub-job): the number of jobs is only known on the device, so the grid is sized from what the
    // previous recompute had (+ 1/4) -- a shortfall is made up by the workgroups taking a second unit, surplus ones leave at once
    const long long units = (long long)(m->mesh_jobs_hint > 0 ? m->mesh_jobs_hint + m->mesh_jobs_hint / 4 + 8 : 1024) * mesh_subjobs(m);
    const dim3 grid((unsigned)(std::min<long long>(std::max<long long>(units, 2048), 1 << 17) + 7) / 8 * 8);
    const int part = B.tri_capacity / MESH_PARTS, keep = m->mesh_detached ? 1 : 0;
    const int done_seq = (int)m->launch_seq;  // (every integration launched so far sits in front of this kernel on the map's stream)
    switch (m->N) {
        case 8: hipLaunchKernelGGL(mesh_count_kernel_8, grid, dim3(64), 0, m->stream, m->view, ids, ids_capacity, B.jobs, n_jobs, B.info, d_totals, B.cnt, B.job_acc, B.tris, B.corners, part, B.flags, keep, done_seq); break;
        case 16: hipLaunchKernelGGL(mesh_count_kernel_16, grid, dim3(64), 0, m->stream, m->view, ids, ids_capacity, B.jobs, n_jobs, B.info, d_totals, B.cnt, B.job_acc, B.tris, B.corners, part, B.flags, keep, done_seq); break;
        case 32: hipLaunchKernelGGL(mesh_count_kernel_32, grid, dim3(64), 0, m->stream, m->view, ids, ids_capacity, B.jobs, n_jobs, B.info, d_totals, B.cnt, B.job_acc, B.tris, B.corners, part, B.flags, keep, done_seq); break;
    }
}

// meshes of the chunks whose ids sit in mesh_buf.ids (device; their number too).  Everything is queued at once -- job
// table, count kernel, triangle kernel into an arena sized from the previous recompute, dirty-flag reset -- and the host
// then finds the totals in pinned memory, written by the triangle kernel's first thread: the device never waits for
// the host and nothing but kernels sits on the map's stream.  Only a batch
// that outgrew the triangle list or the arena is emitted again after a full wait (the map has not changed meanwhile:
// nothing else was queued).
int recompute_meshes(chisel_hip_map *m) {
    RoctxRange range("chisel_hip mesh recompute: count, triangles");
    MeshBuffers &B = m->mesh_buf;
    const MeshParams P = mesh_params(m);
    if (!B.tris) {
        B.tri_capacity = std::max(B.tri_capacity, m->mesh_tiny ? 4 * MESH_PARTS : 1 << 20);
        HIP_TRY(hipMalloc(&B.tris, (size_t)B.tri_capacity * sizeof(TriRec)));
        HIP_TRY(hipMalloc(&B.corners, (size_t)B.tri_capacity * sizeof(CubeCorners)));
    }
    // an arena record and a buffer that should do: twice what the previous recompute needed
    int arena_id = -1;
    for (size_t i = 0; i < m->arenas.size() && arena_id < 0; i++)
        if (!m->arenas[i].dev) arena_id = (int)i;
    if (arena_id < 0) {
        m->arenas.emplace_back();
        arena_id = (int)m->arenas.size() - 1;
    }
    {
        MeshArena &A = m->arenas[arena_id];
        A = MeshArena();
        int rc_a = take_arena_buffer(m, m->mesh_tiny ? (size_t)4096 : std::max<size_t>(2 * m->mesh_need_hint, (size_t)1 << 22), &A.dev, &A.capacity);
        if (rc_a) return rc_a;
    }
    launch_mesh_count(m);
    if (!m->mesh_detached) m->dirty_epoch++;  // (the count kernel empties the list of dirty slots: meshesToUpdate.clear(), Chisel.cpp:57)
    // the totals go straight into pinned memory from the first thread of the triangle kernel; they are looked at when the caller
    // next touches the map (check_mesh_totals polls the sequence number): until then the host is free to queue the next batch's
    // front half, and by then the triangle kernel is usually still running, so the next integration queues up behind it without a gap
    m->mesh_seq++;
    m->recomputes++;
    {
        ProfScope ps(m, CHISEL_HIP_KERNEL_MESH);
        launch_mesh_triangles(m, P, m->arenas[arena_id].dev, m->arenas[arena_id].capacity, 1 | (B.n_jobs ? 2 : 0));
    }
    HIP_TRY(hipGetLastError());
    m->pending_meshes.unchecked = true;
    m->pending_meshes.active = false;
    m->pending_meshes.arena = arena_id;
    // a buffer for the NEXT recompute, allocated now that the device is busy (arenas stay referenced for long, the pool is
    // usually empty, and an allocation in front of the next recompute would sit on its critical path)
    {
        const size_t want = std::max<size_t>(2 * m->mesh_need_hint, (size_t)1 << 22);
        bool have = false;
        for (const auto &b : m->arena_pool) have = have || b.second >= want;
        if (!have) {
            float *spare = nullptr;
            const size_t cap = want + want / 4 + 1024;
            if (hipMalloc(&spare, cap * sizeof(float)) == hipSuccess) m->arena_pool.emplace_back(spare, cap);
        }
    }
    return CHISEL_HIP_OK;
}

// The totals of the recompute in flight: sizes the arena's contents, emits again when the batch outgrew the triangle
// list or the arena.  Must run before anything else changes the map (a second emission reads the voxels): every entry
// point that queues map-changing work calls it first; it waits for the count kernel only, not for the stream.
bool mesh_totals_published(const chisel_hip_map *m) { return ((volatile const int *)m->mesh_totals_host)[5] == m->mesh_seq; }
int replay_deferred_set(chisel_hip_map *m, int set);  // chisel_hip.hip
void launch_fixed_drop(chisel_hip_map *m, const int *latch);  // chisel_hip.hip
int check_mesh_totals(chisel_hip_map *m) {
    if (!m->pending_meshes.unchecked) return CHISEL_HIP_OK;
    m->pending_meshes.unchecked = false;
    // a launch set queued behind this recompute before its totals were seen (launch_back): if the recompute is emitted again below, that
    // set's integration kernel has left the map alone (MC_LATCH) and is launched again afterwards
    const int deferred = m->deferred_set;
    m->deferred_set = -1;
    MeshBuffers &B = m->mesh_buf;
    int *d_totals = mesh_totals(m);
    const bool color = m->cfg.use_color != 0;
    const MeshParams P = mesh_params(m);
    int arena_id = m->pending_meshes.arena;
    {
        // the device writes totals and sequence number as one 16-byte store (mesh_triangle_kernel): word 3 is the sequence number
        volatile int *host = m->mesh_totals_host;
        const auto t0 = std::chrono::steady_clock::now();
        while (host[5] != m->mesh_seq) {  // (the copy of the sequence number written behind a system-scope fence: words 0-3 are complete)
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
                HIP_TRY(hipStreamSynchronize(m->stream));  // long queue in front of the recompute, or a failed launch: no more polling
                if (host[5] != m->mesh_seq || host[3] != m->mesh_seq) return fail(CHISEL_HIP_ERR_HIP, "mesh totals were not published");
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    const bool device_unfit = m->mesh_totals_host[4] != 0;  // the triangle kernel's own verdict (what MC_LATCH was set by)
    // From here on MC_LATCH is known to be set (device_unfit) and the deferred set's kernel to have left the map alone: whatever way this
    // function ends -- also through one of the error returns below -- the latch is cleared and that integration launched again; a frame
    // must not be lost, and no later launch find the latch still up, because a mesh buffer could not be grown.
    struct LatchGuard {
        chisel_hip_map *m;
        int deferred;
        bool armed, settled = false;
        ~LatchGuard() {
            if (!armed || settled) return;
            (void)hipStreamSynchronize(m->stream);
            (void)hipMemsetAsync(mesh_totals(m) + MC_LATCH, 0, sizeof(int), m->stream);
            if (deferred >= 0) (void)replay_deferred_set(m, deferred);
        }
    } latch_guard{m, deferred, device_unfit};
    const unsigned packed_jobs = (unsigned)m->mesh_totals_host[2];
    int totals[4] = {m->mesh_totals_host[0], m->mesh_totals_host[1], (int)(packed_jobs >> 31), (int)(packed_jobs & 0x7fffffffu)};
    // A chunk of an earlier batch could not be allocated (word [0] of the map's error flags: pool / hash; cloud reports live in
    // word [1] and are none of this function's business): the map is incomplete.  The recompute is finished all the same -- its
    // mark / collect kernels have already consumed the dirty flags, abandoning it would leave those chunks without a mesh for
    // good -- and the failure is reported afterwards.
    // (read from the flag itself: every kernel that could have raised it finished before the count kernel started)
    const int pool_error = reinterpret_cast<volatile int *>(m->error_flag_host)[0];
    bool redo = false;
    if (totals[MT_OVERFLOW]) {
        // a partition of the record lists was too small: grow them to (at least) twice what this batch needs and list again, until every
        // partition holds its share (dirty flags are not read by the count kernel; the map has not changed: nothing else was queued)
        const int n_again = totals[MT_JOBS];
        for (int attempt = 0;; attempt++) {
            HIP_TRY(hipStreamSynchronize(m->stream));
            HIP_TRY(hipFree(B.tris));
            HIP_TRY(hipFree(B.corners));
            B.tris = nullptr;
            B.corners = nullptr;
            const long long want = std::max<long long>(2ll * B.tri_capacity, 2ll * std::max(totals[MT_TRIS], totals[MT_GRIDS]) + 16 * MESH_PARTS);
            if (want > (1ll << 30)) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "mesh record lists beyond 2^30 entries");
            B.tri_capacity = (int)((want + MESH_PARTS - 1) / MESH_PARTS * MESH_PARTS);
            HIP_TRY(hipMalloc(&B.tris, (size_t)B.tri_capacity * sizeof(TriRec)));
            HIP_TRY(hipMalloc(&B.corners, (size_t)B.tri_capacity * sizeof(CubeCorners)));
            HIP_TRY(zero_mesh_counters(m));
            // (the kept job list's counter has been emptied by the first emission; its entries are untouched -- nothing has integrated since --
            // and their number is in the totals)
            if (B.n_jobs) {
                m->mesh_totals_host[8] = n_again;  // (page-locked: an asynchronous copy must not read a local of this function)
                HIP_TRY(hipMemcpyAsync(B.n_jobs, &m->mesh_totals_host[8], sizeof(int), hipMemcpyHostToDevice, m->stream));
            }
            launch_mesh_count(m);
            if (B.n_jobs) HIP_TRY(hipMemsetAsync(B.n_jobs, 0, sizeof(int), m->stream));
            HIP_TRY(hipMemcpyAsync(totals, d_totals, 4 * sizeof(int), hipMemcpyDeviceToHost, m->stream));
            HIP_TRY(hipStreamSynchronize(m->stream));
            if (!totals[MT_OVERFLOW]) break;
            if (attempt == 12) return fail(CHISEL_HIP_ERR_HIP, "mesh record lists overflow after growing them");
        }
        redo = true;
        HIP_TRY(hipMemcpy(m->mesh_info_host, B.info, (size_t)std::min(MESH_INFO_PREFETCH, B.capacity) * sizeof(JobInfo), hipMemcpyDeviceToHost));
    }
    const int n = totals[MT_JOBS];
    if ((size_t)totals[MT_TRIS] > 0x7fffffffull / 9) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "more than 2^31 / 9 mesh triangles in one recompute");
    const size_t nv = (size_t)totals[MT_TRIS] * 3, ng = (size_t)totals[MT_GRIDS];
    {
        MeshArena &A = m->arenas[arena_id];
        A.nv = nv;
        A.ng = ng;
        A.color = color;
        if (redo ? !device_unfit : (A.floats() > A.capacity) != device_unfit)  // (the device's verdict is what a deferred launch went by)
            return fail(CHISEL_HIP_ERR_HIP, "mesh recompute: host and device disagree on whether it fitted");
        if (redo || A.floats() > A.capacity) {
            // did not fit (or was listed again): emit once more into a buffer of the right size
            HIP_TRY(hipStreamSynchronize(m->stream));
            HIP_TRY(hipMemsetAsync(mesh_totals(m) + MC_LATCH, 0, sizeof(int), m->stream));  // (in front of the replay below)
            if (A.floats() > A.capacity) {
                m->arena_pool.emplace_back(A.dev, A.capacity);
                A.dev = nullptr;
                int rc_a = take_arena_buffer(m, A.floats(), &A.dev, &A.capacity);
                if (rc_a) return rc_a;
            }
            {
                ProfScope ps(m, CHISEL_HIP_KERNEL_MESH);
                launch_mesh_triangles(m, P, A.dev, A.capacity);
            }
            HIP_TRY(hipGetLastError());
            latch_guard.settled = true;
            if (m->shell_redrop && m->ghost_packed)  // the ghosts of a wait-free sharded recompute: their drop kernel left them for this emission
                launch_fixed_drop(m, nullptr);
            if (deferred >= 0) {
                int rc_r = replay_deferred_set(m, deferred);
                if (rc_r) return rc_r;
            }
        }
        m->mesh_need_hint = A.floats();
        if (nv + ng == 0) {
            free_arena(m, A);
            arena_id = -1;
        }
    }
    if (m->shell_redrop) {
        m->shell_redrop = false;
        m->ghost_packed = nullptr;
    }
    m->mesh_jobs_hint = n;
    if (n != 0) {
        // The per-chunk results (sizes, positions in the arena, ids) stay on the device for now: the bookkeeping follows when
        // a mesh is next asked for or recomputed (resolve_pending_meshes).
        m->pending_meshes.active = true;
        m->pending_meshes.n = n;
        m->pending_meshes.arena = arena_id;
    }
    if (pool_error != 0)
        return fail(CHISEL_HIP_ERR_POOL_FULL, pool_error == 1 ? "chunk pool exhausted: raise chisel_hip_config.max_chunks"
                                                              : "chunk hash table exhausted: raise chisel_hip_config.max_chunks");
    return CHISEL_HIP_OK;
}

// Second half of a recompute: ChunkManager::allMeshes on the host side (which chunk's mesh is where).  The device
// buffers read here were complete when recompute_meshes returned (it waited for the count kernel), so the copies use
// their own stream and do not wait for batches queued since.
int resolve_pending_meshes(chisel_hip_map *m) {
    int rc_c = check_mesh_totals(m);
    if (rc_c) return rc_c;
    if (!m->pending_meshes.active) return CHISEL_HIP_OK;
    m->pending_meshes.active = false;
    MeshBuffers &B = m->mesh_buf;
    const int n = m->pending_meshes.n, arena_id = m->pending_meshes.arena;
    // the records of the first MESH_INFO_PREFETCH jobs are on the host (mesh_triangle_kernel wrote them and then the sequence number)
    {
        volatile int *host = m->mesh_totals_host;
        const auto t0 = std::chrono::steady_clock::now();
        while (host[6] != m->mesh_seq) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
                HIP_TRY(hipStreamSynchronize(m->stream));
                if (host[6] != m->mesh_seq) return fail(CHISEL_HIP_ERR_HIP, "mesh job records were not published");
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    g_mesh_timer.lap(1);
    std::vector<JobInfo> tail;
    if (n > MESH_INFO_PREFETCH) {
        tail.resize((size_t)n - MESH_INFO_PREFETCH);
        HIP_TRY(hipMemcpyAsync(tail.data(), B.info + MESH_INFO_PREFETCH, tail.size() * sizeof(JobInfo), hipMemcpyDeviceToHost, m->copy_stream));
        HIP_TRY(hipStreamSynchronize(m->copy_stream));
    }
    int with_tris = 0;
    for (int j = 0; j < n; j++) {
        const JobInfo &ji = j < MESH_INFO_PREFETCH ? m->mesh_info_host[j] : tail[(size_t)j - MESH_INFO_PREFETCH];
        if (!ji.present) continue;  // RecomputeMesh: "if (!HasChunk(chunkID)) return" (ChunkManager.cpp:93-96)
        const uint64_t key = pack_id(ji.x, ji.y, ji.z);
        const size_t cv = (size_t)ji.n_vertices, cg = (size_t)ji.n_grids;
        with_tris += cv != 0;
        auto it = m->meshes.find(key);
        // the reference regenerates an existing Mesh object in place (it may become empty) and inserts a new one
        // only when it has grids (ChunkManager.cpp:101-127)
        if (it == m->meshes.end()) {
            if (cg == 0) continue;
            it = m->meshes.emplace(key, MeshRef()).first;
        }
        MeshRef &ref = it->second;
        release_mesh_ref(m, ref);
        if (cv + cg) {
            ref.arena = arena_id;
            ref.v_off = 3 * (size_t)ji.tri_base;  // first vertex of the job in the arena
            ref.n_v = cv;
            ref.g_off = (size_t)ji.grid_base;     // first grid entry
            ref.n_g = cg;
            m->arenas[arena_id].live++;
        }
    }
    if (g_host_timer.on) fprintf(stderr, "chisel_hip mesh recompute: %d jobs, %d with triangles\n", n, with_tris);
    if (arena_id >= 0 && m->arenas[arena_id].live == 0) free_arena(m, m->arenas[arena_id]);
    g_mesh_timer.lap(2);
    return CHISEL_HIP_OK;
}

// host copy of an arena, made on first use
int arena_on_host(chisel_hip_map *m, MeshArena &A) {
    if (A.host_valid) return CHISEL_HIP_OK;
    A.host.resize(A.floats());
    if (!A.host.empty()) {
        HIP_TRY(hipMemcpyAsync(A.host.data(), A.dev, A.floats() * sizeof(float), hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipStreamSynchronize(m->stream));
    }
    A.host_valid = true;
    return CHISEL_HIP_OK;
}

// pointers to the four arrays of one mesh in its arena's host copy (nullptr for an empty mesh)
struct MeshView {
    const float *v = nullptr, *n = nullptr, *c = nullptr, *g = nullptr;
    size_t n_v = 0, n_g = 0;
};
int view_mesh(chisel_hip_map *m, const MeshRef &ref, MeshView &out) {
    out = MeshView();
    if (ref.arena < 0) return CHISEL_HIP_OK;
    MeshArena &A = m->arenas[ref.arena];
    int rc = arena_on_host(m, A);
    if (rc) return rc;
    const float *base = A.host.data();
    out.v = base + 3 * ref.v_off;
    out.n = base + 3 * A.nv + 3 * ref.v_off;
    out.c = A.color ? base + 6 * A.nv + 3 * ref.v_off : nullptr;
    out.g = base + 3 * A.nv * (A.color ? 3 : 2) + 3 * ref.g_off;
    out.n_v = ref.n_v;
    out.n_g = ref.n_g;
    return CHISEL_HIP_OK;
}

int query_sdf(chisel_hip_map *m, const float pos[3], int with_gradient, double *dist, float *grad, int *found) {
    HIP_TRY(hipSetDevice(m->device));
    if (!m->mesh_buf.query) HIP_TRY(hipMalloc(&m->mesh_buf.query, 8 * sizeof(double)));
    const MeshParams P = mesh_params(m);
    switch (m->N) {
        case 8: hipLaunchKernelGGL(query_sdf_kernel<8>, dim3(1), dim3(1), 0, m->stream, m->view, P, pos[0], pos[1], pos[2], with_gradient, m->mesh_buf.query); break;
        case 16: hipLaunchKernelGGL(query_sdf_kernel<16>, dim3(1), dim3(1), 0, m->stream, m->view, P, pos[0], pos[1], pos[2], with_gradient, m->mesh_buf.query); break;
        case 32: hipLaunchKernelGGL(query_sdf_kernel<32>, dim3(1), dim3(1), 0, m->stream, m->view, P, pos[0], pos[1], pos[2], with_gradient, m->mesh_buf.query); break;
    }
    double out[5];
    HIP_TRY(hipMemcpyAsync(out, m->mesh_buf.query, sizeof(out), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (dist) *dist = out[0];
    if (grad) {
        grad[0] = (float)out[1];
        grad[1] = (float)out[2];
        grad[2] = (float)out[3];
    }
    if (found) *found = out[4] != 0.0 ? 1 : 0;
    return CHISEL_HIP_OK;
}

// ids of all meshes, sorted (x, then y, then z) so that every listing / export is deterministic
std::vector<uint64_t> sorted_mesh_keys(const chisel_hip_map *m) {
    std::vector<uint64_t> keys;
    keys.reserve(m->meshes.size());
    for (const auto &kv : m->meshes) keys.push_back(kv.first);
    std::sort(keys.begin(), keys.end(), [](uint64_t a, uint64_t b) {
        int ax, ay, az, bx, by, bz;
        unpack_id(a, ax, ay, az);
        unpack_id(b, bx, by, bz);
        if (ax != bx) return ax < bx;
        if (ay != by) return ay < by;
        return az < bz;
    });
    return keys;
}

}  // namespace

extern "C" {

int chisel_hip_update_meshes(chisel_hip_map *m, int force) {
    SETTLE(m);
    if (m && m->is_group) return group::update_meshes(m, force);
    if (!m) return fail(CHISEL_HIP_ERR_INVALID, "null map");
    if (m->cfg.n_shards > 1)
        return fail(CHISEL_HIP_ERR_UNSUPPORTED,
                    "a shard cannot mesh on its own (its chunks' neighbours live on other shards): exchange them with chisel_hip_export_chunks / "
                    "import_ghost_chunks and call chisel_hip_update_meshes_of (cvids_amd/sharded.py: ShardedChisel.UpdateMeshes)");
    HIP_TRY(hipSetDevice(m->device));
    // Chisel.cpp:53-58: "static int cnt = 0; if (cnt++ % 10 == 0)" -- the recompute runs on every 10th call
    if (!force && (m->update_meshes_calls++ % 10) != 0) return CHISEL_HIP_OK;
    int rc = CHISEL_HIP_OK;
    std::vector<int> extra;
    extra.reserve(m->pending_mesh_ids.size() * 3);
    for (uint64_t key : m->pending_mesh_ids) {
        int x, y, z;
        unpack_id(key, x, y, z);
        extra.push_back(x); extra.push_back(y); extra.push_back(z);
    }
    g_mesh_timer.start();
    rc = check_mesh_totals(m);
    g_mesh_timer.lap(0);
    if (!rc) rc = resolve_pending_meshes(m);  // the device buffers of the previous recompute are about to be reused
    if (rc) return rc;
    rc = collect_mesh_ids(m, extra);
    if (!rc) rc = recompute_meshes(m);  // ends with meshesToUpdate.clear() (Chisel.cpp:57)
    g_mesh_timer.lap(3);
    g_mesh_timer.calls++;
    if (rc) {
        // the mark kernel may have flagged slots that no count kernel will now reset: a slot whose flag stays set could never
        // become a job again
        give_up_job_list(m);
        return rc;
    }
    m->pending_mesh_ids.clear();
    m->removed_since_recompute = 0;
    return CHISEL_HIP_OK;
}

int chisel_hip_update_meshes_of(chisel_hip_map *m, const int *ids, int n) {
    SETTLE(m);
    if (m && m->is_group) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "chisel_hip_update_meshes_of is a call between the shards of a map: a group makes it itself (chisel_hip_update_meshes)");
    if (!m || n < 0 || (n > 0 && !ids)) return fail(CHISEL_HIP_ERR_INVALID, "bad id list");
    HIP_TRY(hipSetDevice(m->device));
    int rc = resolve_pending_meshes(m);  // the device buffers of the previous recompute are about to be reused
    if (rc) return rc;
    for (int j = 0; j < n; j++)
        if (chunk_owner(ids[3 * j], ids[3 * j + 1], ids[3 * j + 2], m->cfg.n_shards, m->cfg.shard_block) != m->cfg.shard_rank)
            return fail(CHISEL_HIP_ERR_INVALID, "a shard meshes its own chunks only");
    MeshBuffers &B = m->mesh_buf;
    rc = ensure_mesh_jobs(m, std::max(n, m->view.max_chunks));
    if (rc) return rc;
    const int totals[4] = {0, 0, 0, n};
    B.n_jobs = nullptr;  // the count is the one written here
    B.ext_ids = nullptr;
    HIP_TRY(hipMemcpyAsync(mesh_totals(m), totals, sizeof(totals), hipMemcpyHostToDevice, m->stream));
    HIP_TRY(hipMemsetAsync(mesh_totals(m) + MC_CURSORS, 0, 2 * MESH_PARTS * sizeof(int), m->stream));
    m->mesh_jobs_hint = n;
    if (n) HIP_TRY(hipMemcpyAsync(B.ids, ids, (size_t)n * 3 * sizeof(int), hipMemcpyHostToDevice, m->stream));
    if (!m->mesh_detached) hipLaunchKernelGGL(clear_dirty_kernel, dim3((m->view.max_chunks + 255) / 256), dim3(256), 0, m->stream, m->view, (const int *)nullptr);
    HIP_TRY(hipGetLastError());
    rc = recompute_meshes(m);
    if (rc) return rc;
    if (!m->mesh_detached) m->pending_mesh_ids.clear();
    return CHISEL_HIP_OK;
}

// Step 5 of a sharded recompute: the jobs of the latest plan (chisel_hip_shell_plan_device) -- a list that never left the device.
int chisel_hip_update_meshes_planned(chisel_hip_map *m) {
    SETTLE(m);
    if (m && m->is_group) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "a call between the shards of a map: a group makes it itself (chisel_hip_update_meshes)");
    if (!m || !m->shell_plan.my_jobs) return fail(CHISEL_HIP_ERR_INVALID, "no plan: chisel_hip_shell_plan_device first");
    HIP_TRY(hipSetDevice(m->device));
    int rc = resolve_pending_meshes(m);  // the device buffers of the previous recompute are about to be reused
    if (rc) return rc;
    MeshBuffers &B = m->mesh_buf;
    rc = ensure_mesh_jobs(m, m->view.max_chunks);
    if (rc) return rc;
    B.n_jobs = nullptr;
    B.ext_ids = m->shell_plan.my_jobs;
    B.ext_n = m->shell_plan.ctl;
    B.ext_capacity = m->shell_plan.max_jobs;
    HIP_TRY(zero_mesh_counters(m));
    m->mesh_totals_clean = false;
    m->mesh_jobs_hint = m->shell_jobs;
    // (the wait-free form: a recompute that was called off on the device keeps its dirty flags, and the host-held ids until chisel_hip_shell_commit)
    hipLaunchKernelGGL(clear_dirty_kernel, dim3((m->view.max_chunks + 255) / 256), dim3(256), 0, m->stream, m->view, m->shell_uncommitted ? m->shell_abort_dev : (const int *)nullptr);
    HIP_TRY(hipGetLastError());
    rc = recompute_meshes(m);
    if (rc) return rc;
    if (m->shell_uncommitted && m->shell_abort_dev) {
        hipLaunchKernelGGL(shell_abort_relist_kernel, dim3(1), dim3(64), 0, m->stream, m->view, m->shell_abort_dev);
        HIP_TRY(hipGetLastError());
    }
    if (!m->shell_uncommitted) m->pending_mesh_ids.clear();
    return CHISEL_HIP_OK;
}

// ChunkManager::RecomputeMesh(chunkID, mutex) (ChunkManager.cpp:91-128): the mesh of one chunk into allMeshes; meshesToUpdate is the
// caller's business there (Chisel::UpdateMeshes clears it, Chisel.cpp:57), so nothing of it is touched here
int chisel_hip_recompute_mesh(chisel_hip_map *m, const int id[3]) {
    SETTLE(m);
    if (m && m->is_group) return id ? chisel_hip_recompute_mesh(group::owner_map(m, id), id) : fail(CHISEL_HIP_ERR_INVALID, "null argument");
    if (!m || !id) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    m->mesh_detached = true;
    const int rc = chisel_hip_update_meshes_of(m, id, 1);
    m->mesh_detached = false;
    return rc;
}

// ChunkManager::ExtractInsideVoxelMesh / ExtractBorderVoxelMesh (ChunkManager.cpp:259-379): the triangles of ONE cube of a resident
// chunk (mesh_one_cube_kernel): up to 15 vertices and as many (face) normals; *occupied = a grid entry belongs to the cube.
int chisel_hip_mesh_cube(chisel_hip_map *m, const int id[3], const int voxel[3], const float coords[3], float *vertices, float *normals, int *n_vertices,
                         int *occupied) {
    SETTLE(m);
    if (m && m->is_group) return id ? chisel_hip_mesh_cube(group::owner_map(m, id), id, voxel, coords, vertices, normals, n_vertices, occupied)
                                    : fail(CHISEL_HIP_ERR_INVALID, "null argument");
    if (!m || !id || !voxel || !coords || !n_vertices || !occupied) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    if (m->cfg.n_shards > 1) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "a shard does not hold the neighbours of its chunks");
    for (int a = 0; a < 3; a++)
        if (voxel[a] < -1 || voxel[a] >= m->N) return fail(CHISEL_HIP_ERR_INVALID, "voxel index outside the chunk");
    HIP_TRY(hipSetDevice(m->device));
    int rc = check_mesh_totals(m);
    if (rc) return rc;
    if (!m->mesh_buf.query) HIP_TRY(hipMalloc(&m->mesh_buf.query, 8 * sizeof(double)));
    if (!m->mesh_buf.cube) HIP_TRY(hipMalloc(&m->mesh_buf.cube, 92 * sizeof(float)));
    const MeshParams P = mesh_params(m);
    float *out = m->mesh_buf.cube;
    switch (m->N) {
        case 8: hipLaunchKernelGGL(mesh_one_cube_kernel<8>, dim3(1), dim3(1), 0, m->stream, m->view, P, id[0], id[1], id[2], voxel[0], voxel[1], voxel[2], coords[0], coords[1], coords[2], out); break;
        case 16: hipLaunchKernelGGL(mesh_one_cube_kernel<16>, dim3(1), dim3(1), 0, m->stream, m->view, P, id[0], id[1], id[2], voxel[0], voxel[1], voxel[2], coords[0], coords[1], coords[2], out); break;
        case 32: hipLaunchKernelGGL(mesh_one_cube_kernel<32>, dim3(1), dim3(1), 0, m->stream, m->view, P, id[0], id[1], id[2], voxel[0], voxel[1], voxel[2], coords[0], coords[1], coords[2], out); break;
    }
    float host[92];
    HIP_TRY(hipMemcpyAsync(host, out, sizeof(host), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    const int nv = (int)host[0];
    *n_vertices = nv;
    *occupied = host[1] != 0.0f ? 1 : 0;
    if (vertices) memcpy(vertices, host + 2, (size_t)nv * 3 * sizeof(float));
    if (normals) memcpy(normals, host + 2 + 45, (size_t)nv * 3 * sizeof(float));
    return CHISEL_HIP_OK;
}

// ChunkManager::GenerateMesh(chunk, mesh) (ChunkManager.cpp:381-447) -- marching cubes of ONE chunk into the caller's arrays, face
// normals, nothing else -- or, with stages, the whole of RecomputeMesh (bit 0: + ComputeNormalsFromGradients, bit 1: + ColorizeMesh)
// without touching ChunkManager::allMeshes or meshesToUpdate.  The ordinary recompute runs with the dirty-flag housekeeping switched
// off, its result is copied out and the map's own mesh entry of the chunk (if any) is put back.
int chisel_hip_generate_mesh(chisel_hip_map *m, const int id[3], int stages, int64_t capacity_vertices, int64_t capacity_grids, float *vertices,
                             float *normals, float *colors, float *grids, int64_t *n_vertices, int64_t *n_grids) {
    SETTLE(m);
    if (m && m->is_group) return id ? chisel_hip_generate_mesh(group::owner_map(m, id), id, stages, capacity_vertices, capacity_grids, vertices, normals, colors, grids, n_vertices, n_grids)
                                    : fail(CHISEL_HIP_ERR_INVALID, "null argument");
    if (!m || !id || !n_vertices || !n_grids) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(m->device));
    int rc = resolve_pending_meshes(m);
    if (rc) return rc;
    const uint64_t key = pack_id(id[0], id[1], id[2]);
    MeshRef saved;
    bool had = false;
    {
        auto it = m->meshes.find(key);
        if (it != m->meshes.end()) {
            had = true;
            saved = it->second;
            if (saved.arena >= 0) m->arenas[saved.arena].live++;  // keeps the arena while the entry is replaced below
        }
    }
    m->mesh_stages = stages & 3;
    m->mesh_detached = true;
    rc = chisel_hip_update_meshes_of(m, id, 1);
    if (!rc) rc = resolve_pending_meshes(m);
    m->mesh_stages = 3;
    m->mesh_detached = false;
    *n_vertices = *n_grids = 0;
    auto it = m->meshes.find(key);
    if (!rc && it != m->meshes.end() && !(had && it->second.arena == saved.arena && it->second.v_off == saved.v_off && saved.arena >= 0)) {
        MeshView v;
        rc = view_mesh(m, it->second, v);
        if (!rc) {
            *n_vertices = (int64_t)v.n_v;
            *n_grids = (int64_t)v.n_g;
            if ((int64_t)v.n_v <= capacity_vertices && (int64_t)v.n_g <= capacity_grids) {
                if (vertices && v.n_v) memcpy(vertices, v.v, v.n_v * 3 * sizeof(float));
                if (normals && v.n_v) memcpy(normals, v.n, v.n_v * 3 * sizeof(float));
                if (colors && v.n_v && v.c) memcpy(colors, v.c, v.n_v * 3 * sizeof(float));
                if (grids && v.n_g) memcpy(grids, v.g, v.n_g * 3 * sizeof(float));
            }
        }
    }
    // put the map's own entry back
    it = m->meshes.find(key);
    if (it != m->meshes.end()) {
        const bool replaced = !(had && it->second.arena == saved.arena && it->second.v_off == saved.v_off && it->second.n_v == saved.n_v);
        if (replaced || !had) {
            release_mesh_ref(m, it->second);
            if (had) it->second = saved;  // (its arena's count still carries the +1 from above: that is this reference again)
            else m->meshes.erase(it);
        } else if (saved.arena >= 0) {
            m->arenas[saved.arena].live--;  // nothing was replaced (the chunk is gone): drop the extra hold
        }
    } else if (had && saved.arena >= 0) {
        m->arenas[saved.arena].live--;
    }
    return rc;
}

int chisel_hip_num_meshes(chisel_hip_map *m, int64_t *out) {
    SETTLE(m);
    if (m && m->is_group) {
        if (!out) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
        *out = 0;
        for (chisel_hip_map *s : m->shards) {
            int64_t n = 0;
            const int rc = chisel_hip_num_meshes(s, &n);
            if (rc) return rc;
            *out += n;
        }
        return CHISEL_HIP_OK;
    }
    if (!m || !out) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(m->device));
    int rc_p = resolve_pending_meshes(m);
    if (rc_p) return rc_p;
    *out = (int64_t)m->meshes.size();
    return CHISEL_HIP_OK;
}

int chisel_hip_list_meshes(chisel_hip_map *m, int *ids, int64_t max_ids, int64_t *count) {
    SETTLE(m);
    if (m && m->is_group) return group::list_meshes(m, ids, max_ids, count);
    if (!m || !count) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(m->device));
    int rc_p = resolve_pending_meshes(m);
    if (rc_p) return rc_p;
    const std::vector<uint64_t> keys = sorted_mesh_keys(m);
    *count = (int64_t)keys.size();
    if (ids)
        for (int64_t k = 0; k < std::min<int64_t>(*count, max_ids); k++) unpack_id(keys[k], ids[3 * k], ids[3 * k + 1], ids[3 * k + 2]);
    return CHISEL_HIP_OK;
}

int chisel_hip_mesh_size(chisel_hip_map *m, const int id[3], int64_t *nv, int64_t *ng) {
    SETTLE(m);
    if (m && m->is_group) return id ? chisel_hip_mesh_size(group::owner_map(m, id), id, nv, ng) : fail(CHISEL_HIP_ERR_INVALID, "null argument");
    if (!m || !id) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(m->device));
    int rc_p = resolve_pending_meshes(m);
    if (rc_p) return rc_p;
    auto it = m->meshes.find(pack_id(id[0], id[1], id[2]));
    if (it == m->meshes.end()) return fail(CHISEL_HIP_ERR_NOT_FOUND, "no mesh for this chunk (ChunkManager::GetMesh would throw std::out_of_range)");
    if (nv) *nv = (int64_t)it->second.n_v;
    if (ng) *ng = (int64_t)it->second.n_g;
    return CHISEL_HIP_OK;
}

int chisel_hip_download_mesh(chisel_hip_map *m, const int id[3], float *v, float *n, float *c, float *g) {
    SETTLE(m);
    if (m && m->is_group) return id ? chisel_hip_download_mesh(group::owner_map(m, id), id, v, n, c, g) : fail(CHISEL_HIP_ERR_INVALID, "null argument");
    if (!m || !id) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(m->device));
    int rc_p = resolve_pending_meshes(m);
    if (rc_p) return rc_p;
    auto it = m->meshes.find(pack_id(id[0], id[1], id[2]));
    if (it == m->meshes.end()) return fail(CHISEL_HIP_ERR_NOT_FOUND, "no mesh for this chunk (ChunkManager::GetMesh would throw std::out_of_range)");
    HIP_TRY(hipSetDevice(m->device));
    MeshView mv;
    int rc = view_mesh(m, it->second, mv);
    if (rc) return rc;
    if (v && mv.n_v) memcpy(v, mv.v, mv.n_v * 3 * sizeof(float));
    if (n && mv.n_v) memcpy(n, mv.n, mv.n_v * 3 * sizeof(float));
    if (c && mv.n_v && mv.c) memcpy(c, mv.c, mv.n_v * 3 * sizeof(float));
    if (g && mv.n_g) memcpy(g, mv.g, mv.n_g * 3 * sizeof(float));
    return CHISEL_HIP_OK;
}

int chisel_hip_get_sdf(chisel_hip_map *m, const float pos[3], double *dist, int *found) {
    SETTLE(m);
    if (m && m->is_group) return pos ? group::get_sdf(m, pos, dist, found) : fail(CHISEL_HIP_ERR_INVALID, "null argument");
    if (!m || !pos) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    return query_sdf(m, pos, 0, dist, nullptr, found);
}

int chisel_hip_get_sdf_and_gradient(chisel_hip_map *m, const float pos[3], double *dist, float grad[3], int *found) {
    SETTLE(m);
    if (m && m->is_group) return pos ? group::get_sdf_and_gradient(m, pos, dist, grad, found) : fail(CHISEL_HIP_ERR_INVALID, "null argument");
    if (!m || !pos) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    return query_sdf(m, pos, 1, dist, grad, found);
}

}  // extern "C"
namespace {
// SaveMeshPLYASCII (io/PLY.cpp:29-88) over the concatenated meshes: same text format, same number formatting (operator<< of float / int)
void write_ply(std::ofstream &stream, const std::vector<MeshView> &views, size_t numPoints, bool any_color, const int64_t *indices = nullptr,
               size_t n_indices = 0) {
    stream << "ply" << std::endl;
    stream << "format ascii 1.0" << std::endl;
    stream << "element vertex " << numPoints << std::endl;
    stream << "property float x" << std::endl;
    stream << "property float y" << std::endl;
    stream << "property float z" << std::endl;
    if (any_color) {
        stream << "property uchar red" << std::endl;
        stream << "property uchar green" << std::endl;
        stream << "property uchar blue" << std::endl;
    }
    stream << "element face " << numPoints / 3 << std::endl;
    stream << "property list uchar int vertex_index" << std::endl;
    stream << "end_header" << std::endl;
    for (const MeshView &mv : views) {
        for (size_t i = 0; i + 2 < mv.n_v * 3; i += 3) {
            stream << mv.v[i] << " " << mv.v[i + 1] << " " << mv.v[i + 2];
            if (any_color) {
                const int r = static_cast<int>(mv.c[i] * 255.0f);
                const int g = static_cast<int>(mv.c[i + 1] * 255.0f);
                const int b = static_cast<int>(mv.c[i + 2] * 255.0f);
                stream << " " << r << " " << g << " " << b;
            }
            stream << std::endl;
        }
    }
    if (indices) {  // a caller's mesh: its own index list (PLY.cpp:74-84)
        for (size_t i = 0; i + 2 < n_indices; i += 3) {
            stream << "3 ";
            for (int j = 0; j < 3; j++) stream << indices[i + j] << " ";
            stream << std::endl;
        }
        return;
    }
    for (size_t i = 0; i < numPoints; i += 3) {
        stream << "3 ";
        for (int j = 0; j < 3; j++) stream << (i + j) << " ";
        stream << std::endl;
    }
}

}  // namespace
extern "C" {

// Chisel::SaveAllMeshesToPLY (Chisel.cpp:69-105) + SaveMeshPLYASCII (io/PLY.cpp:29-88): same text format, same number
// formatting (operator<< of float / int).  The reference concatenates the meshes in the iteration order of its
// std::unordered_map (unspecified); here chunks are written in ascending id order.
int chisel_hip_save_ply(chisel_hip_map *m, const char *path) {
    SETTLE(m);
    if (m && m->is_group) return path ? group::save_ply(m, path) : fail(CHISEL_HIP_ERR_INVALID, "null argument");
    if (!m || !path) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    std::ofstream stream(path);
    if (!stream) return fail(CHISEL_HIP_ERR_IO, std::string("cannot open ") + path);
    HIP_TRY(hipSetDevice(m->device));
    int rc_p = resolve_pending_meshes(m);
    if (rc_p) return rc_p;
    const std::vector<uint64_t> keys = sorted_mesh_keys(m);
    size_t numPoints = 0;
    bool any_color = false;
    std::vector<MeshView> views(keys.size());
    for (size_t i = 0; i < keys.size(); i++) {
        int rc = view_mesh(m, m->meshes.at(keys[i]), views[i]);
        if (rc) return rc;
        numPoints += views[i].n_v;
        any_color = any_color || (views[i].n_v && views[i].c);
    }
    write_ply(stream, views, numPoints, any_color);
    if (!stream) return fail(CHISEL_HIP_ERR_IO, std::string("write failed: ") + path);
    return CHISEL_HIP_OK;
}

// SaveMeshPLYASCII(fileName, mesh) (io/PLY.cpp:29-88) for a caller's own mesh: vertices (3 floats each), colours in [0, 1] or NULL,
// the face list as the mesh's indices give it (three per face).  Host I/O only: no device is touched.
int chisel_hip_write_mesh_ply(const char *path, const float *vertices, const float *colors, int64_t n_vertices, const int64_t *indices, int64_t n_indices) {
    if (!path || n_vertices < 0 || n_indices < 0 || (n_vertices && !vertices) || (n_indices && !indices)) return fail(CHISEL_HIP_ERR_INVALID, "bad argument");
    std::ofstream stream(path);
    if (!stream) return fail(CHISEL_HIP_ERR_IO, std::string("cannot open ") + path);
    MeshView mv;
    mv.v = vertices;
    mv.c = colors;
    mv.n_v = (size_t)n_vertices;
    write_ply(stream, std::vector<MeshView>(1, mv), (size_t)n_vertices, colors != nullptr, indices, (size_t)n_indices);
    if (!stream) return fail(CHISEL_HIP_ERR_IO, std::string("write failed: ") + path);
    return CHISEL_HIP_OK;
}

}  // extern "C"
