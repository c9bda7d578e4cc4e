// host_group.h -- one TSDF map spread over several GPUs of the node INSIDE ONE PROCESS (chisel_hip_create_group).
//
// The reference has one chisel::Chisel object per map (Chisel.h:38-230) and its caller is one process with one thread
// (chisel_ros: ros::spin); a C++ caller that links libchisel_hip.so therefore cannot start one process per GPU the way
// bench.py / cvids_amd/sharded.py do.  A group handle is a chisel_hip_map* like any other: every entry point of chisel_hip.h
// works on it, the library does what the ranks of the multi-process form do --
//   * one shard map per entry of device_ids[] (n_shards = n, shard_rank = i: chunk_owner() decides who holds a chunk; the same
//     device may be named several times, which is how the tests run on one GPU);
//   * integrate: every shard is handed every frame, in order (host frames as they are; device frames are used in place by the
//     shards on the frame's device and copied peer-to-peer, on a copy stream ordered with events, for the others), and all
//     shards integrate concurrently on their own streams;
//   * UpdateMeshes: the union of the shards' meshesToUpdate is formed, every shard meshes the ids it owns after importing the
//     neighbour chunks it lacks from their owners as ghosts (payload exported into / imported from HBM, peer copy between
//     devices) -- the protocol of cvids_amd/sharded.py: ShardedChisel.UpdateMeshes without the collectives;
//   * queries are routed to the owner (chunks, meshes, SDF) or merged in ascending id order (listings, PLY, map dump).
// No data-path collective is involved: a voxel has one owner and no reduction exists on this path.
#pragma once

namespace {
namespace group {

struct Stage {                       // device frames copied to a shard on another device (two sets alternate per batch)
    float *depth[2] = {nullptr, nullptr};
    uint8_t *color[2] = {nullptr, nullptr};
    size_t depth_elems = 0, color_bytes = 0;   // per frame
    hipStream_t copy = nullptr;
    hipEvent_t ready[2] = {nullptr, nullptr}, consumed[2] = {nullptr, nullptr};
    bool armed[2] = {false, false};
    unsigned turn = 0;
};

// HOST frames of a group (what chisel_ros hands over: Conversions.h:107-200) reach the devices ONCE: one copy over the bus into a staging
// set on the launch set's ingest device (the devices take turns), from there RCCL broadcasts it to the other devices of the group over
// xGMI (ncclBroadcast on every device's copy stream, one grouped call), and the shards that share a device read the same staged copy.
// Before round 5 every shard staged its own copy: eight shards, eight reads of one 2.1 MB frame over PCIe.  RCCL is loaded at run time
// (librccl.so, only when the group spans two or more devices: a process that also holds torch's copy of the library meets it once).
struct Rccl {
    typedef int (*InitAll)(void **, int, const int *);
    typedef int (*Bcast)(const void *, void *, size_t, int, int, void *, hipStream_t);
    typedef int (*Void)();
    typedef int (*Destroy)(void *);
    InitAll comm_init_all = nullptr;
    Bcast broadcast = nullptr;
    Void group_start = nullptr, group_end = nullptr;
    Destroy comm_destroy = nullptr;
    bool load() {
        void *h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return false;
        comm_init_all = reinterpret_cast<InitAll>(dlsym(h, "ncclCommInitAll"));
        broadcast = reinterpret_cast<Bcast>(dlsym(h, "ncclBroadcast"));
        group_start = reinterpret_cast<Void>(dlsym(h, "ncclGroupStart"));
        group_end = reinterpret_cast<Void>(dlsym(h, "ncclGroupEnd"));
        comm_destroy = reinterpret_cast<Destroy>(dlsym(h, "ncclCommDestroy"));
        return comm_init_all && broadcast && group_start && group_end && comm_destroy;
    }
};
struct DeviceStage {                 // per distinct device of the group: two staging sets that alternate per launch set
    int device = 0;
    float *depth[2] = {nullptr, nullptr};
    uint8_t *color[2] = {nullptr, nullptr};
    hipStream_t copy = nullptr;
    hipEvent_t ready[2] = {nullptr, nullptr};
    void *comm = nullptr;            // ncclComm_t of this device (groups over two or more devices)
};
struct HostFanout {
    std::vector<DeviceStage> dev;    // distinct devices, in order of first appearance
    std::vector<int> dev_of_shard;   // index into dev
    std::vector<hipEvent_t> consumed[2];  // [set][shard]: the shard has integrated the launch set that used this staging set
    bool armed[2] = {false, false};
    size_t depth_elems = 0, color_bytes = 0;  // per frame
    unsigned turn = 0;
    Rccl rccl;
    bool rccl_ready = false;
};

// One issuing host thread per shard.  A launch set costs the host 35-120 us to issue (five to six kernel launches, a handful of event
// calls); eight shards issued in turn by the caller's thread cost eight times that per batch, which capped a group below the rate of ONE
// map.  The shard maps share nothing, so every call that fans out over the shards -- integrate, the phases of update_meshes, the id
// listings -- hands shard i's part to worker i and joins.  Workers spin for a while after a job (a stream of frames keeps them hot: a
// condition-variable wake-up costs as much as the job) and then sleep; shard 0's part runs on the calling thread.
// CHISEL_HIP_GROUP_THREADS=0: everything on the calling thread, in turn (A/B, debugging).
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}
struct Pool {
    std::vector<std::thread> workers;           // worker w serves shard w + 1
    std::mutex fan;                             // one fan-out at a time: a second caller thread (a mesh / publish thread beside the integrating one) waits its turn
    std::function<int(int)> job;
    std::atomic<uint64_t> seq{0};               // bumped once per fan-out
    std::atomic<int> done{0};
    std::atomic<int> sleepers{0};
    std::atomic<bool> stop{false};
    std::vector<int> rc;
    std::vector<std::string> err;
    std::mutex mu;
    std::condition_variable cv;
    bool serial = false;
    void loop(int shard) {
        uint64_t seen = 0;
        for (;;) {
            const auto idle = std::chrono::steady_clock::now();
            while (seq.load(std::memory_order_acquire) == seen && !stop.load(std::memory_order_relaxed)) {
                if (std::chrono::steady_clock::now() - idle > std::chrono::microseconds(500)) {
                    std::unique_lock<std::mutex> lk(mu);
                    sleepers.fetch_add(1);
                    // (sleepers += 1, then seq read; the issuing side: seq += 1, then sleepers read -- all four sequentially consistent, so one
                    // of the two sees the other: no lost wake-up)
                    cv.wait(lk, [&] { return seq.load() != seen || stop.load(); });
                    sleepers.fetch_sub(1);
                } else {
                    cpu_relax();
                }
            }
            if (stop.load()) return;
            seen = seq.load(std::memory_order_acquire);
            int r;
            try {
                r = job(shard);
            } catch (const std::exception &e) {  // (bad_alloc from a vector inside a job: an error of the call, not the end of the process)
                r = fail(CHISEL_HIP_ERR_INVALID, std::string("group worker: ") + e.what());
            }
            rc[(size_t)shard] = r;
            if (r) err[(size_t)shard] = g_last_error;
            done.fetch_add(1, std::memory_order_release);
        }
    }
};
// what a shard keeps for the group's recomputes (all on its own device, grow-only, reused by every recompute) with the events that
// order the steps between the shards' streams: the gathered list of updated chunks, the segments it sends and the segments it receives
struct MeshStage {
    int *gathered = nullptr;                   // [W][1 + 4 * cap] ints: every shard's dirty list (its own block is written by its own kernel)
    int cap = 0;
    unsigned char *send = nullptr, *recv = nullptr;
    long long send_cap = 0, recv_cap = 0;
    hipEvent_t listed = nullptr, gathered_ev = nullptr, exported = nullptr, copied = nullptr, imported = nullptr;
    bool armed = false;                        // `copied` / `imported` have been recorded at least once
    int *status = nullptr;                     // the wait-free form: [W + 2][8] ints -- row 0 the all-reduced vector, row 1 this shard's own (its export kernel writes it), rows 2.. every shard's, copied here
};
struct MeshStages {
    std::vector<MeshStage> of;                 // [shard]
    std::vector<hipStream_t> copy;             // [r]: the peer copies that bring the other shards' lists and segments to shard r
    int cap = 1 << 12;                         // entries per shard in the gathered list (doubled when a shard has more)
    // the wait-free form of the recompute (update_meshes_wait_free): what the previous recompute needed (all shards' maximum), whether one is
    // in flight whose status the host has not looked at (settle), where shard 0's all-reduced status reaches the host
    bool have_sizes = false, unsettled = false, retrying = false;
    int64_t need_max_count = 0, need_seg_bytes = 0, need_jobs = 0, need_recv_items = 0, need_send_items = 0;
    long long stride = 0;
    int *status_host = nullptr;                // pinned [8]
    hipEvent_t status_ev = nullptr;
    int wait_free_recomputes = 0, called_off = 0;
};

inline int n_shards(const chisel_hip_map *g) { return (int)g->shards.size(); }
inline Pool *pool_of(const chisel_hip_map *g) { return static_cast<Pool *>(g->pool); }

// fn(i) for every shard i, concurrently (shard 0 on the calling thread); the first failure's code and message
template <class F>
int run_shards(chisel_hip_map *g, F fn) {
    Pool *P = pool_of(g);
    const int W = n_shards(g);
    if (!P || P->serial || W == 1) {
        for (int i = 0; i < W; i++) {
            const int rc = fn(i);
            if (rc) return rc;
        }
        return CHISEL_HIP_OK;
    }
    std::lock_guard<std::mutex> one_at_a_time(P->fan);
    P->job = fn;
    P->done.store(0, std::memory_order_relaxed);
    P->seq.fetch_add(1);
    if (P->sleepers.load() > 0) {
        std::lock_guard<std::mutex> lk(P->mu);
        P->cv.notify_all();
    }
    int rc0;
    try {
        rc0 = fn(0);
    } catch (const std::exception &e) {
        rc0 = fail(CHISEL_HIP_ERR_INVALID, std::string("group: ") + e.what());
    }
    for (unsigned spins = 0; P->done.load(std::memory_order_acquire) < W - 1; spins++) {
        if (spins < (1u << 14)) cpu_relax();
        else std::this_thread::yield();  // (a worker that was descheduled: give it the core)
    }
    if (rc0) return rc0;
    for (int i = 1; i < W; i++)
        if (P->rc[(size_t)i]) return fail(P->rc[(size_t)i], P->err[(size_t)i]);
    return CHISEL_HIP_OK;
}
inline int owner_of(const chisel_hip_map *g, const int id[3]) { return chunk_owner(id[0], id[1], id[2], n_shards(g), g->cfg.shard_block); }
inline bool id_less(const int *a, const int *b) { return a[0] != b[0] ? a[0] < b[0] : (a[1] != b[1] ? a[1] < b[1] : a[2] < b[2]); }

int create(const chisel_hip_config *cfg, const int *device_ids, int n, chisel_hip_map **out) {
    if (!cfg || !out || !device_ids || n < 1 || n > 64) return fail(CHISEL_HIP_ERR_INVALID, "bad device list");
    *out = nullptr;
    chisel_hip_map *g = new chisel_hip_map();
    g->cfg = *cfg;
    g->cfg.n_shards = n;
    g->cfg.shard_rank = 0;
    g->cfg.shard_block = cfg->shard_block < 1 ? 2 : cfg->shard_block;
    g->is_group = true;
    for (int i = 0; i < n; i++) {
        chisel_hip_config c = *cfg;
        c.device_id = device_ids[i];
        c.n_shards = n;
        c.shard_rank = i;
        chisel_hip_map *s = nullptr;
        const int rc = chisel_hip_create(&c, &s);
        if (rc) {
            for (chisel_hip_map *p : g->shards) chisel_hip_destroy(p);
            delete g;
            return rc;
        }
        g->shards.push_back(s);
    }
    g->N = g->shards[0]->N;
    g->V = g->shards[0]->V;
    g->device = g->shards[0]->device;
    g->stages = new std::vector<Stage>(n);
    {
        HostFanout *F = new HostFanout();
        for (int i = 0; i < n; i++) {
            int k = -1;
            for (size_t d = 0; d < F->dev.size(); d++)
                if (F->dev[d].device == g->shards[(size_t)i]->device) k = (int)d;
            if (k < 0) {
                F->dev.emplace_back();
                F->dev.back().device = g->shards[(size_t)i]->device;
                k = (int)F->dev.size() - 1;
            }
            F->dev_of_shard.push_back(k);
        }
        F->consumed[0].assign((size_t)n, nullptr);
        F->consumed[1].assign((size_t)n, nullptr);
        g->host_fanout = F;
    }
    {
        Pool *P = new Pool();
        P->rc.assign((size_t)n, 0);
        P->err.assign((size_t)n, std::string());
        if (const char *e = getenv("CHISEL_HIP_GROUP_THREADS")) P->serial = atoi(e) == 0;
        if (!P->serial)
            for (int i = 1; i < n; i++) P->workers.emplace_back([P, i] { P->loop(i); });
        g->pool = P;
        MeshStages *MS = new MeshStages();
        MS->of.resize((size_t)n);
        MS->copy.assign((size_t)n, nullptr);
        g->mesh_stages_group = MS;
    }
    // peer access between the devices of the group (a failure here only costs the copies their direct path)
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
            if (g->shards[i]->device != g->shards[j]->device) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, g->shards[i]->device, g->shards[j]->device) == hipSuccess && can) {
                    (void)hipSetDevice(g->shards[i]->device);
                    (void)hipDeviceEnablePeerAccess(g->shards[j]->device, 0);
                    (void)hipGetLastError();
                }
            }
    *out = g;
    return CHISEL_HIP_OK;
}

int destroy(chisel_hip_map *g) {
    if (Pool *P = pool_of(g)) {
        {
            std::lock_guard<std::mutex> lk(P->mu);
            P->stop.store(true);
            P->cv.notify_all();
        }
        for (std::thread &t : P->workers) t.join();
        delete P;
        g->pool = nullptr;
    }
    if (MeshStages *MS = static_cast<MeshStages *>(g->mesh_stages_group)) {
        for (chisel_hip_map *sh : g->shards) (void)chisel_hip_synchronize(sh);
        const int W = n_shards(g);
        for (int i = 0; i < W; i++) {
            MeshStage &S = MS->of[(size_t)i];
            (void)hipSetDevice(g->shards[(size_t)i]->device);
            for (void *p : {(void *)S.gathered, (void *)S.send, (void *)S.recv, (void *)S.status})
                if (p) (void)hipFree(p);
            for (hipEvent_t e : {S.listed, S.gathered_ev, S.exported, S.copied, S.imported})
                if (e) (void)hipEventDestroy(e);
        }
        for (int r = 0; r < W; r++)
            if (MS->copy[(size_t)r]) {
                (void)hipSetDevice(g->shards[r]->device);
                (void)hipStreamDestroy(MS->copy[(size_t)r]);
            }
        if (MS->status_host) (void)hipHostFree(MS->status_host);
        if (MS->status_ev) (void)hipEventDestroy(MS->status_ev);
        delete MS;
        g->mesh_stages_group = nullptr;
    }
    if (HostFanout *F = static_cast<HostFanout *>(g->host_fanout)) {
        for (chisel_hip_map *sh : g->shards) (void)chisel_hip_synchronize(sh);
        for (DeviceStage &D : F->dev) {
            (void)hipSetDevice(D.device);
            if (D.copy) (void)hipStreamSynchronize(D.copy);
            if (D.comm && F->rccl.comm_destroy) (void)F->rccl.comm_destroy(D.comm);
            for (int b = 0; b < 2; b++) {
                if (D.depth[b]) (void)hipFree(D.depth[b]);
                if (D.color[b]) (void)hipFree(D.color[b]);
                if (D.ready[b]) (void)hipEventDestroy(D.ready[b]);
            }
            if (D.copy) (void)hipStreamDestroy(D.copy);
        }
        for (int b = 0; b < 2; b++)
            for (size_t i = 0; i < F->consumed[b].size(); i++)
                if (F->consumed[b][i]) {
                    (void)hipSetDevice(g->shards[i]->device);
                    (void)hipEventDestroy(F->consumed[b][i]);
                }
        delete F;
        g->host_fanout = nullptr;
    }
    std::vector<Stage> *st = static_cast<std::vector<Stage> *>(g->stages);
    for (size_t i = 0; i < g->shards.size(); i++) {
        if (st) {
            Stage &S = (*st)[i];
            (void)hipSetDevice(g->shards[i]->device);
            if (S.copy) (void)hipStreamSynchronize(S.copy);
            for (int b = 0; b < 2; b++) {
                if (S.depth[b]) (void)hipFree(S.depth[b]);
                if (S.color[b]) (void)hipFree(S.color[b]);
                if (S.ready[b]) (void)hipEventDestroy(S.ready[b]);
                if (S.consumed[b]) (void)hipEventDestroy(S.consumed[b]);
            }
            if (S.copy) (void)hipStreamDestroy(S.copy);
        }
        chisel_hip_destroy(g->shards[i]);
    }
    delete st;
    delete g;
    return CHISEL_HIP_OK;
}

template <class F>
int for_all(chisel_hip_map *g, F fn) {
    for (chisel_hip_map *s : g->shards) {
        const int rc = fn(s);
        if (rc) return rc;
    }
    return CHISEL_HIP_OK;
}

int device_of(const void *p, int fallback) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type == hipMemoryTypeDevice) return attr.device;
    (void)hipGetLastError();
    return fallback;
}

// n <= KMAX frames of one image size to every shard
int integrate_set(chisel_hip_map *g, int n, const chisel_hip_depth_frame *frames, const chisel_hip_color_frame *colors) {
    std::vector<Stage> &st = *static_cast<std::vector<Stage> *>(g->stages);
    // test hook: device frames take the staging path (peer copy on the copy stream, event hand-over) even when they already live on the
    // shard's device -- the box the tests run on has one GPU
    static const bool force_stage = getenv("CHISEL_HIP_GROUP_FORCE_STAGE") != nullptr;
    const size_t npx = (size_t)frames[0].width * frames[0].height;
    size_t cbytes = 0;
    if (colors)
        for (int k = 0; k < n; k++) cbytes = std::max(cbytes, (size_t)colors[k].width * colors[k].height * colors[k].channels);
    // ---- host frames: staged once (HostFanout), every shard then sees device frames of its own device
    HostFanout &F = *static_cast<HostFanout *>(g->host_fanout);
    bool host_frames = n_shards(g) > 1;
    for (int k = 0; k < n; k++) host_frames = host_frames && !frames[k].on_device && (!colors || !colors[k].on_device);
    int fb = -1;
    if (host_frames) {
        const int ND = (int)F.dev.size();
        if (ND > 1 && !F.rccl_ready) {
            if (!F.rccl.load()) return fail(CHISEL_HIP_ERR_HIP, "group over several devices: librccl.so could not be loaded (frame fan-out over xGMI)");
            std::vector<void *> comms((size_t)ND, nullptr);
            std::vector<int> devs;
            for (const DeviceStage &D : F.dev) devs.push_back(D.device);
            if (F.rccl.comm_init_all(comms.data(), ND, devs.data()) != 0) return fail(CHISEL_HIP_ERR_HIP, "ncclCommInitAll failed for the group's devices");
            for (int d = 0; d < ND; d++) F.dev[(size_t)d].comm = comms[(size_t)d];
            F.rccl_ready = true;
        }
        // buffers (both sets of every device) and events
        if (npx > F.depth_elems || cbytes > F.color_bytes || !F.dev[0].copy) {
            for (chisel_hip_map *sh : g->shards) {
                int rc = chisel_hip_synchronize(sh);
                if (rc) return rc;
            }
            const size_t de = std::max(npx, F.depth_elems), cb = std::max(cbytes, F.color_bytes);
            for (DeviceStage &D : F.dev) {
                HIP_TRY(hipSetDevice(D.device));
                if (!D.copy) {
                    HIP_TRY(hipStreamCreateWithFlags(&D.copy, hipStreamNonBlocking));
                    for (int b = 0; b < 2; b++) HIP_TRY(hipEventCreateWithFlags(&D.ready[b], hipEventDisableTiming));
                }
                HIP_TRY(hipStreamSynchronize(D.copy));
                for (int b = 0; b < 2; b++) {
                    if (D.depth[b]) HIP_TRY(hipFree(D.depth[b]));
                    if (D.color[b]) HIP_TRY(hipFree(D.color[b]));
                    D.depth[b] = nullptr;
                    D.color[b] = nullptr;
                    HIP_TRY(hipMalloc(&D.depth[b], de * KMAX * sizeof(float)));
                    if (cb) HIP_TRY(hipMalloc(&D.color[b], cb * KMAX));
                }
            }
            for (int b = 0; b < 2; b++) {
                F.armed[b] = false;
                for (size_t i = 0; i < g->shards.size(); i++)
                    if (!F.consumed[b][i]) {
                        HIP_TRY(hipSetDevice(g->shards[i]->device));
                        HIP_TRY(hipEventCreateWithFlags(&F.consumed[b][i], hipEventDisableTiming));
                    }
            }
            F.depth_elems = de;
            F.color_bytes = cb;
        }
        fb = (int)(F.turn & 1u);
        const int root = (int)((F.turn >> 1) % (unsigned)ND);  // the ingest device: the devices take turns
        F.turn++;
        // the launch set that last used this staging set has been integrated by every shard
        for (DeviceStage &D : F.dev) {
            HIP_TRY(hipSetDevice(D.device));
            if (F.armed[fb])
                for (size_t i = 0; i < g->shards.size(); i++) HIP_TRY(hipStreamWaitEvent(D.copy, F.consumed[fb][i], 0));
        }
        DeviceStage &R = F.dev[(size_t)root];
        HIP_TRY(hipSetDevice(R.device));
        for (int k = 0; k < n; k++) {
            HIP_TRY(hipMemcpyAsync(R.depth[fb] + (size_t)k * F.depth_elems, frames[k].depth, npx * sizeof(float), hipMemcpyHostToDevice, R.copy));
            if (colors)
                HIP_TRY(hipMemcpyAsync(R.color[fb] + (size_t)k * F.color_bytes, colors[k].color, (size_t)colors[k].width * colors[k].height * colors[k].channels,
                                       hipMemcpyHostToDevice, R.copy));
        }
        if (ND > 1) {
            // xGMI fan-out: one broadcast of the depth block and one of the colour block per device, all in one group call
            if (F.rccl.group_start() != 0) return fail(CHISEL_HIP_ERR_HIP, "ncclGroupStart failed");
            for (int d = 0; d < ND; d++) {
                DeviceStage &D = F.dev[(size_t)d];
                // (a failed call must not leave the group open: every later RCCL call of the process would join it)
                if (F.rccl.broadcast(R.depth[fb], D.depth[fb], (size_t)n * F.depth_elems, /* ncclFloat32 */ 7, root, D.comm, D.copy) != 0) {
                    (void)F.rccl.group_end();
                    return fail(CHISEL_HIP_ERR_HIP, "ncclBroadcast (depth) failed");
                }
                if (colors && F.rccl.broadcast(R.color[fb], D.color[fb], (size_t)n * F.color_bytes, /* ncclUint8 */ 1, root, D.comm, D.copy) != 0) {
                    (void)F.rccl.group_end();
                    return fail(CHISEL_HIP_ERR_HIP, "ncclBroadcast (colour) failed");
                }
            }
            if (F.rccl.group_end() != 0) return fail(CHISEL_HIP_ERR_HIP, "ncclGroupEnd failed");
        }
        for (DeviceStage &D : F.dev) {
            HIP_TRY(hipSetDevice(D.device));
            HIP_TRY(hipEventRecord(D.ready[fb], D.copy));
        }
    }
    const int rc_all = run_shards(g, [&](int i) -> int {
        chisel_hip_map *s = g->shards[(size_t)i];
        std::vector<chisel_hip_depth_frame> f(frames, frames + n);
        std::vector<chisel_hip_color_frame> c;
        if (colors) c.assign(colors, colors + n);
        if (host_frames) {
            // the staged copy on this shard's device, ready behind the device's event; `consumed` tells the next user of the staging set
            const DeviceStage &D = F.dev[(size_t)F.dev_of_shard[(size_t)i]];
            HIP_TRY(hipSetDevice(s->device));
            for (int k = 0; k < n; k++) {
                f[k].depth = D.depth[fb] + (size_t)k * F.depth_elems;
                f[k].on_device = 1;
                if (colors) {
                    c[k].color = D.color[fb] + (size_t)k * F.color_bytes;
                    c[k].on_device = 1;
                }
            }
            int rc = chisel_hip_wait_event(s, D.ready[fb]);
            if (rc) return rc;
            rc = chisel_hip_integrate_batch(s, n, f.data(), colors ? c.data() : nullptr);
            if (rc) return rc;
            return chisel_hip_record_event(s, F.consumed[fb][(size_t)i]);
        }
        bool foreign = false;
        for (int k = 0; k < n; k++) {
            foreign |= f[k].on_device && (force_stage || device_of(f[k].depth, s->device) != s->device);
            if (colors) foreign |= c[k].on_device && (force_stage || device_of(c[k].color, s->device) != s->device);
        }
        Stage &S = st[(size_t)i];
        int b = -1;
        if (foreign) {
            HIP_TRY(hipSetDevice(s->device));
            if (!S.copy) {
                HIP_TRY(hipStreamCreateWithFlags(&S.copy, hipStreamNonBlocking));
                for (int q = 0; q < 2; q++) {
                    HIP_TRY(hipEventCreateWithFlags(&S.ready[q], hipEventDisableTiming));
                    HIP_TRY(hipEventCreateWithFlags(&S.consumed[q], hipEventDisableTiming));
                }
            }
            if (npx > S.depth_elems || cbytes > S.color_bytes) {
                HIP_TRY(hipStreamSynchronize(S.copy));
                int rc = chisel_hip_synchronize(s);
                if (rc) return rc;
                for (int q = 0; q < 2; q++) {
                    if (S.depth[q]) HIP_TRY(hipFree(S.depth[q]));
                    if (S.color[q]) HIP_TRY(hipFree(S.color[q]));
                    S.depth[q] = nullptr;
                    S.color[q] = nullptr;
                    HIP_TRY(hipMalloc(&S.depth[q], std::max(npx, S.depth_elems) * KMAX * sizeof(float)));
                    if (std::max(cbytes, S.color_bytes)) HIP_TRY(hipMalloc(&S.color[q], std::max(cbytes, S.color_bytes) * KMAX));
                    S.armed[q] = false;
                }
                S.depth_elems = std::max(npx, S.depth_elems);
                S.color_bytes = std::max(cbytes, S.color_bytes);
            }
            b = (int)(S.turn++ & 1u);
            if (S.armed[b]) HIP_TRY(hipStreamWaitEvent(S.copy, S.consumed[b], 0));  // the batch that last used this set has been integrated
            // the caller's event (chisel_hip_wait_event on the group) guards the frames the peer copies read; the shard itself then
            // waits for the copies (S.ready below), which orders it behind the caller's event as well
            if (s->input_event) HIP_TRY(hipStreamWaitEvent(S.copy, s->input_event, 0));
            for (int k = 0; k < n; k++) {
                if (f[k].on_device) {
                    const int src = device_of(f[k].depth, s->device);
                    if (src != s->device || force_stage) {
                        float *dst = S.depth[b] + (size_t)k * S.depth_elems;
                        HIP_TRY(hipMemcpyPeerAsync(dst, s->device, f[k].depth, src, npx * sizeof(float), S.copy));
                        f[k].depth = dst;
                    }
                }
                if (colors && c[k].on_device) {
                    const int src = device_of(c[k].color, s->device);
                    if (src != s->device || force_stage) {
                        uint8_t *dst = S.color[b] + (size_t)k * S.color_bytes;
                        HIP_TRY(hipMemcpyPeerAsync(dst, s->device, c[k].color, src, (size_t)c[k].width * c[k].height * c[k].channels, S.copy));
                        c[k].color = dst;
                    }
                }
            }
            HIP_TRY(hipEventRecord(S.ready[b], S.copy));
            int rc = chisel_hip_wait_event(s, S.ready[b]);
            if (rc) return rc;
        }
        int rc = chisel_hip_integrate_batch(s, n, f.data(), colors ? c.data() : nullptr);
        if (rc) return rc;
        if (foreign) {
            rc = chisel_hip_record_event(s, S.consumed[b]);
            if (rc) return rc;
            S.armed[b] = true;
        }
        return CHISEL_HIP_OK;
    });
    if (!rc_all && fb >= 0) F.armed[fb] = true;
    return rc_all;
}

// frames in order; consecutive frames of one image size go out KMAX at a time (as integrate_frames cuts them for one map)
int integrate(chisel_hip_map *g, int n, const chisel_hip_depth_frame *frames, const chisel_hip_color_frame *colors) {
    if (n < 0 || (n > 0 && !frames)) return fail(CHISEL_HIP_ERR_INVALID, "bad frame list");
    {
        const int rc_s = settle(g);  // (a wait-free recompute in flight: its status before the maps change)
        if (rc_s) return rc_s;
    }
    // chisel_hip_wait_event on the group: the event covers every frame of this call, so every launch set arms its shards with it
    const hipEvent_t call_input = g->input_event;
    g->input_event = nullptr;
    int i = 0, rc = CHISEL_HIP_OK;
    while (i < n && !rc) {
        int run = 1;
        while (i + run < n && run < KMAX && frames[i + run].width == frames[i].width && frames[i + run].height == frames[i].height) run++;
        for (chisel_hip_map *s : g->shards) s->input_event = call_input;
        rc = integrate_set(g, run, frames + i, colors ? colors + i : nullptr);
        i += run;
    }
    for (chisel_hip_map *s : g->shards) s->input_event = nullptr;
    return rc;
}

int integrate_cloud(chisel_hip_map *g, const chisel_hip_pointcloud *cloud) {
    if (!cloud) return fail(CHISEL_HIP_ERR_INVALID, "null cloud");
    if (cloud->on_device) {
        for (chisel_hip_map *s : g->shards)
            if (device_of(cloud->points, s->device) != s->device)
                return fail(CHISEL_HIP_ERR_UNSUPPORTED, "a device-resident point cloud must live on the device of every shard of the group: pass a host cloud");
    }
    const hipEvent_t call_input = g->input_event;  // chisel_hip_wait_event on the group
    g->input_event = nullptr;
    return for_all(g, [&](chisel_hip_map *s) {
        s->input_event = call_input;
        return chisel_hip_integrate_pointcloud(s, cloud);
    });
}

int garbage_collect(chisel_hip_map *g, const int *ids, int n) {
    if (n < 0 || (n > 0 && !ids)) return fail(CHISEL_HIP_ERR_INVALID, "bad id list");
    std::vector<std::vector<int>> per(g->shards.size());
    for (int j = 0; j < n; j++) {
        std::vector<int> &v = per[owner_of(g, ids + 3 * j)];
        v.insert(v.end(), ids + 3 * j, ids + 3 * j + 3);
    }
    for (size_t i = 0; i < g->shards.size(); i++)
        if (!per[i].empty()) {
            const int rc = chisel_hip_garbage_collect(g->shards[i], per[i].data(), (int)per[i].size() / 3);
            if (rc) return rc;
        }
    return CHISEL_HIP_OK;
}

// sorted union / concatenation of per-shard id listings
template <class F>
int gather_ids(chisel_hip_map *g, F list_fn, bool unique, std::vector<int> &out) {
    std::vector<std::vector<int>> per(g->shards.size());
    int rc_all = run_shards(g, [&](int i) -> int {
        chisel_hip_map *s = g->shards[(size_t)i];
        int64_t n = 0;
        int rc = list_fn(s, nullptr, 0, &n);
        if (rc) return rc;
        std::vector<int> &ids = per[(size_t)i];
        ids.resize((size_t)n * 3);
        if (n) {
            rc = list_fn(s, ids.data(), n, &n);
            if (rc) return rc;
            ids.resize((size_t)n * 3);
        }
        return CHISEL_HIP_OK;
    });
    if (rc_all) return rc_all;
    std::vector<std::array<int, 3>> all;
    for (const std::vector<int> &ids : per)
        for (size_t j = 0; j + 2 < ids.size(); j += 3) all.push_back({ids[j], ids[j + 1], ids[j + 2]});
    std::sort(all.begin(), all.end());
    if (unique) all.erase(std::unique(all.begin(), all.end()), all.end());
    out.clear();
    for (const auto &a : all) out.insert(out.end(), a.begin(), a.end());
    return CHISEL_HIP_OK;
}
int emit_ids(const std::vector<int> &all, int *ids, int64_t max_ids, int64_t *count) {
    if (!count) return fail(CHISEL_HIP_ERR_INVALID, "null count");
    *count = (int64_t)all.size() / 3;
    if (ids) memcpy(ids, all.data(), (size_t)std::min<int64_t>(max_ids, *count) * 3 * sizeof(int));
    return CHISEL_HIP_OK;
}

// Chisel::UpdateMeshes of the group: the protocol of cvids_amd/sharded.py: ShardedChisel.UpdateMeshes -- planned on the device
// (kernels_map.h: ShellPlan) -- with peer copies between the shards' devices where the multi-process form has its two collectives.
// Four fan-outs over the shards (one issuing thread each); the only host wait of a shard is the one inside its plan call:
//   A  every shard lists its updated chunks into its block of its gathered list (chisel_hip_dirty_ids_device) and records `listed`;
//   B  every shard copies the other shards' blocks into its own list (its copy stream, behind their `listed` events) and plans:
//      its jobs, the segments it owes every other shard, how much every owner owes it (chisel_hip_shell_plan_device);
//   C  every OWNER packs its segments (chisel_hip_export_shells_packed) and records `exported`;
//   D  every MESHING shard copies its segments out of the owners' buffers (behind their `exported`), installs the ghosts, recomputes
//      its jobs, drops the ghosts and records `imported`, behind which the next recompute's copies into the same buffer are ordered.
// the shards' gathered lists at MS.cap entries per shard (a list that must grow is grown here, on the calling thread, after everything that
// reads the old one: the previous recompute's plan kernels, the other shards' copies out of it)
int grow_lists(chisel_hip_map *g, MeshStages &MS) {
    const int W = n_shards(g), cap = MS.cap, blk = 1 + 4 * cap;
    bool grow = false;
    for (int i = 0; i < W; i++) grow = grow || MS.of[(size_t)i].cap < cap;
    if (!grow) return CHISEL_HIP_OK;
    for (int i = 0; i < W; i++) {
        int rc0 = chisel_hip_synchronize(g->shards[(size_t)i]);
        if (rc0) return rc0;
    }
    for (int r = 0; r < W; r++) {
        HIP_TRY(hipSetDevice(g->shards[(size_t)r]->device));
        HIP_TRY(hipStreamSynchronize(MS.copy[(size_t)r]));
    }
    for (int i = 0; i < W; i++) {
        MeshStage &S = MS.of[(size_t)i];
        if (S.cap >= cap) continue;
        HIP_TRY(hipSetDevice(g->shards[(size_t)i]->device));
        if (S.gathered) HIP_TRY(hipFree(S.gathered));
        S.gathered = nullptr;
        HIP_TRY(hipMalloc(&S.gathered, (size_t)W * blk * sizeof(int)));
        S.cap = cap;
    }
    return CHISEL_HIP_OK;
}
inline int make_stage_events(MeshStage &S) {
    if (S.listed) return CHISEL_HIP_OK;
    for (hipEvent_t *e : {&S.listed, &S.gathered_ev, &S.exported, &S.copied, &S.imported}) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    return CHISEL_HIP_OK;
}

int update_meshes_blocking(chisel_hip_map *g) {
    const int W = n_shards(g);
    MeshStages &MS = *static_cast<MeshStages *>(g->mesh_stages_group);
    const bool timing = g_host_timer.on;
    auto t_prev = std::chrono::steady_clock::now();
    double t_phase[5] = {0, 0, 0, 0, 0};
    auto lap = [&](int i) {
        if (!timing) return;
        const auto n = std::chrono::steady_clock::now();
        t_phase[i] += std::chrono::duration<double, std::micro>(n - t_prev).count();
        t_prev = n;
    };
    const bool color = g->cfg.use_color != 0;
    auto make_events = [](MeshStage &S) -> int {
        if (S.listed) return CHISEL_HIP_OK;
        for (hipEvent_t *e : {&S.listed, &S.gathered_ev, &S.exported, &S.copied, &S.imported}) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
        return CHISEL_HIP_OK;
    };
    std::vector<std::array<int64_t, 4 + 4 * SHELL_MAX_SHARDS>> plan((size_t)W);
    // Every shard's events and copy stream exist before the first fan-out, and a list that must grow is grown here, on the calling thread:
    // the shard threads below read each other's handles (MS.copy[o], MS.of[o].listed), which must not be coming into being meanwhile.
    for (int i = 0; i < W; i++) {
        HIP_TRY(hipSetDevice(g->shards[(size_t)i]->device));
        int rc0 = make_events(MS.of[(size_t)i]);
        if (rc0) return rc0;
        if (!MS.copy[(size_t)i]) HIP_TRY(hipStreamCreateWithFlags(&MS.copy[(size_t)i], hipStreamNonBlocking));
    }
    for (;;) {
        const int cap = MS.cap, blk = 1 + 4 * cap;
        {
            int rc0 = grow_lists(g, MS);
            if (rc0) return rc0;
        }
        // ---- A
        int rc = run_shards(g, [&](int i) -> int {
            chisel_hip_map *sh = g->shards[(size_t)i];
            MeshStage &S = MS.of[(size_t)i];
            HIP_TRY(hipSetDevice(sh->device));
            int rc2;
            rc2 = chisel_hip_dirty_ids_device(sh, S.gathered + (size_t)i * blk, cap);
            if (rc2) return rc2;
            return chisel_hip_record_event(sh, S.listed);
        });
        if (rc) return rc;
        lap(0);
        // ---- B
        rc = run_shards(g, [&](int r) -> int {
            chisel_hip_map *sh = g->shards[(size_t)r];
            MeshStage &S = MS.of[(size_t)r];
            HIP_TRY(hipSetDevice(sh->device));
            hipStream_t cs = MS.copy[(size_t)r];
            for (int o = 0; o < W; o++) {
                if (o == r) continue;
                HIP_TRY(hipStreamWaitEvent(cs, MS.of[(size_t)o].listed, 0));
                HIP_TRY(hipMemcpyPeerAsync(S.gathered + (size_t)o * blk, sh->device, MS.of[(size_t)o].gathered + (size_t)o * blk, g->shards[(size_t)o]->device, (size_t)blk * sizeof(int), cs));
            }
            HIP_TRY(hipEventRecord(S.gathered_ev, cs));
            int rc2 = chisel_hip_wait_event(sh, S.gathered_ev);
            if (rc2) return rc2;
            return chisel_hip_shell_plan_device(sh, S.gathered, W, cap, plan[(size_t)r].data());
        });
        if (rc) return rc;
        lap(1);
        int64_t mx = 0;
        for (int r = 0; r < W; r++) mx = std::max(mx, plan[(size_t)r][2]);
        if (mx <= cap) break;
        MS.cap = (int)std::min<int64_t>(2 * mx, 1 << 26);  // a shard had more entries than a block holds: gather again with room for them
    }
    // where the segment of pair (owner o -> meshing shard r) lies in send[o] and in recv[r]: bytes from the plans' counts (what o plans
    // to send r is what r plans to receive from o: the same enumeration from both sides)
    auto seg = [&](int o, int r) { return (long long)shell_segment_bytes(plan[(size_t)o][4 + 2 * r], plan[(size_t)o][5 + 2 * r], color); };
    for (int o = 0; o < W; o++)
        for (int r = 0; r < W; r++)
            if (plan[(size_t)o][4 + 2 * r] != plan[(size_t)r][4 + 2 * W + 2 * o] || plan[(size_t)o][5 + 2 * r] != plan[(size_t)r][5 + 2 * W + 2 * o])
                return fail(CHISEL_HIP_ERR_HIP, "group recompute: two shards' plans disagree on a segment");
    // ---- C
    int rc = run_shards(g, [&](int o) -> int {
        chisel_hip_map *src = g->shards[(size_t)o];
        MeshStage &S = MS.of[(size_t)o];
        HIP_TRY(hipSetDevice(src->device));
        long long bytes = 0;
        for (int r = 0; r < W; r++) bytes += seg(o, r);
        // the previous recompute's readers of this buffer: the copies of every meshing shard
        for (int r = 0; r < W; r++)
            if (MS.of[(size_t)r].armed) {
                if (bytes > S.send_cap) HIP_TRY(hipEventSynchronize(MS.of[(size_t)r].copied));
                else HIP_TRY(hipStreamWaitEvent(src->stream, MS.of[(size_t)r].copied, 0));
            }
        if (bytes > S.send_cap) {
            if (S.send) HIP_TRY(hipFree(S.send));
            S.send = nullptr;
            S.send_cap = std::max<long long>(bytes + bytes / 2, 1 << 20);
            HIP_TRY(hipMalloc(&S.send, (size_t)S.send_cap));
        }
        int rc2 = chisel_hip_export_shells_packed(src, S.send, bytes);
        if (rc2) return rc2;
        return chisel_hip_record_event(src, S.exported);
    });
    if (rc) return rc;
    lap(2);
    // ---- D
    std::vector<uint64_t> moved((size_t)W, 0);
    rc = run_shards(g, [&](int r) -> int {
        chisel_hip_map *dst = g->shards[(size_t)r];
        MeshStage &S = MS.of[(size_t)r];
        HIP_TRY(hipSetDevice(dst->device));
        long long bytes = 0;
        for (int o = 0; o < W; o++) bytes += seg(o, r);
        hipStream_t cs = MS.copy[(size_t)r];
        if (bytes > S.recv_cap) {
            if (S.armed) HIP_TRY(hipEventSynchronize(S.imported));
            int rc2 = chisel_hip_synchronize(dst);  // (the previous recompute's drop kernel reads the old buffer)
            if (rc2) return rc2;
            if (S.recv) HIP_TRY(hipFree(S.recv));
            S.recv = nullptr;
            S.recv_cap = std::max<long long>(bytes + bytes / 2, 1 << 20);
            HIP_TRY(hipMalloc(&S.recv, (size_t)S.recv_cap));
        } else if (S.armed) {
            HIP_TRY(hipStreamWaitEvent(cs, S.imported, 0));  // the previous recompute's import and drop have read the buffer
        }
        long long at = 0;
        for (int o = 0; o < W; o++) {
            const long long n = seg(o, r);
            long long from = 0;
            for (int q = 0; q < r; q++) from += seg(o, q);
            HIP_TRY(hipStreamWaitEvent(cs, MS.of[(size_t)o].exported, 0));
            HIP_TRY(hipMemcpyPeerAsync(S.recv + at, dst->device, MS.of[(size_t)o].send + from, g->shards[(size_t)o]->device, (size_t)n, cs));
            at += n;
            moved[(size_t)r] += (uint64_t)plan[(size_t)r][5 + 2 * W + 2 * o] * (color ? 12 : 8);
        }
        HIP_TRY(hipEventRecord(S.copied, cs));
        int rc2 = chisel_hip_wait_event(dst, S.copied);
        if (rc2) return rc2;
        rc2 = chisel_hip_import_shells_packed(dst, S.recv, bytes);
        if (rc2) return rc2;
        rc2 = chisel_hip_update_meshes_planned(dst);
        if (rc2) return rc2;
        rc2 = chisel_hip_drop_ghost_chunks(dst);
        if (rc2) return rc2;
        rc2 = chisel_hip_record_event(dst, S.imported);
        if (rc2) return rc2;
        S.armed = true;
        return CHISEL_HIP_OK;
    });
    lap(3);
    if (timing)
        fprintf(stderr, "chisel_hip group recompute, host us: dirty lists %.0f | list copies + plan (device) + wait %.0f | exports %.0f | copies + imports + recompute + drop %.0f\n",
                t_phase[0], t_phase[1], t_phase[2], t_phase[3]);
    for (uint64_t v : moved) g->ghost_bytes += v;
    if (!rc) {  // what the next recompute may lay its fixed segments out from (update_meshes_wait_free)
        MS.have_sizes = true;
        MS.need_max_count = MS.need_seg_bytes = MS.need_jobs = MS.need_recv_items = MS.need_send_items = 0;
        for (int r = 0; r < W; r++) {
            MS.need_max_count = std::max(MS.need_max_count, plan[(size_t)r][2]);
            MS.need_jobs = std::max(MS.need_jobs, plan[(size_t)r][0]);
            MS.need_send_items = std::max(MS.need_send_items, plan[(size_t)r][3]);
            int64_t items = 0;
            for (int o = 0; o < W; o++) {
                items += plan[(size_t)r][4 + 2 * W + 2 * o];
                MS.need_seg_bytes = std::max<int64_t>(MS.need_seg_bytes, seg(o, r));
            }
            MS.need_recv_items = std::max(MS.need_recv_items, items);
        }
    }
    return rc;
}


// The same recompute with nothing read in between (kernels_map.h: "the wait-free form"; cvids_amd/sharded.py: _recompute_wait_free is its
// multi-process twin): every (owner, meshing shard) segment has MS.stride bytes to itself -- half as much again as the largest segment of the
// previous recompute --, the owners' export kernels leave their status vectors, every meshing shard copies the slots AND the status vectors of
// all owners and reduces the latter itself (shell_status_max_kernel: the all-reduce), and its import / mesh / drop kernels do nothing if the
// reduced word 0 says that anybody's list, plan or segment did not fit.  The host looks at shard 0's copy of that vector when the group is
// next entered (settle): a recompute that was called off is then made again by update_meshes_blocking, from maps nobody has touched.
// Three fan-outs (an event must have been recorded before another thread makes a stream wait for it), no host wait.
int update_meshes_wait_free(chisel_hip_map *g, bool exact = false) {  // exact: settle()'s second go at a recompute whose slots were too small -- MS.need_* is what THIS recompute needs
    const int W = n_shards(g);
    MeshStages &MS = *static_cast<MeshStages *>(g->mesh_stages_group);
    const bool timing = g_host_timer.on;
    auto t_prev = std::chrono::steady_clock::now();
    double t_phase[4] = {0, 0, 0, 0};
    auto lap = [&](int i) {
        if (!timing) return;
        const auto n = std::chrono::steady_clock::now();
        t_phase[i] += std::chrono::duration<double, std::micro>(n - t_prev).count();
        t_prev = n;
    };
    if (2 * MS.need_max_count > MS.cap) MS.cap = (int)std::min<int64_t>(4 * MS.need_max_count, 1 << 26);
    long long stride = ((long long)MS.need_seg_bytes + (exact ? 0 : MS.need_seg_bytes / 2) + 4096 + 15) / 16 * 16;
    static const bool tiny_slots = getenv("CHISEL_HIP_GROUP_TINY_SLOTS") != nullptr;  // test hook: every other wait-free recompute gets slots of 64 bytes -- called off on the device, made again by settle()
    if (tiny_slots && !exact && (MS.wait_free_recomputes & 1)) stride = 64;
    const long long bytes = stride * W;
    for (int i = 0; i < W; i++) {
        HIP_TRY(hipSetDevice(g->shards[(size_t)i]->device));
        int rc0 = make_stage_events(MS.of[(size_t)i]);
        if (rc0) return rc0;
        if (!MS.copy[(size_t)i]) HIP_TRY(hipStreamCreateWithFlags(&MS.copy[(size_t)i], hipStreamNonBlocking));
    }
    {
        int rc0 = grow_lists(g, MS);
        if (rc0) return rc0;
    }
    // buffers that must grow: after everything that reads the old ones (rare: the slots are laid out with room to spare)
    bool grow = false;
    for (int i = 0; i < W; i++) grow = grow || MS.of[(size_t)i].send_cap < bytes || MS.of[(size_t)i].recv_cap < bytes || !MS.of[(size_t)i].status;
    if (grow || !MS.status_host) {
        for (int i = 0; i < W; i++) {
            int rc0 = chisel_hip_synchronize(g->shards[(size_t)i]);
            if (rc0) return rc0;
        }
        for (int r = 0; r < W; r++) {
            HIP_TRY(hipSetDevice(g->shards[(size_t)r]->device));
            HIP_TRY(hipStreamSynchronize(MS.copy[(size_t)r]));
        }
        for (int i = 0; i < W; i++) {
            MeshStage &S = MS.of[(size_t)i];
            HIP_TRY(hipSetDevice(g->shards[(size_t)i]->device));
            if (S.send_cap < bytes) {
                if (S.send) HIP_TRY(hipFree(S.send));
                S.send = nullptr;
                S.send_cap = std::max<long long>(bytes + bytes / 2, 1 << 20);
                HIP_TRY(hipMalloc(&S.send, (size_t)S.send_cap));
            }
            if (S.recv_cap < bytes) {
                if (S.recv) HIP_TRY(hipFree(S.recv));
                S.recv = nullptr;
                S.recv_cap = std::max<long long>(bytes + bytes / 2, 1 << 20);
                HIP_TRY(hipMalloc(&S.recv, (size_t)S.recv_cap));
            }
            if (!S.status) {
                HIP_TRY(hipMalloc(&S.status, (size_t)(W + 2) * SHELL_STATUS_INTS * sizeof(int)));
                HIP_TRY(hipMemset(S.status, 0, (size_t)(W + 2) * SHELL_STATUS_INTS * sizeof(int)));
            }
        }
        HIP_TRY(hipSetDevice(g->shards[0]->device));
        if (!MS.status_host) HIP_TRY(hipHostMalloc((void **)&MS.status_host, SHELL_STATUS_INTS * sizeof(int), hipHostMallocDefault));
        if (!MS.status_ev) HIP_TRY(hipEventCreateWithFlags(&MS.status_ev, hipEventDisableTiming));
    }
    const int cap = MS.cap, blk = 1 + 4 * cap;
    lap(0);
    // ---- A: the dirty lists
    int rc = run_shards(g, [&](int i) -> int {
        chisel_hip_map *sh = g->shards[(size_t)i];
        MeshStage &S = MS.of[(size_t)i];
        HIP_TRY(hipSetDevice(sh->device));
        int rc2 = chisel_hip_dirty_ids_device(sh, S.gathered + (size_t)i * blk, cap);
        if (rc2) return rc2;
        return chisel_hip_record_event(sh, S.listed);
    });
    if (rc) return rc;
    lap(1);
    // ---- B: the other shards' lists, the plan, the export into fixed slots, this shard's status
    rc = run_shards(g, [&](int r) -> int {
        chisel_hip_map *sh = g->shards[(size_t)r];
        MeshStage &S = MS.of[(size_t)r];
        HIP_TRY(hipSetDevice(sh->device));
        hipStream_t cs = MS.copy[(size_t)r];
        for (int o = 0; o < W; o++) {
            if (o == r) continue;
            HIP_TRY(hipStreamWaitEvent(cs, MS.of[(size_t)o].listed, 0));
            HIP_TRY(hipMemcpyPeerAsync(S.gathered + (size_t)o * blk, sh->device, MS.of[(size_t)o].gathered + (size_t)o * blk, g->shards[(size_t)o]->device, (size_t)blk * sizeof(int), cs));
        }
        HIP_TRY(hipEventRecord(S.gathered_ev, cs));
        // the previous recompute's readers of the send buffer and of this shard's status row: the copies of every meshing shard
        for (int q = 0; q < W; q++)
            if (MS.of[(size_t)q].armed) HIP_TRY(hipStreamWaitEvent(sh->stream, MS.of[(size_t)q].copied, 0));
        int rc2 = chisel_hip_wait_event(sh, S.gathered_ev);
        if (rc2) return rc2;
        rc2 = chisel_hip_shell_plan_queue(sh, S.gathered, W, cap, stride, S.status + SHELL_STATUS_INTS, S.send, (int)MS.need_send_items);
        if (rc2) return rc2;
        return chisel_hip_record_event(sh, S.exported);
    });
    if (rc) return rc;
    lap(2);
    // ---- D: the slots and the status vectors of all owners, their maximum, ghosts, meshes, drop
    rc = run_shards(g, [&](int r) -> int {
        chisel_hip_map *dst = g->shards[(size_t)r];
        MeshStage &S = MS.of[(size_t)r];
        HIP_TRY(hipSetDevice(dst->device));
        hipStream_t cs = MS.copy[(size_t)r];
        if (S.armed) HIP_TRY(hipStreamWaitEvent(cs, S.imported, 0));  // the previous recompute's import and drop have read the buffer
        for (int o = 0; o < W; o++) {
            HIP_TRY(hipStreamWaitEvent(cs, MS.of[(size_t)o].exported, 0));
            if (o != r)  // (a shard owes itself nothing: its own slot only needs its head)
                HIP_TRY(hipMemcpyPeerAsync(S.recv + (size_t)o * stride, dst->device, MS.of[(size_t)o].send + (size_t)r * stride, g->shards[(size_t)o]->device, (size_t)stride, cs));
            else
                HIP_TRY(hipMemcpyAsync(S.recv + (size_t)o * stride, MS.of[(size_t)o].send + (size_t)r * stride, 16, hipMemcpyDeviceToDevice, cs));
            HIP_TRY(hipMemcpyPeerAsync(S.status + (size_t)(2 + o) * SHELL_STATUS_INTS, dst->device, MS.of[(size_t)o].status + SHELL_STATUS_INTS, g->shards[(size_t)o]->device,
                                       SHELL_STATUS_INTS * sizeof(int), cs));
        }
        HIP_TRY(hipEventRecord(S.copied, cs));
        HIP_TRY(hipStreamWaitEvent(dst->stream, S.copied, 0));
        hipLaunchKernelGGL(shell_status_max_kernel, dim3(1), dim3(64), 0, dst->stream, (const int *)(S.status + 2 * SHELL_STATUS_INTS), W, S.status);
        HIP_TRY(hipGetLastError());
        if (r == 0) {
            HIP_TRY(hipMemcpyAsync(MS.status_host, S.status, SHELL_STATUS_INTS * sizeof(int), hipMemcpyDeviceToHost, dst->stream));
            HIP_TRY(hipEventRecord(MS.status_ev, dst->stream));
        }
        int rc2 = chisel_hip_import_shells_fixed(dst, S.recv, stride, S.status, (int)MS.need_jobs, (int)MS.need_recv_items);
        if (rc2) return rc2;
        rc2 = chisel_hip_update_meshes_planned(dst);
        if (rc2) return rc2;
        rc2 = chisel_hip_drop_ghost_chunks(dst);
        if (rc2) return rc2;
        rc2 = chisel_hip_record_event(dst, S.imported);
        if (rc2) return rc2;
        S.armed = true;
        return CHISEL_HIP_OK;
    });
    lap(3);
    if (timing)
        fprintf(stderr, "chisel_hip group recompute (wait-free), host us: buffers %.0f | dirty lists %.0f | list copies + plan + export %.0f | slot copies + import + recompute + drop %.0f\n",
                t_phase[0], t_phase[1], t_phase[2], t_phase[3]);
    MS.stride = stride;
    MS.unsettled = true;  // (also after a failure: whatever was queued is settled before the maps are entered again)
    MS.wait_free_recomputes++;
    return rc;
}

// The host's look at a wait-free recompute, first thing when the group is entered again (chisel_hip.hip: settle): the all-reduced status,
// the shards' commits, and the recompute made again if it was called off.
int settle(chisel_hip_map *g) {
    MeshStages *MSp = static_cast<MeshStages *>(g->mesh_stages_group);
    if (!MSp || !MSp->unsettled) return CHISEL_HIP_OK;
    MeshStages &MS = *MSp;
    MS.unsettled = false;
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipSetDevice(g->shards[0]->device));
    HIP_TRY(hipEventSynchronize(MS.status_ev));
    const auto t1 = std::chrono::steady_clock::now();
    int st[SHELL_STATUS_INTS];
    for (int k = 0; k < SHELL_STATUS_INTS; k++) st[k] = reinterpret_cast<volatile int *>(MS.status_host)[k];
    MS.need_max_count = st[1]; MS.need_seg_bytes = st[2]; MS.need_jobs = st[3]; MS.need_recv_items = st[4]; MS.need_send_items = st[5];
    const int W = n_shards(g);
    int rc = CHISEL_HIP_OK;
    for (int i = 0; i < W && !rc; i++) rc = chisel_hip_shell_commit(g->shards[(size_t)i], st[0] != 0);
    if (g_host_timer.on)
        fprintf(stderr, "chisel_hip group settle, host us: status %.0f | commits %.0f | status %d\n", std::chrono::duration<double, std::micro>(t1 - t0).count(),
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count(), st[0]);
    if (rc) return rc;
    if (st[0] == 0) {
        g->ghost_bytes += (uint64_t)st[6] * (g->cfg.use_color ? 12 : 8) * (uint64_t)W;  // (the shard that received most, times the shards)
        return CHISEL_HIP_OK;
    }
    MS.called_off++;
    if (st[0] == 4 && !MS.retrying) {
        // only the slots were too small, and the status says by how much: the same recompute again (the maps are as they were, so are the
        // plans), still without reading a plan, in slots that hold exactly this; settled right away -- it cannot fail the same way
        MS.retrying = true;
        rc = update_meshes_wait_free(g, true);
        if (!rc) rc = settle(g);
        MS.retrying = false;
        return rc;
    }
    if (st[0] & 1) MS.cap = (int)std::min<int64_t>(2 * (int64_t)st[1], 1 << 26);
    return update_meshes_blocking(g);
}

void forget_recompute(chisel_hip_map *g) {
    if (MeshStages *MS = static_cast<MeshStages *>(g->mesh_stages_group)) MS->unsettled = MS->have_sizes = false;
}

int update_meshes(chisel_hip_map *g, int force) {
    if (!force && (g->update_meshes_calls++ % 10) != 0) return CHISEL_HIP_OK;  // Chisel.cpp:53-58: every 10th call
    const int W = n_shards(g);
    if (W == 1) return chisel_hip_update_meshes(g->shards[0], 1);
    int rc = settle(g);
    if (rc) return rc;
    MeshStages &MS = *static_cast<MeshStages *>(g->mesh_stages_group);
    static const bool blocking_only = getenv("CHISEL_HIP_GROUP_BLOCKING_MESH") != nullptr;  // A/B hook: the form of rounds 5 (two host waits per shard)
    if (MS.have_sizes && !blocking_only) return update_meshes_wait_free(g);
    return update_meshes_blocking(g);
}

int list_meshes(chisel_hip_map *g, int *ids, int64_t max_ids, int64_t *count) {
    std::vector<int> all;
    const int rc = gather_ids(g, chisel_hip_list_meshes, false, all);
    return rc ? rc : emit_ids(all, ids, max_ids, count);
}

// Chisel::SaveAllMeshesToPLY over the shards: every mesh from its owner, ascending chunk id, one file in the single-map format
int save_ply(chisel_hip_map *g, const char *path) {
    std::ofstream stream(path);
    if (!stream) return fail(CHISEL_HIP_ERR_IO, std::string("cannot open ") + path);
    std::vector<int> ids;
    int rc = gather_ids(g, chisel_hip_list_meshes, false, ids);
    if (rc) return rc;
    const size_t n = ids.size() / 3;
    const bool color = g->cfg.use_color != 0;
    std::vector<std::vector<float>> v(n), c(n);
    std::vector<MeshView> views(n);
    size_t numPoints = 0;
    bool any_color = false;
    for (size_t i = 0; i < n; i++) {
        chisel_hip_map *s = owner_map(g, &ids[3 * i]);
        int64_t nv = 0, ng = 0;
        rc = chisel_hip_mesh_size(s, &ids[3 * i], &nv, &ng);
        if (rc) return rc;
        v[i].resize((size_t)nv * 3);
        std::vector<float> nrm((size_t)nv * 3), grids((size_t)ng * 3);
        if (color) c[i].resize((size_t)nv * 3);
        rc = chisel_hip_download_mesh(s, &ids[3 * i], v[i].data(), nrm.data(), color ? c[i].data() : nullptr, grids.data());
        if (rc) return rc;
        views[i].v = v[i].data();
        views[i].c = color ? c[i].data() : nullptr;
        views[i].n_v = (size_t)nv;
        numPoints += (size_t)nv;
        any_color = any_color || (nv && color);
    }
    write_ply(stream, views, numPoints, any_color);
    if (!stream) return fail(CHISEL_HIP_ERR_IO, std::string("write failed: ") + path);
    return CHISEL_HIP_OK;
}

// chisel_hip_save_map of the group: the single-map file (every chunk from its owner, ascending id): a group and a single map
// read each other's files
int save_map(chisel_hip_map *g, const char *path) {
    if (!path) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    std::vector<int> ids;
    int rc = gather_ids(g, chisel_hip_list_chunks, false, ids);
    if (rc) return rc;
    std::ofstream out(path, std::ios::binary);
    if (!out) return fail(CHISEL_HIP_ERR_IO, std::string("cannot open ") + path);
    MapFileHeader h;
    memcpy(h.magic, "CHSLHIP1", 8);
    h.chunk_edge = g->N;
    h.resolution = g->cfg.voxel_resolution;
    h.has_color = g->cfg.use_color ? 1 : 0;
    h.spare = 0;
    h.n_chunks = (int64_t)(ids.size() / 3);
    out.write(reinterpret_cast<const char *>(&h), sizeof(h));
    const size_t V = (size_t)g->V;
    std::vector<float> sdf(V), wgt(V);
    std::vector<uint8_t> col(h.has_color ? 4 * V : 0);
    for (size_t i = 0; i + 2 < ids.size(); i += 3) {
        rc = chisel_hip_download_chunk(owner_map(g, &ids[i]), &ids[i], sdf.data(), wgt.data(), h.has_color ? col.data() : nullptr);
        if (rc) return rc;
        out.write(reinterpret_cast<const char *>(&ids[i]), 3 * sizeof(int));
        out.write(reinterpret_cast<const char *>(sdf.data()), (std::streamsize)(V * sizeof(float)));
        out.write(reinterpret_cast<const char *>(wgt.data()), (std::streamsize)(V * sizeof(float)));
        if (h.has_color) out.write(reinterpret_cast<const char *>(col.data()), (std::streamsize)(4 * V));
    }
    if (!out) return fail(CHISEL_HIP_ERR_IO, std::string("write failed: ") + path);
    return CHISEL_HIP_OK;
}

int chunk_id_at(const chisel_hip_map *g, const float pos[3], int id[3]) {  // ChunkManager::GetIDAt (ChunkManager.h:136-145)
    const float rf = 1.0f / ((float)g->N * g->cfg.voxel_resolution);
    for (int k = 0; k < 3; k++) id[k] = (int)std::floor(pos[k] * rf);
    return owner_of(g, id);
}
int get_sdf(chisel_hip_map *g, const float pos[3], double *dist, int *found) {
    int id[3];
    return chisel_hip_get_sdf(g->shards[chunk_id_at(g, pos, id)], pos, dist, found);
}
// ChunkManager::GetSDFAndGradient (ChunkManager.cpp:449-474): seven GetSDF lookups, each at its own owner
int get_sdf_and_gradient(chisel_hip_map *g, const float pos[3], double *dist, float grad[3], int *found) {
    const float r = g->cfg.voxel_resolution;
    const float posf[3] = {std::floor(pos[0] / r) * r + r / 2.0f, std::floor(pos[1] / r) * r + r / 2.0f, std::floor(pos[2] / r) * r + r / 2.0f};
    double d[7];
    for (int i = 0; i < 7; i++) {
        float q[3] = {posf[0], posf[1], posf[2]};
        if (i >= 1 && i <= 3) q[i - 1] = posf[i - 1] + r;
        if (i >= 4) q[i - 4] = posf[i - 4] - r;
        int ok = 0;
        const int rc = get_sdf(g, q, &d[i], &ok);
        if (rc) return rc;
        if (!ok) {
            if (found) *found = 0;
            return CHISEL_HIP_OK;
        }
    }
    if (dist) *dist = d[0];
    if (grad) {
        const float gx = (float)(d[1] - d[4]), gy = (float)(d[2] - d[5]), gz = (float)(d[3] - d[6]);
        const float z = gx * gx + (gy * gy + gz * gz);  // grad->normalize(): z > 0 ? a / sqrt(z) : a
        if (z > 0.0f) {
            const float s = std::sqrt(z);
            grad[0] = gx / s; grad[1] = gy / s; grad[2] = gz / s;
        } else {
            grad[0] = gx; grad[1] = gy; grad[2] = gz;
        }
    }
    if (found) *found = 1;
    return CHISEL_HIP_OK;
}

int get_counters(chisel_hip_map *g, uint64_t *out, int reset) {
    uint64_t sum[CHISEL_HIP_NUM_COUNTERS] = {0};
    for (size_t i = 0; i < g->shards.size(); i++) {
        uint64_t c[CHISEL_HIP_NUM_COUNTERS];
        const int rc = chisel_hip_get_counters(g->shards[i], c, reset);
        if (rc) return rc;
        for (int k = 0; k < CHISEL_HIP_NUM_COUNTERS; k++) sum[k] = (k == 8) ? std::max(sum[k], c[k]) : sum[k] + c[k];  // [8]: frames (every shard sees all)
    }
    memcpy(out, sum, sizeof(sum));
    return CHISEL_HIP_OK;
}
int get_profile(chisel_hip_map *g, double *ms_total, int64_t *launches, int reset) {
    for (int k = 0; k < CHISEL_HIP_NUM_KERNELS; k++) {
        ms_total[k] = 0.0;
        launches[k] = 0;
    }
    for (chisel_hip_map *s : g->shards) {
        double ms[CHISEL_HIP_NUM_KERNELS];
        int64_t n[CHISEL_HIP_NUM_KERNELS];
        const int rc = chisel_hip_get_profile(s, ms, n, reset);
        if (rc) return rc;
        for (int k = 0; k < CHISEL_HIP_NUM_KERNELS; k++) {
            ms_total[k] += ms[k];
            launches[k] += n[k];
        }
    }
    return CHISEL_HIP_OK;
}

}  // namespace group
}  // namespace
