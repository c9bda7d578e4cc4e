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

// One issuing host thread per shard.  A launch set costs the host 35-120 us to issue (five to six kernel launches, a handful of event
// calls); eight shards issued in turn by the caller's thread cost eight times that per batch, which capped a group below the rate of ONE
// map.  The shard maps share nothing, so every call that fans out over the shards -- integrate, the phases of update_meshes, the id
// listings -- hands shard i's part to worker i and joins.  Workers spin for a while after a job (a stream of frames keeps them hot: a
// condition-variable wake-up costs as much as the job) and then sleep; shard 0's part runs on the calling thread.
// CHISEL_HIP_GROUP_THREADS=0: everything on the calling thread, in turn (A/B, debugging).
struct Pool {
    std::vector<std::thread> workers;           // worker w serves shard w + 1
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
                    __builtin_ia32_pause();
                }
            }
            if (stop.load()) return;
            seen = seq.load(std::memory_order_acquire);
            const int r = job(shard);
            rc[(size_t)shard] = r;
            if (r) err[(size_t)shard] = g_last_error;
            done.fetch_add(1, std::memory_order_release);
        }
    }
};
// a staging buffer of update_meshes (four arrays, grow-only, reused by every recompute) with the events that guard it
struct PairStage {
    float *sdf[2] = {nullptr, nullptr}, *wgt[2] = {nullptr, nullptr};   // [0] on the owner's device, [1] on the meshing shard's (other device only)
    uint8_t *rgbw[2] = {nullptr, nullptr};
    int *found[2] = {nullptr, nullptr};
    long long cap_vox[2] = {0, 0};
    int cap_items[2] = {0, 0};
    hipEvent_t exported = nullptr, copied = nullptr, imported = nullptr;
    bool armed = false;
};
struct MeshStages {
    std::vector<PairStage> out;                // [o]: what owner o packs for ALL requesters, requester by requester (side 0, o's device; `exported`)
    std::vector<PairStage> in;                 // [r]: what meshing shard r installs, owner by owner (side 0, r's device; `copied`, `imported`)
    std::vector<hipStream_t> copy;             // [r]: the copies that assemble in[r] from the owners' out[o]
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
    P->job = fn;
    P->done.store(0, std::memory_order_relaxed);
    P->seq.fetch_add(1);
    if (P->sleepers.load() > 0) {
        std::lock_guard<std::mutex> lk(P->mu);
        P->cv.notify_all();
    }
    const int rc0 = fn(0);
    while (P->done.load(std::memory_order_acquire) < W - 1) __builtin_ia32_pause();
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
        Pool *P = new Pool();
        P->rc.assign((size_t)n, 0);
        P->err.assign((size_t)n, std::string());
        if (const char *e = getenv("CHISEL_HIP_GROUP_THREADS")) P->serial = atoi(e) == 0;
        if (!P->serial)
            for (int i = 1; i < n; i++) P->workers.emplace_back([P, i] { P->loop(i); });
        g->pool = P;
        MeshStages *MS = new MeshStages();
        MS->out.resize((size_t)n);
        MS->in.resize((size_t)n);
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
        for (int side = 0; side < 2; side++)
            for (int i = 0; i < W; i++) {
                PairStage &S = side ? MS->in[(size_t)i] : MS->out[(size_t)i];
                (void)hipSetDevice(g->shards[(size_t)i]->device);
                if (S.sdf[0]) (void)hipFree(S.sdf[0]);
                if (S.wgt[0]) (void)hipFree(S.wgt[0]);
                if (S.rgbw[0]) (void)hipFree(S.rgbw[0]);
                if (S.found[0]) (void)hipFree(S.found[0]);
                if (S.exported) (void)hipEventDestroy(S.exported);
                if (S.copied) (void)hipEventDestroy(S.copied);
                if (S.imported) (void)hipEventDestroy(S.imported);
            }
        for (int r = 0; r < W; r++)
            if (MS->copy[(size_t)r]) {
                (void)hipSetDevice(g->shards[r]->device);
                (void)hipStreamDestroy(MS->copy[(size_t)r]);
            }
        delete MS;
        g->mesh_stages_group = nullptr;
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
    return run_shards(g, [&](int i) -> int {
        chisel_hip_map *s = g->shards[(size_t)i];
        std::vector<chisel_hip_depth_frame> f(frames, frames + n);
        std::vector<chisel_hip_color_frame> c;
        if (colors) c.assign(colors, colors + n);
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
}

// frames in order; consecutive frames of one image size go out KMAX at a time (as integrate_frames cuts them for one map)
int integrate(chisel_hip_map *g, int n, const chisel_hip_depth_frame *frames, const chisel_hip_color_frame *colors) {
    if (n < 0 || (n > 0 && !frames)) return fail(CHISEL_HIP_ERR_INVALID, "bad frame list");
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

// Chisel::UpdateMeshes of the group: cvids_amd/sharded.py: ShardedChisel.UpdateMeshes, with direct calls for the collectives.
// Three fan-outs over the shards, nothing allocated and no stream waited for once the staging buffers have their size:
//   A  every shard lists its meshesToUpdate (one host read per shard, concurrently); the union is the plan's input, the plan of every
//      shard (its jobs; per owner the shells it needs) is pure host arithmetic;
//   B  every OWNER packs, for every shard that asked, the shells into that pair's staging buffer on its own device and records the
//      pair's `exported` event on its stream;
//   C  every MESHING shard waits (on its stream, not on the host) for its owners' events -- with a peer copy on a copy stream in between
//      when the owner lives on another device --, installs the ghosts, recomputes its jobs, drops the ghosts and records `imported`,
//      behind which the owner's next export into the same buffer is ordered.
int ensure_pair(chisel_hip_map *g, PairStage &S, int side, int device, long long vox, int n_items, bool color) {
    if (vox <= S.cap_vox[side] && n_items <= S.cap_items[side]) return CHISEL_HIP_OK;
    HIP_TRY(hipSetDevice(device));
    // (the buffers may still be read by the previous recompute's import / copy: wait for that pair only)
    if (S.armed && S.imported) HIP_TRY(hipEventSynchronize(S.imported));
    if (S.sdf[side]) HIP_TRY(hipFree(S.sdf[side]));
    if (S.wgt[side]) HIP_TRY(hipFree(S.wgt[side]));
    if (S.rgbw[side]) HIP_TRY(hipFree(S.rgbw[side]));
    if (S.found[side]) HIP_TRY(hipFree(S.found[side]));
    S.sdf[side] = S.wgt[side] = nullptr;
    S.rgbw[side] = nullptr;
    S.found[side] = nullptr;
    const long long cv = std::max<long long>(vox + vox / 2, 1 << 16);
    const int ci = std::max(n_items + n_items / 2, 1024);
    HIP_TRY(hipMalloc(&S.sdf[side], (size_t)cv * sizeof(float)));
    HIP_TRY(hipMalloc(&S.wgt[side], (size_t)cv * sizeof(float)));
    if (color) HIP_TRY(hipMalloc(&S.rgbw[side], (size_t)cv * 4));
    HIP_TRY(hipMalloc(&S.found[side], (size_t)ci * sizeof(int)));
    S.cap_vox[side] = cv;
    S.cap_items[side] = ci;
    (void)g;
    return CHISEL_HIP_OK;
}

int update_meshes(chisel_hip_map *g, int force) {
    if (!force && (g->update_meshes_calls++ % 10) != 0) return CHISEL_HIP_OK;  // Chisel.cpp:53-58: every 10th call
    const int W = n_shards(g);
    if (W == 1) return chisel_hip_update_meshes(g->shards[0], 1);
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
    // ---- A: what the shards have updated since the last recompute, as the planner's entries: (x, y, z, 0) per dirty chunk -- the planner
    // expands the 27-neighbourhoods itself (Chisel.h:175-189), on its grid, cheaper than a host set per shard -- and (x, y, z, 1) per
    // id the host holds (neighbourhoods of chunks that were removed while dirty).  Duplicates are the planner's business.
    std::vector<std::vector<int>> part((size_t)W);
    int rc = run_shards(g, [&](int i) -> int {
        chisel_hip_map *sh = g->shards[(size_t)i];
        HIP_TRY(hipSetDevice(sh->device));
        int rc2 = check_mesh_totals(sh);
        if (rc2) return rc2;
        std::vector<int> dirty;
        rc2 = fetch_listed(sh, true, dirty, nullptr);
        if (rc2) return rc2;
        std::vector<int> &e = part[(size_t)i];
        e.reserve(dirty.size() / 3 * 4 + sh->pending_mesh_ids.size() * 4);
        for (size_t j = 0; j + 2 < dirty.size(); j += 3) {
            e.insert(e.end(), dirty.begin() + j, dirty.begin() + j + 3);
            e.push_back(0);
        }
        for (uint64_t key : sh->pending_mesh_ids) {
            int x, y, z;
            unpack_id(key, x, y, z);
            e.push_back(x); e.push_back(y); e.push_back(z); e.push_back(1);
        }
        return CHISEL_HIP_OK;
    });
    if (rc) return rc;
    std::vector<int> entries;
    for (const std::vector<int> &e : part) entries.insert(entries.end(), e.begin(), e.end());
    lap(0);
    const bool color = g->cfg.use_color != 0;
    struct Ask {            // what shard r needs of owner o
        std::vector<int> it4;
        long long vox = 0;
    };
    std::vector<std::vector<int>> jobs((size_t)W);
    std::vector<Ask> ask((size_t)W * W);
    {
        // the plans of all shards in one pass (chisel_hip_mesh_shell_plan_all)
        std::vector<int64_t> jo((size_t)W + 1), io((size_t)W * W + 1);
        rc = chisel_hip_mesh_shell_plan_all(entries.data(), (int64_t)entries.size() / 4, W, g->cfg.shard_block, nullptr, 0, jo.data(), nullptr, 0, io.data());
        if (rc) return rc;
        std::vector<int> jflat((size_t)jo[(size_t)W] * 3 + 1), iflat((size_t)io[(size_t)W * W] * 4 + 1);
        rc = chisel_hip_mesh_shell_plan_all(entries.data(), (int64_t)entries.size() / 4, W, g->cfg.shard_block, jflat.data(), jo[(size_t)W], jo.data(), iflat.data(),
                                            io[(size_t)W * W], io.data());
        if (rc) return rc;
        for (int r = 0; r < W; r++) jobs[(size_t)r].assign(jflat.begin() + 3 * jo[(size_t)r], jflat.begin() + 3 * jo[(size_t)r + 1]);
        for (int p2 = 0; p2 < W * W; p2++) {
            Ask &A = ask[(size_t)p2];
            A.it4.assign(iflat.begin() + 4 * io[(size_t)p2], iflat.begin() + 4 * io[(size_t)p2 + 1]);
            for (size_t k = 3; k < A.it4.size(); k += 4) A.vox += shell_volume(A.it4[k], g->N);
        }
    }
    lap(1);
    // what goes where: owner o packs requester by requester, meshing shard r installs owner by owner
    std::vector<long long> vox_out((size_t)W, 0), vox_in((size_t)W, 0);
    std::vector<int> n_out((size_t)W, 0), n_in((size_t)W, 0);
    std::vector<long long> voff_out((size_t)W * W, 0), voff_in((size_t)W * W, 0);   // [r * W + o]: where pair (r, o) starts in out[o] / in[r] (voxels)
    std::vector<int> noff_out((size_t)W * W, 0), noff_in((size_t)W * W, 0);         // ... (items)
    for (int o = 0; o < W; o++)
        for (int r = 0; r < W; r++) {
            const Ask &A = ask[(size_t)r * W + o];
            voff_out[(size_t)r * W + o] = vox_out[(size_t)o];
            noff_out[(size_t)r * W + o] = n_out[(size_t)o];
            vox_out[(size_t)o] += A.vox;
            n_out[(size_t)o] += (int)(A.it4.size() / 4);
        }
    for (int r = 0; r < W; r++)
        for (int o = 0; o < W; o++) {
            const Ask &A = ask[(size_t)r * W + o];
            voff_in[(size_t)r * W + o] = vox_in[(size_t)r];
            noff_in[(size_t)r * W + o] = n_in[(size_t)r];
            vox_in[(size_t)r] += A.vox;
            n_in[(size_t)r] += (int)(A.it4.size() / 4);
        }
    auto make_events = [](PairStage &S) -> int {
        if (S.exported) return CHISEL_HIP_OK;
        HIP_TRY(hipEventCreateWithFlags(&S.exported, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&S.copied, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&S.imported, hipEventDisableTiming));
        return CHISEL_HIP_OK;
    };
    // ---- B: every owner packs the shells of ALL requesters with one call (2 W export / import calls per recompute, not 2 W^2)
    rc = run_shards(g, [&](int o) -> int {
        chisel_hip_map *src = g->shards[(size_t)o];
        if (n_out[(size_t)o] == 0) return CHISEL_HIP_OK;
        PairStage &S = MS.out[(size_t)o];
        if (vox_out[(size_t)o] > S.cap_vox[0] || n_out[(size_t)o] > S.cap_items[0])  // (about to be reallocated: its readers are the meshing shards' copies)
            for (int r = 0; r < W; r++)
                if (MS.in[(size_t)r].armed) HIP_TRY(hipEventSynchronize(MS.in[(size_t)r].copied));
        int rc2 = ensure_pair(g, S, 0, src->device, vox_out[(size_t)o], n_out[(size_t)o], color);
        if (rc2) return rc2;
        HIP_TRY(hipSetDevice(src->device));
        rc2 = make_events(S);
        if (rc2) return rc2;
        std::vector<int> items;
        items.reserve((size_t)n_out[(size_t)o] * 4);
        for (int r = 0; r < W; r++) {
            const Ask &A = ask[(size_t)r * W + o];
            items.insert(items.end(), A.it4.begin(), A.it4.end());
            // the previous recompute's readers of this buffer: the copies of every meshing shard
            if (MS.in[(size_t)r].armed) HIP_TRY(hipStreamWaitEvent(src->stream, MS.in[(size_t)r].copied, 0));
        }
        rc2 = chisel_hip_export_shells(src, items.data(), n_out[(size_t)o], S.sdf[0], S.wgt[0], S.rgbw[0], S.found[0], 1);
        if (rc2) return rc2;
        return chisel_hip_record_event(src, S.exported);
    });
    if (rc) return rc;
    lap(2);
    // ---- C: every meshing shard assembles its payload from the owners' buffers (copies on its copy stream, behind the owners'
    // events), installs the ghosts with one call, recomputes its jobs, drops the ghosts
    std::vector<uint64_t> moved((size_t)W, 0);
    rc = run_shards(g, [&](int r) -> int {
        chisel_hip_map *dst = g->shards[(size_t)r];
        if (n_in[(size_t)r] > 0) {
            PairStage &S = MS.in[(size_t)r];
            int rc2 = ensure_pair(g, S, 0, dst->device, vox_in[(size_t)r], n_in[(size_t)r], color);
            if (rc2) return rc2;
            HIP_TRY(hipSetDevice(dst->device));
            rc2 = make_events(S);
            if (rc2) return rc2;
            hipStream_t &cs = MS.copy[(size_t)r];
            if (!cs) HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
            if (S.armed) HIP_TRY(hipStreamWaitEvent(cs, S.imported, 0));  // the previous recompute's import has read the buffer
            std::vector<int> items;
            items.reserve((size_t)n_in[(size_t)r] * 4);
            for (int o = 0; o < W; o++) {
                const Ask &A = ask[(size_t)r * W + o];
                if (A.it4.empty()) continue;
                items.insert(items.end(), A.it4.begin(), A.it4.end());
                const PairStage &O = MS.out[(size_t)o];
                chisel_hip_map *src = g->shards[(size_t)o];
                const long long vo = voff_out[(size_t)r * W + o], vi = voff_in[(size_t)r * W + o];
                const int no = noff_out[(size_t)r * W + o], ni = noff_in[(size_t)r * W + o], n = (int)(A.it4.size() / 4);
                HIP_TRY(hipStreamWaitEvent(cs, O.exported, 0));
                HIP_TRY(hipMemcpyPeerAsync(S.sdf[0] + vi, dst->device, O.sdf[0] + vo, src->device, (size_t)A.vox * sizeof(float), cs));
                HIP_TRY(hipMemcpyPeerAsync(S.wgt[0] + vi, dst->device, O.wgt[0] + vo, src->device, (size_t)A.vox * sizeof(float), cs));
                HIP_TRY(hipMemcpyPeerAsync(S.found[0] + ni, dst->device, O.found[0] + no, src->device, (size_t)n * sizeof(int), cs));
                if (color) HIP_TRY(hipMemcpyPeerAsync(S.rgbw[0] + 4 * vi, dst->device, O.rgbw[0] + 4 * vo, src->device, (size_t)A.vox * 4, cs));
                moved[(size_t)r] += (uint64_t)A.vox * (color ? 12 : 8);
            }
            HIP_TRY(hipEventRecord(S.copied, cs));
            rc2 = chisel_hip_wait_event(dst, S.copied);
            if (rc2) return rc2;
            rc2 = chisel_hip_import_ghost_shells(dst, items.data(), n_in[(size_t)r], S.sdf[0], S.wgt[0], S.rgbw[0], S.found[0], 1);
            if (rc2) return rc2;
            rc2 = chisel_hip_record_event(dst, S.imported);
            if (rc2) return rc2;
            S.armed = true;
        }
        int rc2 = chisel_hip_update_meshes_of(dst, jobs[(size_t)r].data(), (int)(jobs[(size_t)r].size() / 3));
        if (rc2) return rc2;
        return chisel_hip_drop_ghost_chunks(dst);
    });
    lap(3);
    if (timing)
        fprintf(stderr, "chisel_hip group recompute, host us: dirty lists %.0f | plan %.0f | exports %.0f | copies + imports + recompute + drop %.0f (%zu entries)\n",
                t_phase[0], t_phase[1], t_phase[2], t_phase[3], entries.size() / 4);
    for (uint64_t v : moved) g->ghost_bytes += v;
    return rc;
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
