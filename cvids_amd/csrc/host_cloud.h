// host_cloud.h -- chisel_hip_integrate_pointcloud (included by chisel_hip.hip): Chisel::IntegratePointCloud, Chisel.cpp:107-157.
// Kernels and the design: kernels_cloud.h.
namespace {

// Eigen::Affine3f::inverse() (Transform.h, Mode == Affine): linear().inverse() by cofactors (InverseImpl.h compute_inverse<.., 3>),
// translation = -(inverse_linear * t), each 3-term sum taken as a0 + (a1 + a2).  Row-major 3x4 in and out.
void invert_affine(const float *m, float *r) {
    auto at = [&](int i, int j) { return m[4 * i + j]; };
    auto cof = [&](int i, int j) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        return at(i1, j1) * at(i2, j2) - at(i1, j2) * at(i2, j1);
    };
    const float c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
    const float det = c0 * at(0, 0) + (c1 * at(1, 0) + c2 * at(2, 0));
    const float invdet = 1.0f / det;
    r[0] = c0 * invdet; r[1] = c1 * invdet; r[2] = c2 * invdet;
    r[4] = cof(0, 1) * invdet; r[5] = cof(1, 1) * invdet; r[6] = cof(2, 1) * invdet;
    r[8] = cof(0, 2) * invdet; r[9] = cof(1, 2) * invdet; r[10] = cof(2, 2) * invdet;
    for (int i = 0; i < 3; i++) r[4 * i + 3] = -(r[4 * i] * m[3] + (r[4 * i + 1] * m[7] + r[4 * i + 2] * m[11]));
}

void free_cloud_buffers(chisel_hip_map::CloudBuffers &B) {
    void *ptrs[] = {B.points, B.colors, B.view.rays, B.view.rgb, B.view.tile_prefix, B.view.table_keys /* + ctl, offsets, cursors */, B.view.table_vals,
                    B.view.listed, B.view.pairs, B.view.sorted};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    B = chisel_hip_map::CloudBuffers();
}

int ensure_cloud_buffers(chisel_hip_map *m, int64_t n) {
    chisel_hip_map::CloudBuffers &B = m->cloud;
    CloudView &C = B.view;
    if (!C.table_keys) {
        // one allocation for everything a cloud starts from zero: table keys | control words | per-unit counts (+1) | per-unit cursors
        const size_t units = (size_t)CLOUD_MAX_LISTED * CloudUnits(m->N, 0, cloud_unit_depth(m->N)).count;
        B.zeroed_bytes = (size_t)CLOUD_TABLE_SLOTS * sizeof(uint64_t) + 16 * sizeof(int) + (units + 1) * sizeof(int) + units * sizeof(int);
        char *base = nullptr;
        HIP_TRY(hipMalloc(&base, B.zeroed_bytes));
        C.table_keys = reinterpret_cast<uint64_t *>(base);
        C.ctl = reinterpret_cast<int *>(base + (size_t)CLOUD_TABLE_SLOTS * sizeof(uint64_t));
        C.offsets = C.ctl + 16;
        C.cursors = C.offsets + units + 1;
        HIP_TRY(hipMalloc(&C.table_vals, (size_t)CLOUD_TABLE_SLOTS * sizeof(int)));
        HIP_TRY(hipMalloc(&C.listed, (size_t)CLOUD_MAX_LISTED * sizeof(uint64_t)));
    }
    if (n <= B.capacity) return CHISEL_HIP_OK;
    HIP_TRY(hipStreamSynchronize(m->stream));
    void *ptrs[] = {B.points, B.colors, C.rays, C.rgb, C.tile_prefix, C.pairs, C.sorted};
    for (void *p : ptrs)
        if (p) HIP_TRY(hipFree(p));
    B.points = B.colors = nullptr;
    C.rays = nullptr; C.rgb = nullptr; C.tile_prefix = nullptr; C.pairs = nullptr; C.sorted = nullptr;
    B.capacity = 0;
    int64_t cap = 1 << 16;
    while (cap < n) cap *= 2;
    HIP_TRY(hipMalloc(&B.points, (size_t)cap * 3 * sizeof(float)));
    HIP_TRY(hipMalloc(&B.colors, (size_t)cap * 3 * sizeof(float)));
    HIP_TRY(hipMalloc(&C.rays, (size_t)cap * sizeof(CloudRay)));
    HIP_TRY(hipMalloc(&C.rgb, (size_t)cap * sizeof(unsigned)));
    HIP_TRY(hipMalloc(&C.tile_prefix, (size_t)(cap / CLOUD_TILE + 2) * sizeof(int)));
    HIP_TRY(hipMalloc(&C.pairs, (size_t)cap * CLOUD_PAIRS_PER_POINT * sizeof(int)));
    HIP_TRY(hipMalloc(&C.sorted, (size_t)cap * CLOUD_PAIRS_PER_POINT * sizeof(int)));
    C.pairs_capacity = (int)std::min<int64_t>(cap * CLOUD_PAIRS_PER_POINT, 0x7fffffff);
    B.capacity = cap;
    return CHISEL_HIP_OK;
}

template <int N>
void launch_cloud_integrate(chisel_hip_map *m, const CloudParams &P, const CloudView &C) {
    m->mesh_mark_needed = true;  // (this kernel dirties slots without listing their neighbourhoods: the next recompute runs mesh_mark_kernel)
    if (m->cfg.use_color)
        hipLaunchKernelGGL((cloud_integrate_kernel<N, true>), dim3(CLOUD_GRID), dim3(64 * CloudGeom<N>::WAVES), 0, m->stream, P, m->view, m->view_dev, C);
    else
        hipLaunchKernelGGL((cloud_integrate_kernel<N, false>), dim3(CLOUD_GRID), dim3(64 * CloudGeom<N>::WAVES), 0, m->stream, P, m->view, m->view_dev, C);
}

// what both entry points below need of a cloud: buffers, the points in HBM, the parameters of the per-point kernels
int cloud_setup(chisel_hip_map *m, const chisel_hip_pointcloud *cloud, CloudParams &P, CloudView &C) {
    int rc = ensure_cloud_buffers(m, cloud->n_points);
    if (rc) return rc;
    if (m->input_event) {  // chisel_hip_wait_event: a device cloud produced on another stream is ready behind this event
        HIP_TRY(hipStreamWaitEvent(m->stream, m->input_event, 0));
        m->input_event = nullptr;
    }
    const int n = (int)cloud->n_points;
    C = m->cloud.view;
    if (cloud->on_device) {
        C.points = cloud->points;
        C.colors = cloud->colors;
    } else {
        HIP_TRY(hipMemcpyAsync(m->cloud.points, cloud->points, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, m->stream));
        if (cloud->colors)
            HIP_TRY(hipMemcpyAsync(m->cloud.colors, cloud->colors, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, m->stream));
        C.points = m->cloud.points;
        C.colors = cloud->colors ? m->cloud.colors : nullptr;
    }
    memset(&P, 0, sizeof(P));
    P.ip.trunc_kind = m->integ.truncator_kind;
    P.ip.trunc_param = m->integ.truncator_param;
    P.ip.weight = m->integ.weight;
    P.ip.carving = m->integ.carving_enabled ? 1 : 0;
    P.ip.carving_dist = m->integ.carving_dist;
    P.ip.res = m->cfg.voxel_resolution;
    P.ip.half_res = m->cfg.voxel_resolution * 0.5f;  // ChunkManager.cpp:52
    P.ip.n_shards = m->cfg.n_shards;
    P.ip.shard_rank = m->cfg.shard_rank;
    P.ip.shard_block = m->cfg.shard_block;
    memcpy(P.pose, cloud->pose, sizeof(P.pose));
    invert_affine(cloud->pose, P.inv);
    P.truncation = cloud->truncation;
    P.max_dist = cloud->max_dist;
    P.with_color = (C.colors != nullptr && m->cfg.use_color) ? 1 : 0;  // ProjectionIntegrator.cpp:42
    P.depth_limit = P.with_color ? 5.0f : 2.0f;                         // :131 / :69
    P.n_points = n;
    P.N = m->N;
    P.depth = cloud_unit_depth(m->N);
    // Register axis of cloud_integrate_kernel: the world axis closest to the sensor's y axis (second column of the rotation).
    // Any choice gives the same voxels; this one spreads a batch of consecutive points of an organised cloud over the lanes.
    {
        const float ay[3] = {fabsf(cloud->pose[1]), fabsf(cloud->pose[5]), fabsf(cloud->pose[9])};
        P.jaxis = ay[0] >= ay[1] ? (ay[0] >= ay[2] ? 0 : 2) : (ay[1] >= ay[2] ? 1 : 2);
    }
    if (const char *e = getenv("CHISEL_HIP_CLOUD_AXIS")) P.jaxis = std::max(0, std::min(2, atoi(e)));  // test / tuning hook
    return CHISEL_HIP_OK;
}

}  // namespace

extern "C" int chisel_hip_integrate_pointcloud(chisel_hip_map *m, const chisel_hip_pointcloud *cloud) {
    SETTLE(m);
    if (m && m->is_group) return group::integrate_cloud(m, cloud);
    if (!m || !cloud) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    if (cloud->n_points < 0 || (cloud->n_points > 0 && !cloud->points)) return fail(CHISEL_HIP_ERR_INVALID, "bad point list");
    if (cloud->n_points > (int64_t)(0x7fffffff / (CLOUD_PAIRS_PER_POINT * 2)))
        return fail(CHISEL_HIP_ERR_UNSUPPORTED, "more than 2^26 points in one cloud");
    if (cloud->n_points == 0) return CHISEL_HIP_OK;  // no chunk is listed: Chisel.cpp:112-113
    HIP_TRY(hipSetDevice(m->device));
    int rc = check_mesh_totals(m);  // a recompute in flight reads the voxels as they are
    if (rc) return rc;
    rc = ensure_free_exact(m, CLOUD_MAX_LISTED);  // (a growable pool: a cloud creates at most the chunks it can list)
    if (rc) return rc;
    CloudParams P;
    CloudView C;
    rc = cloud_setup(m, cloud, P, C);
    if (rc) return rc;
    const int n = (int)cloud->n_points;

    ProfScope ps(m, CHISEL_HIP_KERNEL_CLOUD);
    const int tiles = (n + CLOUD_TILE - 1) / CLOUD_TILE;
    const int units_per_chunk = CloudUnits(m->N, P.jaxis, P.depth).count;
    HIP_TRY(hipMemsetAsync(C.table_keys, 0, m->cloud.zeroed_bytes, m->stream));  // table, control words, counts, cursors: one fill
    if (P.with_color) {  // the rank of a point among the accepted ones only selects its colour
        hipLaunchKernelGGL(cloud_tile_count_kernel, dim3(tiles), dim3(CLOUD_TILE), 0, m->stream, P, C);
        hipLaunchKernelGGL(cloud_scan_kernel, dim3(1), dim3(1024), 0, m->stream, C.tile_prefix, (const int *)nullptr, tiles, 1, (int *)nullptr, 0,
                           m->view.error_flag);
    }
    hipLaunchKernelGGL(cloud_prepare_kernel, dim3(tiles), dim3(CLOUD_TILE), 0, m->stream, P, C, m->view);
    hipLaunchKernelGGL(cloud_bin_kernel<false>, dim3(tiles), dim3(CLOUD_TILE), 0, m->stream, P, C, m->view);
    hipLaunchKernelGGL(cloud_scan_kernel, dim3(1), dim3(1024), 0, m->stream, C.offsets, (const int *)C.ctl, CLOUD_MAX_LISTED, units_per_chunk,
                       C.ctl + 1, C.pairs_capacity, m->view.error_flag);
    hipLaunchKernelGGL(cloud_bin_kernel<true>, dim3(tiles), dim3(CLOUD_TILE), 0, m->stream, P, C, m->view);
    hipLaunchKernelGGL(cloud_sort_kernel, dim3(CLOUD_GRID), dim3(256), 0, m->stream, P, C);
    switch (m->N) {
        case 8: launch_cloud_integrate<8>(m, P, C); break;
        case 16: launch_cloud_integrate<16>(m, P, C); break;
        case 32: launch_cloud_integrate<32>(m, P, C); break;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(note_map_mutation(m));
    // the staging buffers (host clouds) and the per-cloud lists are reused by the next cloud: same stream, so no wait here
    return CHISEL_HIP_OK;
}

// ChunkManager::GetChunkIDsIntersecting(cloud, cameraTransform, truncation, maxDist, chunkList) (ChunkManager.cpp:214-257): the chunks
// the segments point -+ truncation along the viewing rays pass through -- the listing step of the call above on its own (the same
// kernel), ids in ascending order (the reference's order is that of an unordered_map).  A shard lists the chunks it owns.
extern "C" int chisel_hip_cloud_candidates(chisel_hip_map *m, const chisel_hip_pointcloud *cloud, int *ids, int64_t max_ids, int64_t *count) {
    SETTLE(m);
    if (m && m->is_group) return fail(CHISEL_HIP_ERR_UNSUPPORTED, "per-shard read-out");
    if (!m || !cloud || !count) return fail(CHISEL_HIP_ERR_INVALID, "null argument");
    if (cloud->n_points < 0 || (cloud->n_points > 0 && !cloud->points)) return fail(CHISEL_HIP_ERR_INVALID, "bad point list");
    if (cloud->n_points > (int64_t)(0x7fffffff / (CLOUD_PAIRS_PER_POINT * 2)))
        return fail(CHISEL_HIP_ERR_UNSUPPORTED, "more than 2^26 points in one cloud");
    *count = 0;
    if (cloud->n_points == 0) return CHISEL_HIP_OK;
    HIP_TRY(hipSetDevice(m->device));
    CloudParams P;
    CloudView C;
    int rc = cloud_setup(m, cloud, P, C);
    if (rc) return rc;
    P.with_color = 0;  // (colours play no part in the listing)
    const int tiles = ((int)cloud->n_points + CLOUD_TILE - 1) / CLOUD_TILE;
    HIP_TRY(hipMemsetAsync(C.table_keys, 0, m->cloud.zeroed_bytes, m->stream));
    hipLaunchKernelGGL(cloud_prepare_kernel, dim3(tiles), dim3(CLOUD_TILE), 0, m->stream, P, C, m->view);
    HIP_TRY(hipGetLastError());
    rc = check_device_error(m);  // waits; too many chunks / a ray out of range are reported here
    if (rc) return rc;
    int n_listed = 0;
    HIP_TRY(hipMemcpy(&n_listed, C.ctl, sizeof(int), hipMemcpyDeviceToHost));
    n_listed = std::min(n_listed, CLOUD_MAX_LISTED);
    std::vector<uint64_t> keys((size_t)n_listed);
    if (n_listed) HIP_TRY(hipMemcpy(keys.data(), C.listed, keys.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
    std::vector<std::array<int, 3>> out(keys.size());
    for (size_t i = 0; i < keys.size(); i++) unpack_id(keys[i], out[i][0], out[i][1], out[i][2]);
    std::sort(out.begin(), out.end());
    *count = (int64_t)out.size();
    if (ids)
        for (int64_t i = 0; i < std::min<int64_t>(max_ids, (int64_t)out.size()); i++) {
            ids[3 * i] = out[(size_t)i][0]; ids[3 * i + 1] = out[(size_t)i][1]; ids[3 * i + 2] = out[(size_t)i][2];
        }
    return CHISEL_HIP_OK;
}
