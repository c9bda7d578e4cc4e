// The reference's call pattern through the C ABI itself (no Python in the loop): one 640x480 depth + colour frame per call,
// chisel_hip_synchronize after every call -- what chisel::Chisel::IntegrateDepthScanColor of the C++ facade does.
//   tools/sync_latency_abi [frames] [device | pageable | pinned]
// device:   frames resident in HBM (the kernel-side floor);
// pageable: depth AND colour in ordinary host memory, as chisel_ros hands them over (Conversions.h:107-200 fills std::vectors): the
//           library stages them with hipMemcpyAsync from pageable memory;
// pinned:   the same buffers page-locked (hipHostMalloc): depth is read by the pyramid kernel straight over the bus, colour is staged.
//   g++ -O2 -std=c++17 -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/sync_latency_abi.cpp -o tools/sync_latency_abi \
//       -Lcvids_amd -lchisel_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../cvids_amd' -Wl,-rpath,/opt/rocm/lib
// Scene: the camera inside a sphere of radius 2.5 m, turning 0.5 degrees and moving 1 cm per frame.
#include <chisel_hip.h>
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
int main(int argc, char **argv) {
    const int W = 640, H = 480, n = argc > 1 ? atoi(argv[1]) : 140;
    const char *mode = argc > 2 ? argv[2] : "device";
    const bool on_device = !strcmp(mode, "device"), pinned = !strcmp(mode, "pinned");
    const float fx = 525.0f, fy = 525.0f, cx = 319.5f, cy = 239.5f, R = 2.5f;
    chisel_hip_config cfg = {{16, 16, 16}, 0.01f, 1, -1, 0, 1, 0, 0};
    chisel_hip_map *map = nullptr;
    if (chisel_hip_create(&cfg, &map)) { fprintf(stderr, "%s\n", chisel_hip_last_error()); return 1; }
    chisel_hip_integrator in = {CHISEL_HIP_TRUNC_INVERSE, 1.0f, 1.0f, 1, 0.05f};
    chisel_hip_set_integrator(map, &in);
    std::vector<float> depth((size_t)W * H);
    std::vector<uint8_t> bgr((size_t)W * H * 3);
    for (int v = 0; v < H; v++)
        for (int u = 0; u < W; u++) {
            uint8_t *p = &bgr[((size_t)v * W + u) * 3];
            p[0] = (uint8_t)u; p[1] = (uint8_t)v; p[2] = (uint8_t)(u + v);
        }
    uint8_t *d_bgr = nullptr;
    hipMalloc((void **)&d_bgr, bgr.size());
    hipMemcpy(d_bgr, bgr.data(), bgr.size(), hipMemcpyHostToDevice);
    std::vector<float *> d_depth(n);
    std::vector<std::vector<float>> h_depth;   // pageable copies
    uint8_t *h_bgr = bgr.data();
    if (pinned) {
        hipHostMalloc((void **)&h_bgr, bgr.size(), hipHostMallocDefault);
        memcpy(h_bgr, bgr.data(), bgr.size());
    }
    std::vector<chisel_hip_depth_frame> frames(n);
    for (int k = 0; k < n; k++) {
        const float a = 0.5f * k * 3.14159265f / 180.0f, tx = 0.01f * k;
        const float Rm[9] = {cosf(a), 0, sinf(a), 0, 1, 0, -sinf(a), 0, cosf(a)};  // yaw about world y
        for (int v = 0; v < H; v++)
            for (int u = 0; u < W; u++) {
                const float dx = (u - cx) / fx, dy = (v - cy) / fy;  // ray (dx, dy, 1) in the camera frame, z-depth of the sphere hit
                const float wx = Rm[0] * dx + Rm[2], wy = dy, wz = Rm[6] * dx + Rm[8];
                const float A = wx * wx + wy * wy + wz * wz, B = 2.0f * tx * wx, C = tx * tx - R * R;
                depth[(size_t)v * W + u] = (-B + sqrtf(B * B - 4.0f * A * C)) / (2.0f * A);
            }
        if (on_device) {
            hipMalloc((void **)&d_depth[k], depth.size() * sizeof(float));
            hipMemcpy(d_depth[k], depth.data(), depth.size() * sizeof(float), hipMemcpyHostToDevice);
        } else if (pinned) {
            hipHostMalloc((void **)&d_depth[k], depth.size() * sizeof(float), hipHostMallocDefault);
            memcpy(d_depth[k], depth.data(), depth.size() * sizeof(float));
        } else {
            h_depth.push_back(depth);
            d_depth[k] = nullptr;
        }
        chisel_hip_depth_frame f = {d_depth[k], W, H, on_device ? 1 : 0, {Rm[0], Rm[1], Rm[2], tx, Rm[3], Rm[4], Rm[5], 0, Rm[6], Rm[7], Rm[8], 0}, fx, fy, cx, cy, 0.05f, 5.0f};
        frames[k] = f;
    }
    if (!on_device && !pinned)
        for (int k = 0; k < n; k++) frames[k].depth = h_depth[(size_t)k].data();  // (the vector of vectors has stopped growing)
    std::vector<double> t;
    for (int k = 0; k < n; k++) {
        chisel_hip_color_frame c = {on_device ? d_bgr : h_bgr, W, H, 3, on_device ? 1 : 0, {0}, fx, fy, cx, cy};
        memcpy(c.pose, frames[k].pose, sizeof(c.pose));
        const auto t0 = std::chrono::steady_clock::now();
        if (chisel_hip_integrate_depth_color(map, &frames[k], &c) || chisel_hip_synchronize(map)) { fprintf(stderr, "%s\n", chisel_hip_last_error()); return 1; }
        const auto t1 = std::chrono::steady_clock::now();
        if (k >= 20) t.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    std::sort(t.begin(), t.end());
    int64_t chunks = 0;
    chisel_hip_num_chunks(map, &chunks);
    printf("C ABI, one frame per call, wait after every call, %s frames: p10 %.1f  p50 %.1f  p90 %.1f us per frame (%zu frames, %lld chunks)\n", mode, t[t.size() / 10], t[t.size() / 2],
           t[t.size() * 9 / 10], t.size(), (long long)chunks);
    chisel_hip_destroy(map);
    return 0;
}
