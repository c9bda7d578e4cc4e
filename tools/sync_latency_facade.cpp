// What chisel_ros gets, through the C++ facade itself (no Python, no C ABI calls of the caller's own): ONE chisel::DepthImage<float> and ONE
// chisel::ColorImage<uint8_t> allocated once through the facade's classes and refilled every frame (ChiselServer.cpp:268-273,287-292),
// chisel::Chisel::IntegrateDepthScanColor per frame -- synchronous, as Chisel.h:114-213 -- and GetMeshesToUpdate().size() right after
// (ChiselServer.cpp:346).  The facade's images live in page-locked memory (camera/PinholeCamera.h: ImageBuffer -> chisel_hip_host_alloc),
// so this is the "pinned" row of tools/sync_latency_abi.cpp without a line changed on the caller's side.
//   g++ -O2 -std=c++11 -Iinclude -Icvids_amd/open_chisel/include tools/sync_latency_facade.cpp -o tools/sync_latency_facade \
//       -Lcvids_amd -lchisel_hip -Wl,-rpath,'$ORIGIN/../cvids_amd' -Wl,-rpath,/opt/rocm/lib
//   tools/sync_latency_facade [frames]
// Scene: the camera inside a sphere of radius 2.5 m, turning 0.5 degrees and moving 1 cm per frame.
#include <open_chisel/Chisel.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
using namespace chisel;
int main(int argc, char **argv) {
    const int W = 640, H = 480, n = argc > 1 ? atoi(argv[1]) : 140;
    const float fx = 525.0f, fy = 525.0f, cx = 319.5f, cy = 239.5f, R = 2.5f;
    try {
        Chisel map(Eigen::Vector3i(16, 16, 16), 0.01f, true);
        TruncatorPtr trunc(new InverseTruncator(1.0f));
        WeighterPtr weigh(new ConstantWeighter(1.0f));
        ProjectionIntegrator integ(trunc, weigh, 0.05f, true, map.GetChunkManager().GetCentroids());
        PinholeCamera cam;
        Intrinsics K;
        K.SetFx(fx); K.SetFy(fy); K.SetCx(cx); K.SetCy(cy);
        cam.SetIntrinsics(K);
        cam.SetWidth(W); cam.SetHeight(H);
        cam.SetNearPlane(0.05f); cam.SetFarPlane(5.0f);
        std::shared_ptr<DepthImage<float>> depth(new DepthImage<float>(W, H));           // allocated once ...
        std::shared_ptr<ColorImage<uint8_t>> color(new ColorImage<uint8_t>(W, H, 3));
        std::vector<uint8_t> bgr((size_t)W * H * 3);
        for (int v = 0; v < H; v++)
            for (int u = 0; u < W; u++) {
                uint8_t *p = &bgr[((size_t)v * W + u) * 3];
                p[0] = (uint8_t)u; p[1] = (uint8_t)v; p[2] = (uint8_t)(u + v);
            }
        std::vector<float> frame((size_t)W * H);
        std::vector<double> t;
        size_t to_update = 0;
        for (int k = 0; k < n; k++) {
            const float a = 0.5f * k * 3.14159265f / 180.0f, tx = 0.01f * k;
            const float Rm[9] = {cosf(a), 0, sinf(a), 0, 1, 0, -sinf(a), 0, cosf(a)};  // yaw about world y
            for (int v = 0; v < H; v++)
                for (int u = 0; u < W; u++) {
                    const float dx = (u - cx) / fx, dy = (v - cy) / fy;
                    const float wx = Rm[0] * dx + Rm[2], wy = dy, wz = Rm[6] * dx + Rm[8];
                    const float A = wx * wx + wy * wy + wz * wz, B = 2.0f * tx * wx, C = tx * tx - R * R;
                    frame[(size_t)v * W + u] = (-B + sqrtf(B * B - 4.0f * A * C)) / (2.0f * A);
                }
            // ... and refilled per frame, as Conversions.h:107-200 does in the image callbacks (not timed: the caller's work either way)
            memcpy(depth->GetMutableData(), frame.data(), frame.size() * sizeof(float));
            memcpy(color->GetMutableData(), bgr.data(), bgr.size());
            Transform T;
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) T.linear()(r, c) = Rm[3 * r + c];
            T.translation() = Vec3(tx, 0.0f, 0.0f);
            const auto t0 = std::chrono::steady_clock::now();
            map.IntegrateDepthScanColor<float, uint8_t>(integ, depth, T, cam, color, T, cam);
            to_update = map.GetMeshesToUpdate().size();
            const auto t1 = std::chrono::steady_clock::now();
            if (k >= 20) t.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        }
        std::sort(t.begin(), t.end());
        printf("C++ facade, one frame per call, facade-allocated images refilled per frame, GetMeshesToUpdate().size() after every call: p10 %.1f  p50 %.1f  p90 %.1f us per frame (%zu frames, %zu chunks to mesh at the end)\n",
               t[t.size() / 10], t[t.size() / 2], t[t.size() * 9 / 10], t.size(), to_update);
    } catch (const std::exception &e) {
        fprintf(stderr, "sync_latency_facade: %s\n", e.what());
        return 1;
    }
    return 0;
}
