// How fast does the hardware start waves and refill a wave slot?  Single-wave (or 4-wave) workgroups that each stay alive for T
// microseconds (s_sleep, no ALU load), a grid of R rounds of what the chip holds at 6 waves per SIMD; every wave records its start and
// end (s_memrealtime, 100 MHz).  Printed: wall time, when the first round's last wave started, the largest number of waves alive at once.
// hipcc --offload-arch=gfx950 -O3 tools/micro/refill_rate.hip -o tools/micro/refill_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
template <int WPB>
__global__ __launch_bounds__(64 * WPB, 6) void spin_kernel(unsigned long long *stamps, int ticks, float *out) {
    float keep[40];
#pragma unroll
    for (int i = 0; i < 40; i++) keep[i] = threadIdx.x + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
    float s = 0;
#pragma unroll
    for (int i = 0; i < 40; i++) s += keep[i] * keep[(i + 7) % 40];
    if (s == 12345.678f) out[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * WPB + (threadIdx.x >> 6);
        stamps[2 * w] = t0;
        stamps[2 * w + 1] = __builtin_amdgcn_s_memrealtime();
    }
}
template <int WPB>
void run(unsigned long long *d_stamps, float *out, int rounds, float t_us) {
    const int resident_waves = 256 * 4 * 6, waves = resident_waves * rounds, blocks = waves / WPB;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(spin_kernel<WPB>, dim3(blocks), dim3(64 * WPB), 0, 0, d_stamps, (int)(t_us * 100), out);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b);
    }
    std::vector<unsigned long long> h(2 * (size_t)waves);
    (void)hipMemcpy(h.data(), d_stamps, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0;
    std::vector<std::pair<unsigned long long, int>> ev;
    std::vector<unsigned long long> starts;
    for (int w = 0; w < waves; w++) {
        t0 = std::min(t0, h[2 * w]); t1 = std::max(t1, h[2 * w + 1]);
        ev.push_back({h[2 * w], 1}); ev.push_back({h[2 * w + 1], -1});
        starts.push_back(h[2 * w]);
    }
    std::sort(ev.begin(), ev.end());
    std::sort(starts.begin(), starts.end());
    int alive = 0, peak = 0;
    for (auto &e : ev) { alive += e.second; peak = std::max(peak, alive); }
    printf("waves/workgroup %d, unit %4.0f us, %d rounds: events %6.1f us, first start -> last end %6.1f us (ideal %5.1f) | wave #%d started at %5.1f us, #%d at %5.1f us | peak alive %d\n",
           WPB, t_us, rounds, ms * 1e3, (t1 - t0) * 0.01, rounds * t_us, resident_waves / 2, (starts[resident_waves / 2 - 1] - t0) * 0.01, resident_waves,
           (starts[resident_waves - 1] - t0) * 0.01, peak);
}
int main() {
    float *out;
    unsigned long long *st;
    (void)hipMalloc(&out, 1024);
    (void)hipMalloc(&st, 2 * 8 * (size_t)6144 * 8);
    for (float t : {5.0f, 12.0f, 25.0f})
        for (int r : {1, 3}) {
            run<1>(st, out, r, t);
            run<4>(st, out, r, t);
        }
    return 0;
}
