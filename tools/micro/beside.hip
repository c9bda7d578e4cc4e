// What a short kernel on a second (high-priority) stream takes while a long kernel of single-wave workgroups -- the shape of integrate_kernel:
// 80 vector registers, far more workgroups than the chip holds at once, each alive for ~14 us -- runs on the first.  The 4-agent timeline
// (tools/timeline.sh) shows front-half kernels of 20-40 us lasting as long as the integration kernel beside them; this program varies one
// thing at a time to see what decides that: the short kernel's workgroup size, its register footprint, which of the two was launched first,
// the long kernel's grid (resident at once or 16x the chip), and the priority of the short kernel's stream.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/beside.hip -o tools/micro/beside && tools/micro/beside
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ inline void spin_ticks(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}
// the register footprint is forced by naming the highest register in an asm statement
__global__ __launch_bounds__(64) void long_80(unsigned long long ticks) { asm volatile("v_mov_b32 v79, 0" ::: "v79"); spin_ticks(ticks); }
__global__ __launch_bounds__(64) void long_64(unsigned long long ticks) { asm volatile("v_mov_b32 v63, 0" ::: "v63"); spin_ticks(ticks); }
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void short_24(unsigned long long ticks) { asm volatile("v_mov_b32 v23, 0" ::: "v23"); spin_ticks(ticks); }
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void short_74(unsigned long long ticks) { asm volatile("v_mov_b32 v73, 0" ::: "v73"); spin_ticks(ticks); }

int main() {
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t s_long, s_hi, s_norm;
    CK(hipStreamCreateWithFlags(&s_long, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&s_hi, hipStreamNonBlocking, greatest));
    CK(hipStreamCreateWithPriority(&s_norm, hipStreamNonBlocking, 0));
    hipEvent_t e0, e1, l0, l1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&l0)); CK(hipEventCreate(&l1));
    struct Case { const char *name; int long_regs; int long_grid; int long_us; int short_regs; int short_block; int short_grid; int short_us; bool hi; bool short_first; };
    const Case cases[] = {
        {"alone: short 24 regs, 256 threads", 0, 0, 0, 24, 256, 512, 10, true, false},
        {"alone: short 74 regs, 256 threads", 0, 0, 0, 74, 256, 512, 10, true, false},
        {"long 80 regs x 98304 WGs x 14 us | short 24 regs, 256 thr", 80, 98304, 14, 24, 256, 512, 10, true, false},
        {"long 80 regs x 98304 WGs x 14 us | short 24 regs,  64 thr", 80, 98304, 14, 24, 64, 2048, 10, true, false},
        {"long 80 regs x 98304 WGs x 14 us | short 74 regs, 256 thr", 80, 98304, 14, 74, 256, 512, 10, true, false},
        {"long 80 regs x 98304 WGs x 14 us | short 74 regs,  64 thr", 80, 98304, 14, 74, 64, 2048, 10, true, false},
        {"long 80 regs x 98304 WGs x 14 us | short 74 regs,  64 thr, normal priority", 80, 98304, 14, 74, 64, 2048, 10, false, false},
        {"long 80 regs x 98304 WGs x 14 us | short 74 regs, 256 thr, launched first", 80, 98304, 14, 74, 256, 512, 10, true, true},
        {"long 80 regs x 98304 WGs x 14 us | short 24 regs, 256 thr, launched first", 80, 98304, 14, 24, 256, 512, 10, true, true},
        {"long 80 regs x  6144 WGs x 220 us | short 74 regs, 256 thr", 80, 6144, 220, 74, 256, 512, 10, true, false},
        {"long 80 regs x  5120 WGs x 220 us | short 74 regs, 256 thr", 80, 5120, 220, 74, 256, 512, 10, true, false},
        {"long 80 regs x  5120 WGs x 220 us | short 74 regs,  64 thr", 80, 5120, 220, 74, 64, 2048, 10, true, false},
        {"long 64 regs x 98304 WGs x 14 us | short 24 regs, 256 thr", 64, 98304, 14, 24, 256, 512, 10, true, false},
        {"long 64 regs x 98304 WGs x 14 us | short 74 regs,  64 thr", 64, 98304, 14, 74, 64, 2048, 10, true, false},
    };
    for (const Case &c : cases) {
        std::vector<double> ts, tl, gap;
        for (int it = 0; it < 12; it++) {
            CK(hipDeviceSynchronize());
            hipStream_t ss = c.hi ? s_hi : s_norm;
            auto launch_short = [&]() {
                hipEventRecord(e0, ss);
                const dim3 g(c.short_grid), b(c.short_block);
                const unsigned long long t = (unsigned long long)c.short_us * 100ull;
                if (c.short_regs == 24) { if (c.short_block == 64) hipLaunchKernelGGL(short_24<64>, g, b, 0, ss, t); else hipLaunchKernelGGL(short_24<256>, g, b, 0, ss, t); }
                else { if (c.short_block == 64) hipLaunchKernelGGL(short_74<64>, g, b, 0, ss, t); else hipLaunchKernelGGL(short_74<256>, g, b, 0, ss, t); }
                hipEventRecord(e1, ss);
            };
            auto launch_long = [&]() {
                if (!c.long_grid) return;
                hipEventRecord(l0, s_long);
                if (c.long_regs == 80) hipLaunchKernelGGL(long_80, dim3(c.long_grid), dim3(64), 0, s_long, (unsigned long long)c.long_us * 100ull);
                else hipLaunchKernelGGL(long_64, dim3(c.long_grid), dim3(64), 0, s_long, (unsigned long long)c.long_us * 100ull);
                hipEventRecord(l1, s_long);
            };
            if (c.short_first) { launch_short(); launch_long(); }
            else {
                launch_long();
                std::this_thread::sleep_for(std::chrono::microseconds(40));  // the long kernel is under way
                launch_short();
            }
            CK(hipDeviceSynchronize());
            float ms = 0.0f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 2) ts.push_back(ms * 1e3);
            if (c.long_grid) {
                CK(hipEventElapsedTime(&ms, l0, l1));
                if (it >= 2) tl.push_back(ms * 1e3);
                CK(hipEventElapsedTime(&ms, l0, e1));
                if (it >= 2) gap.push_back(ms * 1e3);
            }
        }
        std::sort(ts.begin(), ts.end()); std::sort(tl.begin(), tl.end()); std::sort(gap.begin(), gap.end());
        printf("%-80s short: p10 %7.1f p50 %7.1f p90 %7.1f us", c.name, ts[ts.size() / 10], ts[ts.size() / 2], ts[ts.size() * 9 / 10]);
        if (c.long_grid) printf(" | long %7.1f us | short ends %7.1f us after the long one began", tl[tl.size() / 2], gap[gap.size() / 2]);
        printf("\n");
    }
    return 0;
}
