// What a stream needs to get from one kernel to the next: dependent kernels back to back, with a hipStreamWaitEvent in between whose event
// (recorded on another stream) fired long before, and with one whose event fires while the first kernel runs -- device timestamps, no profiler.
// hipcc --offload-arch=gfx950 -O3 tools/micro/wait_packet.hip -o tools/micro/wait_packet
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(unsigned long long ticks, unsigned long long *out) {  // 100 MHz ticks; every workgroup spins, block 0 reports
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t0; out[1] = __builtin_amdgcn_s_memrealtime(); }
}
int main() {
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    unsigned long long *d, h[6];
    CK(hipMalloc(&d, 6 * sizeof(unsigned long long)));
    const char *names[3] = {"no wait", "wait, event fired long before", "wait, event fires while the first kernel runs"};
    for (int grid : {1, 2048}) {
        for (int mode = 0; mode < 3; mode++) {
            std::vector<double> gaps;
            for (int it = 0; it < 60; it++) {
                if (mode == 1) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s2, 200ull, d + 4); CK(hipEventRecord(ev, s2)); CK(hipStreamSynchronize(s2)); }
                hipLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, s1, 5000ull, d);   // 50 us
                if (mode == 2) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s2, 2000ull, d + 4); CK(hipEventRecord(ev, s2)); }  // ends 30 us before
                if (mode) CK(hipStreamWaitEvent(s1, ev, 0));
                hipLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, s1, 500ull, d + 2);
                CK(hipStreamSynchronize(s1));
                CK(hipStreamSynchronize(s2));
                CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
                if (it >= 10) gaps.push_back((double)(h[2] - h[1]) / 100.0);
            }
            std::sort(gaps.begin(), gaps.end());
            printf("grid %4d  %-48s end of A -> start of B: p10 %.2f  p50 %.2f  p90 %.2f us\n", grid, names[mode], gaps[5], gaps[25], gaps[45]);
        }
    }
    return 0;
}
