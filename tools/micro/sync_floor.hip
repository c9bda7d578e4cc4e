// Floor of the call pattern "three dependent kernels, then the caller waits" (the one-frame path of a caller that synchronises after every
// frame: pyramid -> cull -> integrate -> hipStreamSynchronize): host wall time from the first launch to the return of the wait, for
// kernels that spin for a given time.  hipcc --offload-arch=gfx950 -O3 tools/micro/sync_floor.hip -o tools/micro/sync_floor
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}
int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (unsigned long long us : {0ull, 5ull, 10ull}) {
        for (int n : {1, 3}) {
            std::vector<double> t;
            for (int it = 0; it < 300; it++) {
                const auto t0 = std::chrono::steady_clock::now();
                for (int k = 0; k < n; k++) hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, us * 100ull);
                CK(hipStreamSynchronize(s));
                const auto t1 = std::chrono::steady_clock::now();
                if (it >= 50) t.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
            }
            std::sort(t.begin(), t.end());
            printf("%d kernel(s) of %2llu us each + wait: p10 %.1f  p50 %.1f  p90 %.1f us  (p50 minus the kernels' own time: %.1f)\n", n, us, t[t.size() / 10], t[t.size() / 2],
                   t[t.size() * 9 / 10], t[t.size() / 2] - (double)(n * us));
        }
    }
    return 0;
}
