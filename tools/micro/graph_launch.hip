// What a launch set's front half would cost the host as a hipGraph: four dependent kernels with 3 KB of by-value arguments each (the
// size of CullParams / IntegrateParams), (a) launched one by one with a fresh argument block every time, as the library does, against
// (b) one instantiated graph of the same four kernel nodes whose arguments are replaced before every launch
// (hipGraphExecKernelNodeSetParams x 4 + hipGraphLaunch), and (c) the graph launched without touching its arguments (what a graph
// is good at, and what a launch set is not: cameras, masks and ranges change with every batch).  Host time per set, stream never full.
// hipcc --offload-arch=gfx950 -O3 tools/micro/graph_launch.hip -o tools/micro/graph_launch
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Args { int v[768]; };  // 3 KB
__global__ void k(Args a, int *out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = a.v[0] + a.v[767]; }
int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int *d;
    CK(hipMalloc(&d, 64));
    Args a{};
    const int sets = 2000;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto t0, auto t1) { return std::chrono::duration<double, std::micro>(t1 - t0).count(); };
    // (a) four launches per set
    for (int rep = 0; rep < 3; rep++) {
        CK(hipStreamSynchronize(s));
        auto t0 = now();
        for (int i = 0; i < sets; i++) {
            for (int j = 0; j < 4; j++) { a.v[0] = i + j; hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, s, a, d); }
            if ((i & 63) == 63) CK(hipStreamSynchronize(s));  // (the queue never fills: this is issue cost, not back-pressure)
        }
        auto t1 = now();
        CK(hipStreamSynchronize(s));
        if (rep == 2) printf("(a) 4 x hipLaunchKernelGGL, 3 KB of arguments each:           %.2f us per set\n", us(t0, t1) / sets);
    }
    // the graph: four kernel nodes in a chain
    hipGraph_t g;
    CK(hipGraphCreate(&g, 0));
    hipGraphNode_t n[4];
    void *kargs[2] = {&a, &d};
    hipKernelNodeParams p{};
    p.func = (void *)k; p.gridDim = dim3(64); p.blockDim = dim3(64); p.sharedMemBytes = 0; p.kernelParams = kargs; p.extra = nullptr;
    for (int j = 0; j < 4; j++) CK(hipGraphAddKernelNode(&n[j], g, j ? &n[j - 1] : nullptr, j ? 1 : 0, &p));
    hipGraphExec_t ge;
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            CK(hipStreamSynchronize(s));
            auto t0 = now();
            for (int i = 0; i < sets; i++) {
                if (mode == 0)
                    for (int j = 0; j < 4; j++) { a.v[0] = i + j; CK(hipGraphExecKernelNodeSetParams(ge, n[j], &p)); }
                CK(hipGraphLaunch(ge, s));
                if ((i & 63) == 63) CK(hipStreamSynchronize(s));
            }
            auto t1 = now();
            CK(hipStreamSynchronize(s));
            if (rep == 2) printf("%s %.2f us per set\n", mode == 0 ? "(b) 4 x hipGraphExecKernelNodeSetParams + hipGraphLaunch:        " : "(c) hipGraphLaunch alone (arguments frozen):                      ", us(t0, t1) / sets);
        }
    }
    return 0;
}
