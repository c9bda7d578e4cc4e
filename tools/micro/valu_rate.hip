// VALU issue rate on gfx950: cycles per wave64 v_fma_f32 per SIMD at 1..8 waves per SIMD.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
int main() {
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float));
    const int iters = 4096;
    for (int w = 1; w <= 8; w++) {  // w blocks of 256 threads per CU = w waves per SIMD
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k, dim3(256 * w), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep == 2) {
                const double instr_per_simd = (double)w * iters * 64;  // wave-instructions per SIMD
                printf("waves/SIMD %d: %.3f ms, %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", w, ms, ms * 1e6 / instr_per_simd,
                       ms * 1e6 / instr_per_simd * 2.4);
            }
        }
    }
    return 0;
}
