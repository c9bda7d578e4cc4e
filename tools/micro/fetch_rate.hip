// Is instruction fetch a limiter on gfx950?  Same count of independent VALU instructions per wave, 4-byte (VOP2) against 8-byte
// (VOP3) encodings, unrolled bodies of 64 / 1024 instructions (the small one fits a wave's instruction buffer reuse, the large one streams).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define REP1024(x) REP64(REP8(x)) REP64(REP8(x))
template <int MODE>
__global__ void k(float *out, int iters, float a) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // 1024 x 4-byte
            REP1024(asm volatile("v_add_f32_e32 %0, %4, %0\n v_add_f32_e32 %1, %4, %1\n v_add_f32_e32 %2, %4, %2\n v_add_f32_e32 %3, %4, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));)
        } else if (MODE == 1) {  // 1024 x 8-byte
            REP1024(asm volatile("v_add_f32_e64 %0, %4, %0\n v_add_f32_e64 %1, %4, %1\n v_add_f32_e64 %2, %4, %2\n v_add_f32_e64 %3, %4, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));)
        } else if (MODE == 2) {  // 64 x 4-byte, 16x more iterations
            for (int j = 0; j < 16; j++) { REP64(asm volatile("v_add_f32_e32 %0, %4, %0\n v_add_f32_e32 %1, %4, %1\n v_add_f32_e32 %2, %4, %2\n v_add_f32_e32 %3, %4, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));) }
        } else {  // 64 x 8-byte
            for (int j = 0; j < 16; j++) { REP64(asm volatile("v_add_f32_e64 %0, %4, %0\n v_add_f32_e64 %1, %4, %1\n v_add_f32_e64 %2, %4, %2\n v_add_f32_e64 %3, %4, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));) }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}
template <int MODE>
void run(float *out, const char *name) {
    const int iters = 64;
    for (int w = 1; w <= 8; w *= 2) {
        hipEvent_t a, b;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(k<MODE>, dim3(256 * w), dim3(256), 0, 0, out, iters, 1.0001f);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&ms, a, b);
        }
        const double instr_per_simd = (double)w * iters * 4096;
        printf("%-28s waves/SIMD %d: %.3f ms, %.2f ns per wave-instruction per SIMD\n", name, w, ms, ms * 1e6 / instr_per_simd);
    }
}
int main() {
    float *out;
    (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    run<0>(out, "4-byte, body 4096 instr");
    run<1>(out, "8-byte, body 4096 instr");
    run<2>(out, "4-byte, body 256 instr");
    run<3>(out, "8-byte, body 256 instr");
    return 0;
}
