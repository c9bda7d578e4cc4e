// SALU issue rate on gfx950: ns per scalar instruction per CU at 1..8 waves per SIMD; also a VALU / SALU mix (2 : 1).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define REP512(x) REP64(REP8(x))
template <int MODE>
__global__ void k(int *out, int iters, int a) {
    int s0 = a, s1 = a + 1, s2 = a + 2, s3 = a + 3;
    float x0 = threadIdx.x, x1 = x0 + 1;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
            REP512(asm volatile("s_add_u32 %0, %0, %4\n s_and_b32 %1, %1, %4\n s_add_u32 %2, %2, %4\n s_or_b32 %3, %3, %4" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(a) : "scc");)
        } else {
            REP512(asm volatile("v_add_f32_e32 %4, %6, %4\n s_add_u32 %0, %0, %7\n v_add_f32_e32 %5, %6, %5\n s_and_b32 %1, %1, %7\n v_add_f32_e32 %4, %6, %4\n s_add_u32 %2, %2, %7\n v_add_f32_e32 %5, %6, %5\n s_or_b32 %3, %3, %7" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(x0), "+v"(x1) : "v"(1.0001f), "s"(a) : "scc");)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s0 + s1 + s2 + s3 + (int)(x0 + x1);
}
template <int MODE>
void run(int *out, const char *name, int per_rep) {
    const int iters = 64;
    for (int w = 1; w <= 8; w *= 2) {
        hipEvent_t a, b;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(k<MODE>, dim3(256 * w), dim3(256), 0, 0, out, iters, 3);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            { hipError_t e = hipGetLastError(); if (e != hipSuccess) { printf("launch error: %s\n", hipGetErrorString(e)); return; } }
            (void)hipEventElapsedTime(&ms, a, b);
        }
        const double per_wave = (double)iters * 512 * per_rep;            // instructions of one wave
        const double per_cu = per_wave * w * 4;                           // waves per CU = 4 w
        printf("%-22s waves/SIMD %d: %.3f ms, %.2f ns per instruction per CU, %.2f ns per instruction per SIMD\n", name, w, ms, ms * 1e6 / per_cu, ms * 1e6 / (per_wave * w));
    }
}
int main() {
    int *out;
    (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(int));
    run<0>(out, "SALU only", 4);
    run<1>(out, "VALU : SALU = 1 : 1", 8);
    return 0;
}
