// Packed fp32 (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) against plain v_mul / v_add / v_fma on gfx950: ns per wave64 instruction
// per SIMD at 1, 2, 4, 8 waves per SIMD.  If a packed instruction issues at the rate of a plain one, two voxels' arithmetic costs one slot.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float *out, int iters, float a, float b) {
    f2 x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    const f2 A = {a, a}, B = {b, b};
    float x0s = x0.x, x1s = x1.x, x2s = x2.x, x3s = x3.x, x4s = x4.x, x5s = x5.x, x6s = x6.x, x7s = x7.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#define STEP(x)                                                                                        \
    if (MODE == 0) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(A));                        \
    if (MODE == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(B));                        \
    if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(A), "v"(B));            \
    if (MODE == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x##s) : "v"(a));                         \
    if (MODE == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x##s) : "v"(b));                         \
    if (MODE == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x##s) : "v"(a), "v"(b));
            STEP(x0) STEP(x1) STEP(x2) STEP(x3) STEP(x4) STEP(x5) STEP(x6) STEP(x7)
        }
    }
    f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + x0s + x1s + x2s + x3s + x4s + x5s + x6s + x7s;
}
template <int MODE>
void run(const char *name, float *out) {
    const int iters = 4096;
    for (int w = 1; w <= 8; w *= 2) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k<MODE>, dim3(256 * w), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
            hipEventRecord(b);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
        }
        printf("%-14s waves/SIMD %d: %.3f ms, %.2f ns per wave-instruction per SIMD\n", name, w, ms, ms * 1e6 / ((double)w * iters * 64));
    }
}
int main() {
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float));
    run<0>("v_pk_mul_f32", out); run<1>("v_pk_add_f32", out); run<2>("v_pk_fma_f32", out);
    run<3>("v_mul_f32", out); run<4>("v_add_f32", out); run<5>("v_fma_f32", out);
    return 0;
}
