// What does v_cvt_pk_u8_f32 do with fractions, negatives and values above 255?  (truncate or round; saturate?)
// hipcc --offload-arch=gfx950 -O3 tools/micro/probe_cvtpk.hip -o tools/micro/probe_cvtpk
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *in, unsigned *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned r = 0xaabbccddu;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 2, %0" : "+v"(r) : "v"(in[i]));
    out[i] = r;
}
int main() {
    const float vals[] = {0.0f, 0.25f, 0.5f, 0.75f, 1.0f, 1.5f, 2.5f, 3.5f, 254.5f, 254.99f, 255.0f, 255.5f, 256.0f, 300.0f, 1e9f, -0.25f, -0.5f, -0.75f, -1.0f, -3.0f, 127.9375f, 128.0625f, 17.9999f};
    const int n = sizeof(vals) / sizeof(float);
    float *d_in; unsigned *d_out; unsigned h[64];
    (void)hipMalloc(&d_in, n * 4); (void)hipMalloc(&d_out, n * 4);
    (void)hipMemcpy(d_in, vals, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, d_out, n);
    (void)hipMemcpy(h, d_out, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("%12.4f -> 0x%08x (byte2 = %u)\n", vals[i], h[i], (h[i] >> 16) & 0xff);
    return 0;
}
