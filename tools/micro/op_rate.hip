// Issue cost of individual VALU instructions on gfx950: ns per wave64 instruction per SIMD at 4 waves per SIMD (8 independent chains
// per wave).  Plain fp32 mul / add / fma run at ~1.0 ns (2.4 cycles); which of the others are full-rate 4-cycle (1.7 ns) operations?
// hipcc --offload-arch=gfx950 -O3 tools/micro/op_rate.hip -o tools/micro/op_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define KERNEL(NAME, ASM, ...)                                                                      \
    __global__ void NAME(float *out, int iters, float a, float b) {                                \
        float x[8];                                                                                \
        unsigned u[8];                                                                             \
        for (int i = 0; i < 8; i++) { x[i] = threadIdx.x + i; u[i] = threadIdx.x * 3 + i; }        \
        unsigned long long m = 0;                                                                  \
        int sg = 0;                                                                                \
        for (int it = 0; it < iters; it++) {                                                       \
            _Pragma("unroll") for (int r = 0; r < 8; r++) {                                        \
                _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : __VA_ARGS__);     \
            }                                                                                      \
        }                                                                                          \
        float s = 0;                                                                               \
        for (int i = 0; i < 8; i++) s += x[i] + u[i];                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)m + sg;                            \
    }
KERNEL(k_mul, "v_mul_f32 %0, %0, %1", "+v"(x[i]) : "v"(a))
KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2", "+v"(x[i]) : "v"(a), "v"(b))
KERNEL(k_mul_s, "v_mul_f32 %0, %1, %0", "+v"(x[i]) : "s"(a))
KERNEL(k_min, "v_min_f32 %0, %0, %1", "+v"(x[i]) : "v"(a))
KERNEL(k_max3, "v_max3_f32 %0, %0, %1, %2", "+v"(x[i]) : "v"(a), "v"(b))
KERNEL(k_mov, "v_mov_b32 %0, %1", "=v"(x[i]) : "v"(a))
KERNEL(k_and, "v_and_b32 %0, %0, %1", "+v"(u[i]) : "v"(0xffffu))
KERNEL(k_or, "v_or_b32 %0, %0, %1", "+v"(u[i]) : "v"(1u))
KERNEL(k_addu, "v_add_u32 %0, %0, %1", "+v"(u[i]) : "v"(3u))
KERNEL(k_lshl, "v_lshlrev_b32 %0, 1, %0", "+v"(u[i]) :)
KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %1", "+v"(u[i]) : "v"(3u))
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2", "+v"(u[i]) : "v"(3u), "v"(5u))
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2", "+v"(u[i]) : "v"(3u), "v"(5u))
KERNEL(k_cvt_fu, "v_cvt_f32_u32 %0, %1", "=v"(x[i]) : "v"(u[i]))
KERNEL(k_cvt_flr, "v_cvt_flr_i32_f32 %0, %1", "=v"(u[i]) : "v"(x[i]))
KERNEL(k_cvt_ub, "v_cvt_f32_ubyte1 %0, %1", "=v"(x[i]) : "v"(u[i]))
KERNEL(k_rcp, "v_rcp_f32 %0, %0", "+v"(x[i]) :)
KERNEL(k_cmp_vcc, "v_cmp_lt_f32 vcc, %0, %1", : "v"(x[i]), "v"(a) : "vcc")
KERNEL(k_cmp_sg, "v_cmp_lt_f32 %0, %1, %2", "=s"(m) : "v"(x[i]), "v"(a))
KERNEL(k_cmp_u, "v_cmp_gt_u32 vcc, %0, %1", : "v"(u[i]), "v"(7u) : "vcc")
KERNEL(k_cnd_vcc, "v_cndmask_b32 %0, %0, %1, vcc", "+v"(x[i]) : "v"(a) : "vcc")
KERNEL(k_cnd_sg, "v_cndmask_b32 %0, %0, %1, %2", "+v"(x[i]) : "v"(a), "s"(m))
KERNEL(k_rfl, "v_readfirstlane_b32 %0, %1", "=s"(sg) : "v"(u[i]))
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 8", "+v"(u[i]) :)
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2", "+v"(u[i]) : "v"(3u), "v"(0x07060504u))
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2", "+v"(x[i]) : "v"(a), "v"(b))
KERNEL(k_sub, "v_sub_f32 %0, %0, %1", "+v"(x[i]) : "v"(a))
KERNEL(k_divfix, "v_div_fixup_f32 %0, %0, %1, %2", "+v"(x[i]) : "v"(a), "v"(b))
KERNEL(k_divfmas, "v_div_fmas_f32 %0, %0, %1, %2", "+v"(x[i]) : "v"(a), "v"(b) : "vcc")
KERNEL(k_floor, "v_floor_f32 %0, %0", "+v"(x[i]) :)
KERNEL(k_cvt_u, "v_cvt_u32_f32 %0, %1", "=v"(u[i]) : "v"(x[i]))
KERNEL(k_cvtpk8, "v_cvt_pk_u8_f32 %0, %1, 1, %0", "+v"(u[i]) : "v"(x[i]))
KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 3, %1", "+v"(u[i]) : "v"(8u))
KERNEL(k_lshlor, "v_lshl_or_b32 %0, %0, 3, %1", "+v"(u[i]) : "v"(8u))
KERNEL(k_min3, "v_min3_f32 %0, %0, %1, %2", "+v"(x[i]) : "v"(a), "v"(b))
KERNEL(k_med3, "v_med3_f32 %0, %0, %1, %2", "+v"(x[i]) : "v"(a), "v"(b))
KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %1", "+v"(u[i]) : "v"(3u))
KERNEL(k_xor, "v_xor_b32 %0, %0, %1", "+v"(u[i]) : "v"(0x7fc00000u))
KERNEL(k_cmpabs, "v_cmp_lt_f32 vcc, |%0|, %1", : "v"(x[i]), "v"(a) : "vcc")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0", "+v"(u[i]) : "v"(0xf0fu))
KERNEL(k_andor, "v_and_or_b32 %0, %0, %1, %2", "+v"(u[i]) : "v"(0xffu), "v"(0x100u))
KERNEL(k_bfi, "v_bfi_b32 %0, %1, %0, %2", "+v"(u[i]) : "v"(0xff00u), "v"(0x100u))
KERNEL(k_fma_lit, "v_fma_f32 %0, %0, %1, 0.5", "+v"(x[i]) : "v"(a))
KERNEL(k_add_s, "v_add_f32 %0, %1, %0", "+v"(x[i]) : "s"(a))
KERNEL(k_divscale, "v_div_scale_f32 %0, vcc, %0, %1, %2", "+v"(x[i]) : "v"(a), "v"(b) : "vcc")
template <typename K>
void run(const char *name, K k, float *out) {
    const int iters = 512, w = 4;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(256 * w), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b);
    }
    printf("%-22s %.2f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name, ms * 1e6 / ((double)w * iters * 64), ms * 1e6 / ((double)w * iters * 64) * 2.4);
}
int main() {
    float *out;
    (void)hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float));
#define R(k) run(#k, k, out);
    R(k_mul) R(k_fma) R(k_fmac) R(k_sub) R(k_mul_s) R(k_min) R(k_max3) R(k_mov) R(k_and) R(k_or) R(k_addu) R(k_lshl) R(k_mul24) R(k_mad24) R(k_add3)
    R(k_bfe) R(k_perm) R(k_cvt_fu) R(k_cvt_flr) R(k_cvt_ub) R(k_rcp) R(k_cmp_vcc) R(k_cmp_sg) R(k_cmp_u) R(k_cnd_vcc) R(k_cnd_sg) R(k_rfl)
    R(k_floor) R(k_cvt_u) R(k_cvtpk8) R(k_lshladd) R(k_lshlor) R(k_min3) R(k_med3) R(k_mullo) R(k_xor) R(k_cmpabs) R(k_bcnt) R(k_andor) R(k_bfi) R(k_fma_lit) R(k_add_s)
    R(k_divscale) R(k_divfmas) R(k_divfix)
    return 0;
}
