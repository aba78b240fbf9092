// ubench3.hip -- issue cost of single gfx950 VALU instructions (inline asm, 8 independent chains per wave,
// 8 waves per SIMD): cycles per wave-instruction per SIMD at 2.4 GHz nominal.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2048
#define CH 8
#define DEF32(name, asmstr)                                                                   \
    __global__ __launch_bounds__(64) void name(float* out, float seed) {                       \
        float v[CH];                                                                           \
        for (int i = 0; i < CH; i++) v[i] = seed + threadIdx.x * 1e-3f + i;                    \
        for (int it = 0; it < ITER; it++) {                                                    \
            _Pragma("unroll") for (int i = 0; i < CH; i++) asm volatile(asmstr : "+v"(v[i]));  \
        }                                                                                      \
        float s = 0; for (int i = 0; i < CH; i++) s += v[i];                                   \
        out[blockIdx.x * 64 + threadIdx.x] = s;                                                \
    }
#define DEF64(name, asmstr)                                                                   \
    __global__ __launch_bounds__(64) void name(float* out, float seed) {                       \
        double v[CH];                                                                          \
        for (int i = 0; i < CH; i++) v[i] = seed + threadIdx.x * 1e-3 + i;                     \
        for (int it = 0; it < ITER; it++) {                                                    \
            _Pragma("unroll") for (int i = 0; i < CH; i++) asm volatile(asmstr : "+v"(v[i]));  \
        }                                                                                      \
        double s = 0; for (int i = 0; i < CH; i++) s += v[i];                                  \
        out[blockIdx.x * 64 + threadIdx.x] = (float)s;                                         \
    }
// mixed: 64-bit dst from 32-bit src etc. handled with two register arrays
#define DEFMIX(name, asmstr)                                                                  \
    __global__ __launch_bounds__(64) void name(float* out, float seed) {                       \
        double d[CH]; float f[CH];                                                             \
        for (int i = 0; i < CH; i++) { d[i] = seed + threadIdx.x * 1e-3 + i; f[i] = (float)d[i]; } \
        for (int it = 0; it < ITER; it++) {                                                    \
            _Pragma("unroll") for (int i = 0; i < CH; i++) asm volatile(asmstr : "+v"(d[i]), "+v"(f[i])); \
        }                                                                                      \
        double s = 0; for (int i = 0; i < CH; i++) s += d[i] + f[i];                           \
        out[blockIdx.x * 64 + threadIdx.x] = (float)s;                                         \
    }
DEF32(k_add_f32, "v_add_f32 %0, 1.0, %0")
DEF32(k_fma_f32, "v_fma_f32 %0, %0, %0, %0")
DEF32(k_add_u32, "v_add_u32 %0, 3, %0")
DEF32(k_and_b32, "v_and_b32 %0, 0x7fffffff, %0")
DEF32(k_alignbit, "v_alignbit_b32 %0, %0, %0, 31")
DEF32(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %0")
DEF32(k_min_u32, "v_min_u32 %0, 77, %0")
DEF32(k_mul_lo, "v_mul_lo_u32 %0, %0, %0")
DEF32(k_mul_hi, "v_mul_hi_u32 %0, %0, %0")
DEF32(k_mul_u24, "v_mul_u32_u24 %0, %0, %0")
DEF32(k_mad_u24, "v_mad_u32_u24 %0, %0, %0, %0")
DEF32(k_log, "v_log_f32 %0, %0")
DEF32(k_sqrt, "v_sqrt_f32 %0, %0")
DEF32(k_cos, "v_cos_f32 %0, %0")
DEF32(k_rcp, "v_rcp_f32 %0, %0")
DEF32(k_floor, "v_floor_f32 %0, %0")
DEF32(k_fract, "v_fract_f32 %0, %0")
DEF32(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
DEF32(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
DEF32(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, -1, %0")
DEF32(k_cndmask, "v_cndmask_b32 %0, 0, %0, vcc")
DEF32(k_cmp, "v_cmp_lt_f32 vcc, 0.5, %0")
DEF32(k_perm, "v_perm_b32 %0, %0, %0, %0")
DEF32(k_bfe, "v_bfe_u32 %0, %0, 8, 23")
DEF32(k_add3, "v_add3_u32 %0, %0, %0, %0")
DEF32(k_sdwa_add, "v_add_u32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
DEF32(k_mul_f32, "v_mul_f32 %0, %0, %0")
DEF32(k_sub_f32, "v_sub_f32 %0, 1.0, %0")
DEF32(k_fmac, "v_fmac_f32 %0, %0, %0")
DEF32(k_fmamk, "v_fmamk_f32 %0, %0, 0xbfb17218, %0")
DEF32(k_max_f32, "v_max_f32 %0, %0, %0")
DEF32(k_or_b32, "v_or_b32 %0, 0x3f800000, %0")
DEF32(k_xor_b32, "v_xor_b32 %0, 0x3f800000, %0")
DEF32(k_lshl_b32, "v_lshlrev_b32 %0, 3, %0")
DEF32(k_lshr_b32, "v_lshrrev_b32 %0, 3, %0")
DEF32(k_mov_b32, "v_mov_b32 %0, %0")
DEF32(k_sub_u32, "v_sub_u32 %0, %0, %0")
DEF32(k_subrev_u32, "v_subrev_u32 %0, 77, %0")
DEF32(k_add_co, "v_add_co_u32 %0, vcc, 3, %0")
DEF32(k_max_u32, "v_max_u32 %0, 77, %0")
DEF32(k_min_i32, "v_min_i32 %0, 77, %0")
DEF32(k_and_or, "v_and_or_b32 %0, %0, 15, %0")
DEF32(k_lshl_or, "v_lshl_or_b32 %0, %0, 4, %0")
DEF32(k_fma_e64, "v_fma_f32 %0, |%0|, %0, %0")
DEF32(k_add_f32_e64, "v_add_f32_e64 %0, |%0|, %0")
DEF32(k_cmp_u32, "v_cmp_gt_u32 vcc, 77, %0")
DEF32(k_cmp_e64, "v_cmp_gt_u32_e64 s[10:11], 7, %0")
DEF32(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
DEF32(k_ldexp, "v_ldexp_f32 %0, %0, 3")
DEF32(k_exp, "v_exp_f32 %0, %0")
DEF32(k_rsq, "v_rsq_f32 %0, %0")
DEF32(k_sin, "v_sin_f32 %0, %0")
DEF32(k_add_f32_b, "v_add_f32 %0, 1.0, %0")
DEF64(k_pk_fma, "v_pk_fma_f32 %0, %0, %0, %0")
DEF64(k_pk_mul, "v_pk_mul_f32 %0, %0, %0")
DEF64(k_pk_add, "v_pk_add_f32 %0, %0, %0")
DEF64(k_mul_f64, "v_mul_f64 %0, %0, %0")
DEF64(k_add_f64, "v_add_f64 %0, %0, %0")
DEF64(k_fma_f64, "v_fma_f64 %0, %0, %0, %0")
DEF64(k_fract_f64, "v_fract_f64 %0, %0")
DEF64(k_floor_f64, "v_floor_f64 %0, %0")
DEF64(k_mov_b64, "v_mov_b64 %0, %0")
DEF64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %0")
DEF64(k_lshr_b64, "v_lshrrev_b64 %0, 3, %0")
DEFMIX(k_cvt_f64_u32, "v_cvt_f64_u32 %0, %1")
DEFMIX(k_cvt_f32_f64, "v_cvt_f32_f64 %1, %0")
DEFMIX(k_cvt_u32_f64, "v_cvt_u32_f64 %1, %0")
DEFMIX(k_mad_u64, "v_mad_u64_u32 %0, vcc, %1, %1, 0")
DEFMIX(k_mad_u64b, "v_mad_u64_u32 %0, vcc, %1, %1, %0")

template <class K> double run(K kern, float* out) {
    const int w = 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 4 * w;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, 1.5f); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, 1.5f); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e-3 * 2.4e9 / ((double)ITER * CH * w);
}
#define R(k) printf("%-16s %.2f\n", #k, run(k, out));
int main() {
    float* out; hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
    R(k_add_f32) R(k_fma_f32) R(k_add_u32) R(k_and_b32) R(k_alignbit) R(k_lshl_add) R(k_min_u32) R(k_mul_lo) R(k_mul_hi)
    R(k_mul_u24) R(k_mad_u24) R(k_log) R(k_sqrt) R(k_cos) R(k_rcp) R(k_floor) R(k_fract) R(k_cvt_f32_u32) R(k_cvt_i32_f32)
    R(k_mbcnt) R(k_cndmask) R(k_cmp) R(k_perm) R(k_bfe) R(k_add3) R(k_sdwa_add)
    R(k_pk_fma) R(k_pk_mul) R(k_pk_add) R(k_mul_f64) R(k_add_f64) R(k_fma_f64) R(k_fract_f64) R(k_floor_f64) R(k_mov_b64)
    R(k_mul_f32) R(k_sub_f32) R(k_fmac) R(k_fmamk) R(k_max_f32) R(k_or_b32) R(k_xor_b32) R(k_lshl_b32) R(k_lshr_b32) R(k_mov_b32)
    R(k_sub_u32) R(k_subrev_u32) R(k_add_co) R(k_max_u32) R(k_min_i32) R(k_and_or) R(k_lshl_or) R(k_fma_e64) R(k_add_f32_e64) R(k_cmp_u32) R(k_cmp_e64)
    R(k_cvt_f32_i32) R(k_ldexp) R(k_exp) R(k_rsq) R(k_sin) R(k_add_f32_b) R(k_add_f32) R(k_alignbit)
    R(k_lshl_add_u64) R(k_lshr_b64) R(k_cvt_f64_u32) R(k_cvt_f32_f64) R(k_cvt_u32_f64) R(k_mad_u64) R(k_mad_u64b)
    return 0;
}
