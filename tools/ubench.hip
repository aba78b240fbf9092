// ubench.hip -- per-instruction throughput probes for gfx950, used to budget the sample kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip ; run on the GPU box.
// Prints lane-ops per cycle per SIMD (32 = full-rate fp32 VALU) assuming the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include <string>

#define ITER 2048
#define CH 8

template <class F> __global__ __launch_bounds__(256) void k_f32(float* out, float seed, F f) {
    float v[CH];
    for (int i = 0; i < CH; i++) v[i] = seed + threadIdx.x * 1e-3f + i;
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < CH; i++) v[i] = f(v[i]);
    float s = 0; for (int i = 0; i < CH; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <class F> __global__ __launch_bounds__(256) void k_f64(double* out, double seed, F f) {
    double v[CH];
    for (int i = 0; i < CH; i++) v[i] = seed + threadIdx.x * 1e-3 + i;
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < CH; i++) v[i] = f(v[i]);
    double s = 0; for (int i = 0; i < CH; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <class F> __global__ __launch_bounds__(256) void k_u32(uint32_t* out, uint32_t seed, F f) {
    uint32_t v[CH];
    for (int i = 0; i < CH; i++) v[i] = seed + threadIdx.x * 977u + i * 131u;
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < CH; i++) v[i] = f(v[i]);
    uint32_t s = 0; for (int i = 0; i < CH; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static double g_clock_ghz = 2.4;
static const int BLOCKS = 256 * 8;

template <class K> double time_ms(K launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
static void report(const char* name, double ms, double ops_per_thread_iter) {
    const double ops = (double)BLOCKS * 256 * ITER * CH * ops_per_thread_iter;
    const double per_s = ops / (ms * 1e-3);
    const double per_cyc_simd = per_s / (256.0 * 4 * g_clock_ghz * 1e9);
    printf("%-28s %8.3f ms  %8.2f Tops/s  %6.2f lanes/clk/SIMD (at %.2f GHz)  cost=%.2f units\n", name, ms, per_s / 1e12, per_cyc_simd, g_clock_ghz, 32.0 / per_cyc_simd);
}

__device__ static inline uint32_t mulmod(uint32_t a, uint32_t b) {
    const unsigned long long p = (unsigned long long)a * b;
    uint32_t r = (uint32_t)(p & 0x7fffffffu) + (uint32_t)(p >> 31);
    r = (r & 0x7fffffffu) + (r >> 31);
    return r;
}
__device__ static inline uint32_t mulmod_min(uint32_t a, uint32_t b) {
    const unsigned long long p = (unsigned long long)a * b;
    uint32_t r = (uint32_t)(p & 0x7fffffffu) + (uint32_t)(p >> 31);
    return min(r, r - 0x7fffffffu);
}
__device__ static inline uint32_t mul16807(uint32_t c) {     // 24-bit multiplier route
    const uint32_t l = c & 0xffffu, h = c >> 16;
    const uint32_t q = __umul24(h, 16807u), t = __umul24(l, 16807u);
    uint32_t r = (q >> 15) + ((q & 0x7fffu) << 16) + t;
    return min(r, r - 0x7fffffffu);
}

int main() {
    float* of; double* od; uint32_t* ou;
    hipMalloc(&of, BLOCKS * 256 * 4); hipMalloc(&od, BLOCKS * 256 * 8); hipMalloc(&ou, BLOCKS * 256 * 4);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    if (clk > 0) g_clock_ghz = clk / 1e6;
    printf("reported max clock %.3f GHz\n", g_clock_ghz);
#define F32(name, ops, body) { auto f = [] __device__(float x) -> float { body; }; double ms = time_ms([&] { hipLaunchKernelGGL(k_f32, dim3(BLOCKS), dim3(256), 0, 0, of, 1.5f, f); }); report(name, ms, ops); }
#define F64(name, ops, body) { auto f = [] __device__(double x) -> double { body; }; double ms = time_ms([&] { hipLaunchKernelGGL(k_f64, dim3(BLOCKS), dim3(256), 0, 0, od, 1.5, f); }); report(name, ms, ops); }
#define U32(name, ops, body) { auto f = [] __device__(uint32_t x) -> uint32_t { body; }; double ms = time_ms([&] { hipLaunchKernelGGL(k_u32, dim3(BLOCKS), dim3(256), 0, 0, ou, 12345u, f); }); report(name, ms, ops); }
    F32("v_fma_f32", 1, return __builtin_fmaf(x, 1.0000001f, 1e-9f));
    F32("v_mul_f32", 1, return x * 1.0000001f);
    F32("v_log_f32", 1, return __builtin_amdgcn_logf(x) + 3.0f);      // +1 add
    F32("v_sqrt_f32", 1, return __builtin_amdgcn_sqrtf(x) + 1.0f);
    F32("v_cos_f32", 1, return __builtin_amdgcn_cosf(x) + 1.5f);
    F32("v_rcp_f32", 1, return __builtin_amdgcn_rcpf(x) + 1.0f);
    F32("v_rsq_f32", 1, return __builtin_amdgcn_rsqf(x) + 1.0f);
    F32("v_exp_f32", 1, return __builtin_amdgcn_exp2f(x) * 0.25f);
    F32("v_fract_f32", 1, return __builtin_amdgcn_fractf(x) + 1.25f);
    F32("v_floor_f32", 1, return __builtin_floorf(x * 1.37f));
    F32("v_cvt_f32_u32(cvt_u32_f32)", 2, return (float)((uint32_t)x + 3u));
    F64("v_fma_f64", 1, return __builtin_fma(x, 1.0000001, 1e-9));
    F64("v_mul_f64", 1, return x * 1.0000001);
    F64("v_add_f64", 1, return x + 1e-9);
    F64("v_rcp_f64", 1, return __builtin_amdgcn_rcp(x) + 1.0);
    F64("v_rsq_f64", 1, return __builtin_amdgcn_rsq(x) + 1.0);
    F64("v_sqrt_f64(hw)", 1, return __builtin_amdgcn_sqrt(x) + 1.0);
    F64("sqrt(double) ieee", 1, return sqrt(x) + 1.0);
    F64("div double ieee", 1, return 1.0 + 3.0 / x);
    F64("log(double) ocml", 1, return log(x) + 3.0);
    F64("cos(double) ocml", 1, return cos(x) + 1.5);
    F64("v_fract_f64", 1, return __builtin_amdgcn_fract(x) + 1.25);
    F64("v_cvt_f64_u32+back", 2, return (double)((uint32_t)x + 3u));
    F64("cvt f64->f32->f64", 2, return (double)((float)x) * 1.0000001);
    U32("v_mul_lo_u32", 1, return x * 2654435761u + 1u);
    U32("v_mul_hi_u32", 1, return __umulhi(x, 2654435761u) + x);
    U32("v_mul_u32_u24", 1, return __umul24(x, 16807u) + 1u);
    U32("v_mad_u64_u32 (64b prod)", 1, unsigned long long p = (unsigned long long)x * 48271u; return (uint32_t)p ^ (uint32_t)(p >> 32));
    U32("mulmod generic (fold2)", 1, return mulmod(x & 0x7fffffffu, 1234567891u));
    U32("mulmod generic (min)", 1, return mulmod_min(x & 0x7fffffffu, 1234567891u));
    U32("mulmod x16807 (mad64)", 1, return mulmod_min(x & 0x7fffffffu, 16807u));
    U32("mulmod x16807 (u24)", 1, return mul16807(x & 0x7fffffffu));
    U32("v_and_b32", 1, return (x & 0x7ffffff1u) + 3u);
    U32("v_alignbit_b32", 1, return __builtin_amdgcn_alignbit(x, x + 7u, 31u));
    U32("v_min_u32+sub", 2, return min(x, x - 0x7fffffffu) + 5u);
    return 0;
}
