// ubench2.hip -- how VALU throughput depends on ILP (independent chains per wave) and occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
template <int CH> __global__ __launch_bounds__(64) void k(float* out, float seed) {
    float v[CH];
    for (int i = 0; i < CH; i++) v[i] = seed + threadIdx.x * 1e-3f + i;
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < CH; i++) v[i] = __builtin_fmaf(v[i], 1.0000001f, 1e-9f);
    float s = 0; for (int i = 0; i < CH; i++) s += v[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int CH> __global__ __launch_bounds__(64) void kt(float* out, float seed) {   // transcendental chain
    float v[CH];
    for (int i = 0; i < CH; i++) v[i] = seed + threadIdx.x * 1e-3f + i;
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < CH; i++) v[i] = __builtin_amdgcn_sqrtf(v[i]) + 1.0f;
    float s = 0; for (int i = 0; i < CH; i++) s += v[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <class K> float run(K kern, int waves_per_simd, size_t lds, float* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 4 * waves_per_simd;       // one wave per block; fills exactly waves_per_simd per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, out, 1.5f); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, out, 1.5f); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    float* out; hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
    printf("cycles per wave-instruction per SIMD at 2.4 GHz nominal (lower = better); rows: waves/SIMD, cols: chains/wave\n");
    for (int w : {1, 2, 4, 8}) {
        // limit occupancy with dynamic LDS: 160 KB / (4*w) per block
        size_t lds = (160 * 1024) / (4 * w) - 512; if (w == 8) lds = 0;
        float m1 = run(k<1>, w, lds, out), m2 = run(k<2>, w, lds, out), m4 = run(k<4>, w, lds, out), m8 = run(k<8>, w, lds, out);
        auto cyc = [&](float ms, int ch) { return ms * 1e-3 * 2.4e9 / ((double)ITER * ch * w); };
        printf("fma   w=%d: %.2f %.2f %.2f %.2f\n", w, cyc(m1, 1), cyc(m2, 2), cyc(m4, 4), cyc(m8, 8));
        float t1 = run(kt<1>, w, lds, out), t2 = run(kt<2>, w, lds, out), t4 = run(kt<4>, w, lds, out);
        printf("sqrt+add w=%d: %.2f %.2f %.2f (per pair)\n", w, cyc(t1, 1), cyc(t2, 2), cyc(t4, 4));
    }
    return 0;
}
