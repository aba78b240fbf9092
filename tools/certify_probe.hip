// certify_probe.hip -- exhaustive accuracy sweep of candidate fp32 Box-Muller fast paths against
// the FP64 path, over ALL 2^31-2 LCG states c1 (the normal deviate is a function of c1 alone:
// c2 = 16807*c1 mod M).  Exploration tool; the product's own sweep lives in sqg_hip.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#define M31 2147483647u
__device__ static inline uint32_t mulmod(uint32_t a, uint32_t b) {
    unsigned long long p = (unsigned long long)a * b;
    uint32_t r = (uint32_t)(p & M31) + (uint32_t)(p >> 31);
    r = (r & M31) + (r >> 31);
    return r;
}
__device__ static inline double x_exact(uint32_t c1, uint32_t c2) {
    const double u = (double)c1 / 2147483647.0, t = (2.0 * 3.14159265) * ((double)c2 / 2147483647.0);
    return sqrt(-2.0 * log(u)) * cos(t);
}
// variant 0: unscaled log input
__device__ static inline float x_fast0(uint32_t c1, uint32_t r2) {
    const float lg = __builtin_amdgcn_logf((float)c1);
    const float y = __builtin_fmaf(lg, -1.3862943611198906f, 42.97504449647684f);  // -2ln2*lg + 2 ln M
    const float r = __builtin_amdgcn_sqrtf(y);
    const float cs = __builtin_amdgcn_cosf((float)r2 * 4.656612873077393e-10f);
    return r * cs;
}
// variant 1: scaled input u = c1 * 2^-31 (exact), y = -2ln2*log2(u) - correction for M vs 2^31 ignored? no:
// ln(c1/M) = ln(c1/2^31) + ln(2^31/M) ; ln(2^31/M) = 4.6566e-10 -> add as fma constant
__device__ static inline float x_fast1(uint32_t c1, uint32_t r2) {
    const float uf = (float)c1 * 4.656612873077393e-10f;
    const float lg = __builtin_amdgcn_logf(uf);
    const float y = __builtin_fmaf(lg, -1.3862943611198906f, -9.313225750491594e-10f);
    const float r = __builtin_amdgcn_sqrtf(y);
    const float cs = __builtin_amdgcn_cosf((float)r2 * 4.656612873077393e-10f);
    return r * cs;
}
// variant 2: as 1 but near u~1 use w = M - c1: -ln(1-w/M) ~ via log2 of (1 - w/M) computed as float from w exactly:
// uf = 1 - wf*2^-31 loses nothing when w < 2^24 (wf exact) but 1-x rounds to 2^-24 grid -> same as variant 1. skip.

struct Acc { float maxerr[3][40]; unsigned long long cnt[40]; float relmax[3]; };

__global__ __launch_bounds__(256) void sweep(Acc* acc) {
    __shared__ float smax[3][40];
    for (int i = threadIdx.x; i < 120; i += 256) ((float*)smax)[i] = 0.f;
    __syncthreads();
    const unsigned long long stride = (unsigned long long)gridDim.x * 256;
    for (unsigned long long c = 1 + (unsigned long long)blockIdx.x * 256 + threadIdx.x; c < M31; c += stride) {
        const uint32_t c1 = (uint32_t)c, c2 = mulmod(c1, 16807u);
        const double xe = x_exact(c1, c2);
        // bucket by |x| in units of 0.25 up to 10 -> 40 buckets
        int b = (int)(fabs(xe) * 4.0); if (b > 39) b = 39;
        float e0 = fmaxf(fabsf((float)((double)x_fast0(c1, c2) - xe)), fabsf((float)((double)x_fast0(c1, c2 + M31) - xe)));
        float e1 = fmaxf(fabsf((float)((double)x_fast1(c1, c2) - xe)), fabsf((float)((double)x_fast1(c1, c2 + M31) - xe)));
        // variant 2 = variant 1 restricted to c1 <= M - 2^19
        float e2 = (c1 <= M31 - (1u << 19)) ? e1 : 0.f;
        atomicMax((unsigned int*)&smax[0][b], __float_as_uint(e0));
        atomicMax((unsigned int*)&smax[1][b], __float_as_uint(e1));
        atomicMax((unsigned int*)&smax[2][b], __float_as_uint(e2));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 120; i += 256) atomicMax((unsigned int*)&((float*)acc->maxerr)[i], __float_as_uint(((float*)smax)[i]));
}

// error vs u-region: bucket by log2(M - c1) (closeness to 1) and log2(c1) (closeness to 0)
__global__ __launch_bounds__(256) void sweep_u(float* near1 /*[32] by floor(log2(w))*/, float* near0 /*[32]*/, int variant) {
    const unsigned long long stride = (unsigned long long)gridDim.x * 256;
    for (unsigned long long c = 1 + (unsigned long long)blockIdx.x * 256 + threadIdx.x; c < M31; c += stride) {
        const uint32_t c1 = (uint32_t)c, c2 = mulmod(c1, 16807u);
        const double xe = x_exact(c1, c2);
        float e = variant ? fmaxf(fabsf((float)((double)x_fast1(c1, c2) - xe)), fabsf((float)((double)x_fast1(c1, c2 + M31) - xe)))
                          : fmaxf(fabsf((float)((double)x_fast0(c1, c2) - xe)), fabsf((float)((double)x_fast0(c1, c2 + M31) - xe)));
        const uint32_t w = M31 - c1;
        atomicMax((unsigned int*)&near1[31 - __clz(w)], __float_as_uint(e));
        atomicMax((unsigned int*)&near0[31 - __clz(c1)], __float_as_uint(e));
    }
}

int main() {
    Acc* d; hipMalloc(&d, sizeof(Acc)); hipMemset(d, 0, sizeof(Acc));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(sweep, dim3(256 * 16), dim3(256), 0, 0, d);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    Acc h; hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("sweep of 2^31 states: %.1f ms\n", ms);
    printf("max |x_fast - x_exact| by |x| bucket (width 0.25): v0(unscaled) v1(scaled) v2(scaled, c1<=M-2^19)\n");
    float g[3] = {0, 0, 0};
    for (int i = 0; i < 40; i++) {
        if (h.maxerr[0][i] == 0 && h.maxerr[1][i] == 0) continue;
        printf("  |x| in [%.2f,%.2f): %.3e %.3e %.3e\n", i * 0.25, i * 0.25 + 0.25, h.maxerr[0][i], h.maxerr[1][i], h.maxerr[2][i]);
        for (int v = 0; v < 3; v++) g[v] = fmaxf(g[v], h.maxerr[v][i]);
    }
    printf("global max: %.3e %.3e %.3e\n", g[0], g[1], g[2]);
    float *n1, *n0; hipMalloc(&n1, 128); hipMalloc(&n0, 128);
    for (int variant = 0; variant < 2; variant++) {
        hipMemset(n1, 0, 128); hipMemset(n0, 0, 128);
        hipLaunchKernelGGL(sweep_u, dim3(256 * 16), dim3(256), 0, 0, n1, n0, variant);
        float h1[32], h0[32]; hipMemcpy(h1, n1, 128, hipMemcpyDeviceToHost); hipMemcpy(h0, n0, 128, hipMemcpyDeviceToHost);
        printf("variant %d: max err by floor(log2(M-c1)) [near u=1]:\n ", variant);
        for (int i = 0; i < 31; i++) printf(" %d:%.1e", i, h1[i]);
        printf("\nvariant %d: max err by floor(log2(c1)) [near u=0]:\n ", variant);
        for (int i = 0; i < 31; i++) printf(" %d:%.1e", i, h0[i]);
        printf("\n");
    }
    return 0;
}
