// lds_atomic_rate.hip -- how many LDS atomic lanes per clock does a CU serve?  (hipcc --offload-arch=gfx950 -O3 -o tools/bin/lds_atomic_rate tools/lds_atomic_rate.hip)
// W wavefronts per workgroup (one workgroup per CU: 60 KB of LDS each), every lane issues `iters` x 16 atomics on a 4096-entry table
// with pseudo-random addresses; RTN: fetch-add with a used result, else an add without.  Prints lanes per clock per CU at the
// nominal 2.4 GHz and the time per atomic instruction.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <bool RTN, int SPREAD>
__global__ __launch_bounds__(1024) void k(uint32_t* out, int iters) {
    __shared__ uint32_t tab[4096];
    __shared__ uint32_t pad[11000];                               // one workgroup per CU
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = 0;
    if (threadIdx.x == 0) pad[0] = 1;
    __syncthreads();
    uint32_t s = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t a[16];
#pragma unroll
        for (int r = 0; r < 16; r++) { s = s * 1664525u + 1013904223u; a[r] = (s >> 12) & (SPREAD - 1); }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (RTN) acc += __hip_atomic_fetch_add(&tab[a[r]], 13u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(&tab[a[r]], 13u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
        }
    }
    __syncthreads();
    if (acc == 0x12345u || pad[0] == 7) out[threadIdx.x] = acc + tab[threadIdx.x];
}
template <bool RTN, int SPREAD>
static void run(const char* name, int waves) {
    uint32_t* d; CHK(hipMalloc(&d, 4096 * 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 2000, cus = 256;
    hipLaunchKernelGGL((k<RTN, SPREAD>), dim3(cus), dim3(64 * waves), 0, 0, d, 10);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<RTN, SPREAD>), dim3(cus), dim3(64 * waves), 0, 0, d, iters);
    CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double lanes = (double)iters * 16 * 64 * waves;          // per CU
    printf("%-28s %2d waves/CU: %7.3f ms  %6.2f lanes/clk/CU (2.4 GHz)  %6.1f ns per wave-instruction\n", name, waves, ms, lanes / (ms * 1e-3 * 2.4e9), ms * 1e6 / ((double)iters * 16 * waves) );
    CHK(hipFree(d));
}
int main() {
    for (int w : {1, 4, 8, 16}) {
        run<true, 4096>("fetch-add rtn, 4096 addr", w);
        run<false, 4096>("add, 4096 addr", w);
        run<true, 64>("fetch-add rtn, 64 addr", w);
        run<true, 1>("fetch-add rtn, 1 addr", w);
    }
    return 0;
}
