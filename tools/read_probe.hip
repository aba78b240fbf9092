// read_probe.hip -- what the memory system gives k_part_hist's access pattern: 0.66 GB of 4-byte records read once, by (a) a linear grid-stride
// sweep, (b) 4032 workgroups that each walk a contiguous slice of 160 KiB, 8 KiB per step and workgroup (the kernel's pattern) with the records
// only summed, (c) the same with the kernel's LDS atomics, (d) 16 / 32 KiB per step, (e) two workgroups' worth of slice per workgroup side by
// side, (f) workgroups of one wavefront.  Times per launch over 20 launches, best and median.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/read_probe tools/read_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
#define SUB 4096
__global__ void k_fill(uint32_t* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (x & (SUB - 1)) | ((8u + (x >> 28)) << 16);
    }
}
// (a) linear: every workgroup's step is the grid's next 8 KiB
template <int Q>
__global__ __launch_bounds__(256) void k_linear(const uint4* __restrict__ in, size_t n4, uint32_t* out) {
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * 256 * Q;
    for (size_t i = (size_t)blockIdx.x * 256 * Q + threadIdx.x; i < n4; i += stride) {
        uint4 v[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) v[q] = in[min(i + 256 * q, n4 - 1)];
#pragma unroll
        for (int q = 0; q < Q; q++) acc += v[q].x ^ v[q].y ^ v[q].z ^ v[q].w;
    }
    if (acc == 0x12345u) out[0] = acc;
}
// (b)-(f) slices: workgroup g walks records [g * len, (g + 1) * len); Q uint4 per lane and step in flight ahead of the step at work
template <int Q, int THREADS, bool ATOMICS>
__global__ __launch_bounds__(THREADS) void k_slice(const uint32_t* __restrict__ part, uint32_t len, uint32_t* out) {
    __shared__ uint32_t row[SUB];
    const int tid = threadIdx.x;
    if (ATOMICS) { for (int i = tid; i < SUB; i += THREADS) row[i] = 0; __syncthreads(); }
    const uint32_t lo = blockIdx.x * len, hi = lo + len;
    const uint4* in = reinterpret_cast<const uint4*>(part + lo) + tid;
    uint4 rec[Q], nxt[Q];
    uint32_t acc = 0;
#pragma unroll
    for (int q = 0; q < Q; q++) rec[q] = in[THREADS * q];
    for (uint32_t b = lo; b < hi; b += 4 * THREADS * Q) {
#pragma unroll
        for (int q = 0; q < Q; q++) nxt[q] = in[THREADS * Q + THREADS * q];
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const uint32_t w[4] = {rec[q].x, rec[q].y, rec[q].z, rec[q].w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (ATOMICS) atomicAdd(&row[w[i] & (SUB - 1)], w[i] >> 16);
                else acc += w[i];
            }
        }
#pragma unroll
        for (int q = 0; q < Q; q++) rec[q] = nxt[q];
        in += THREADS * Q;
    }
    if (ATOMICS) { __syncthreads(); for (int i = tid; i < SUB; i += THREADS) out[(size_t)blockIdx.x * SUB + i] = row[i]; }
    else if (acc == 0x12345u) out[0] = acc;
}
template <typename F>
static void timeit(const char* name, double bytes, F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> t;
    for (int r = 0; r < 22; r++) {
        hipEventRecord(a, 0); launch(); hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (r >= 2) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    const hipError_t e = hipGetLastError();
    printf("%-58s best %7.1f us  median %7.1f us  %5.2f TB/s%s\n", name, t[0] * 1e3, t[t.size() / 2] * 1e3, bytes / (t[t.size() / 2] * 1e-3) / 1e12, e == hipSuccess ? "" : "  ERROR");
}
int main() {
    const uint32_t len = 40960, ns = 4032;
    const size_t n = (size_t)len * ns, slack = 1 << 16;
    uint32_t *part, *out;
    hipMalloc(&part, (n + slack) * 4); hipMalloc(&out, (size_t)ns * 2 * SUB * 4);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, part, n + slack);
    hipDeviceSynchronize();
    const double B = n * 4.0;
    const uint4* p4 = reinterpret_cast<const uint4*>(part);
    timeit("linear, 2560 workgroups, 2 x 16 B per lane and step", B, [&] { hipLaunchKernelGGL(k_linear<2>, dim3(2560), dim3(256), 0, 0, p4, n / 4, out); });
    timeit("linear, 2560 workgroups, 4 x 16 B", B, [&] { hipLaunchKernelGGL(k_linear<4>, dim3(2560), dim3(256), 0, 0, p4, n / 4, out); });
    timeit("linear, 8192 workgroups, 2 x 16 B", B, [&] { hipLaunchKernelGGL(k_linear<2>, dim3(8192), dim3(256), 0, 0, p4, n / 4, out); });
    timeit("slices of 160 KiB, 4032 x 256 threads, 2 x 16 B, summed", B, [&] { hipLaunchKernelGGL((k_slice<2, 256, false>), dim3(ns), dim3(256), 0, 0, part, len, out); });
    timeit("... with the LDS atomics and the 16-KiB row written", B, [&] { hipLaunchKernelGGL((k_slice<2, 256, true>), dim3(ns), dim3(256), 0, 0, part, len, out); });
    timeit("... 4 x 16 B, summed", B, [&] { hipLaunchKernelGGL((k_slice<4, 256, false>), dim3(ns), dim3(256), 0, 0, part, len, out); });
    timeit("... 4 x 16 B, atomics", B, [&] { hipLaunchKernelGGL((k_slice<4, 256, true>), dim3(ns), dim3(256), 0, 0, part, len, out); });
    timeit("... 8 x 16 B, summed", B, [&] { hipLaunchKernelGGL((k_slice<8, 256, false>), dim3(ns), dim3(256), 0, 0, part, len, out); });
    timeit("... 1 x 16 B, summed", B, [&] { hipLaunchKernelGGL((k_slice<1, 256, false>), dim3(ns), dim3(256), 0, 0, part, len, out); });
    timeit("... 1 x 16 B, atomics", B, [&] { hipLaunchKernelGGL((k_slice<1, 256, true>), dim3(ns), dim3(256), 0, 0, part, len, out); });
    timeit("slices of 160 KiB, 4032 x 512 threads, 1 x 16 B, atomics", B, [&] { hipLaunchKernelGGL((k_slice<1, 512, true>), dim3(ns), dim3(512), 0, 0, part, len, out); });
    timeit("slices of 160 KiB, 4032 x 512 threads, 2 x 16 B, atomics", B, [&] { hipLaunchKernelGGL((k_slice<2, 512, true>), dim3(ns), dim3(512), 0, 0, part, len, out); });
    timeit("slices of 160 KiB, 4032 x 1024 threads, 1 x 16 B, atomics", B, [&] { hipLaunchKernelGGL((k_slice<1, 1024, true>), dim3(ns), dim3(1024), 0, 0, part, len, out); });
    timeit("slices of 80 KiB, 8064 x 256 threads, 2 x 16 B, atomics", B, [&] { hipLaunchKernelGGL((k_slice<2, 256, true>), dim3(2 * ns), dim3(256), 0, 0, part, len / 2, out); });
    timeit("slices of 640 KiB, 1008 x 256 threads, 2 x 16 B, atomics", B, [&] { hipLaunchKernelGGL((k_slice<2, 256, true>), dim3(ns / 4), dim3(256), 0, 0, part, len * 4, out); });
    timeit("slices of 640 KiB, 1008 x 1024 threads, 2 x 16 B, atomics", B, [&] { hipLaunchKernelGGL((k_slice<2, 1024, true>), dim3(ns / 4), dim3(1024), 0, 0, part, len * 4, out); });
    return 0;
}
