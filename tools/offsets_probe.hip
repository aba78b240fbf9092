// k_part_offsets alone on synthetic counts: where do its 30 us go?   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I squigulator_amd/csrc tools/offsets_probe.hip -o /tmp/op
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <cstdio>
#include <vector>
#include "k_common.h"
#include "k_events.h"
#include "k_part.h"
__global__ __launch_bounds__(1024) void empty1024(uint32_t* p) { if (p == nullptr) *p = 0; }
__global__ __launch_bounds__(1024) void touch(const uint32_t* __restrict__ pcnt, uint32_t* __restrict__ out, int n_links) {
    const uint32_t* row = pcnt + (size_t)blockIdx.x * n_links;
    uint32_t s = 0;
    for (int i = threadIdx.x; i < n_links; i += 1024) s += row[i];
    if (s == 0xdeadbeef) out[0] = s;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const int n_part = 64, n_links = 8192;
    std::vector<uint32_t> h((size_t)n_part * n_links);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)((i * 2654435761u) >> 27);
    uint32_t *pcnt, *poff, *ptotal; int* wl; uint32_t* big;
    CK(hipMalloc(&pcnt, h.size() * 4)); CK(hipMalloc(&poff, h.size() * 4)); CK(hipMalloc(&ptotal, 4096)); CK(hipMalloc(&wl, 8)); CK(hipMalloc(&big, 1u << 30));
    CK(hipMemcpy(pcnt, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int w[2] = {0, n_links}; CK(hipMemcpy(wl, w, 8, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](const char* what, auto&& launch, bool dirty) {
        float best = 1e9f, sum = 0;
        for (int r = 0; r < 12; r++) {
            if (dirty) (void)hipMemsetAsync(big, r, 1u << 30, s);          // a big kernel in front: caches full of its lines
            (void)hipEventRecord(a, s); launch(); (void)hipEventRecord(b, s); (void)hipStreamSynchronize(s);
            float ms; (void)hipEventElapsedTime(&ms, a, b); if (r >= 2) { best = ms < best ? ms : best; sum += ms; }
        }
        printf("%-48s %s: best %.1f us, mean %.1f us\n", what, dirty ? "behind a 1-GiB fill" : "alone              ", best * 1e3f, sum / 10 * 1e3f);
    };
    for (int dirty = 0; dirty < 2; dirty++) {
        timeit("empty kernel, 64 x 1024 threads", [&] { hipLaunchKernelGGL(empty1024, dim3(64), dim3(1024), 0, s, poff); }, dirty);
        timeit("read the 2 MiB of counts, 64 x 1024 threads", [&] { hipLaunchKernelGGL(touch, dim3(64), dim3(1024), 0, s, pcnt, poff, n_links); }, dirty);
        timeit("k_part_offsets (64, 1) x 1024", [&] { hipLaunchKernelGGL(k_part_offsets, dim3(n_part, 1), dim3(1024), 0, s, pcnt, poff, n_part, n_links, wl, ptotal); }, dirty);
        timeit("k_part_offsets twice", [&] { for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k_part_offsets, dim3(n_part, 1), dim3(1024), 0, s, pcnt, poff, n_part, n_links, wl, ptotal); }, dirty);
    }
    return 0;
}
