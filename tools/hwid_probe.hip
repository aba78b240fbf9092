// hwid_probe.hip -- where the dispatcher puts the wavefronts of a persistent grid: 4 workgroups per CU, 4 wavefronts each, 38 KiB of
// LDS per workgroup (the shape of k_part_hand_count).  Per wavefront: HW_ID (wave, SIMD, CU, SH, SE) and XCC_ID.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/hwid_probe tools/hwid_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 4) void k_probe(unsigned int* out, int spin) {
    __shared__ unsigned int pad[38880 / 4];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned int hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);       // HW_REG_HW_ID, all 32 bits
    unsigned int xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);      // HW_REG_XCC_ID, bits 0-3
    unsigned int acc = pad[(threadIdx.x * 7) & 255];
    for (int i = 0; i < spin; i++) acc = acc * 1664525u + 1013904223u;            // keep every workgroup resident for a while
    if (lane == 0) { out[(blockIdx.x * 4 + wid) * 2] = hw; out[(blockIdx.x * 4 + wid) * 2 + 1] = xcc | (acc == 12345u ? 0x80000000u : 0u); }
}
int main() {
    const int nb = 1024;
    unsigned int* d; hipMalloc(&d, nb * 4 * 2 * 4);
    hipLaunchKernelGGL(k_probe, dim3(nb), dim3(256), 0, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned int> h(nb * 8);
    hipMemcpy(h.data(), d, nb * 8 * 4, hipMemcpyDeviceToHost);
    // per CU (xcc, se, sh, cu): which blocks, and the SIMD of each of their waves
    std::map<unsigned, std::vector<std::pair<int, unsigned>>> cus;
    for (int b = 0; b < nb; b++) for (int w = 0; w < 4; w++) {
        const unsigned hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 15u;
        const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cus[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back({b * 4 + w, simd});
    }
    printf("%zu distinct CUs\n", cus.size());
    int shown = 0, same_simd_wave = 0, tot = 0;
    for (auto& kv : cus) {
        if (shown++ < 12) {
            printf("cu %05x:", kv.first);
            for (auto& p : kv.second) printf(" b%d.w%d@s%u", p.first / 4, p.first % 4, p.second);
            printf("\n");
        }
        for (auto& p : kv.second) { tot++; if ((unsigned)(p.first % 4) == p.second) same_simd_wave++; }
    }
    printf("waves whose SIMD == wave index in the block: %d of %d\n", same_simd_wave, tot);
    // block ids sharing a CU: differences
    std::map<int, int> diff;
    for (auto& kv : cus) { std::vector<int> bs; for (auto& p : kv.second) if (p.first % 4 == 0) bs.push_back(p.first / 4); for (size_t i = 1; i < bs.size(); i++) diff[bs[i] - bs[0]]++; }
    for (auto& d2 : diff) if (d2.second > 8) printf("block id difference %d: %d times\n", d2.first, d2.second);
    return 0;
}
