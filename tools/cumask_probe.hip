// cumask_probe.hip -- which CUs a stream created with hipExtStreamCreateWithCUMask runs on: a census kernel (many short workgroups,
// each reports XCC_ID and HW_ID) under a list of mask patterns.  What bit i of the mask stands for is not documented.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/cumask_probe tools/cumask_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <string>
#include <vector>
__global__ __launch_bounds__(64) void k_census(unsigned int* out, int spin) {
    unsigned int hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);       // HW_REG_HW_ID
    unsigned int xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);     // HW_REG_XCC_ID, all bits
    unsigned int acc = threadIdx.x;
    for (int i = 0; i < spin; i++) acc = acc * 1664525u + 1013904223u;
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc ^ (acc == 12345u ? 0x80000000u : 0u); }
}
int main() {
    const int nb = 8192, words = 8;
    unsigned int* d; hipMalloc(&d, nb * 2 * 4);
    std::vector<unsigned int> h(nb * 2);
    struct Pat { std::string name; std::vector<uint32_t> m; };
    std::vector<Pat> pats;
    auto mk = [&](const char* name, auto pred) { Pat p; p.name = name; p.m.assign(words, 0u); for (int i = 0; i < 256; i++) if (pred(i)) p.m[i >> 5] |= 1u << (i & 31); pats.push_back(p); };
    mk("all", [](int) { return true; });
    mk("bits 0..63", [](int i) { return i < 64; });
    mk("bits 0..31", [](int i) { return i < 32; });
    mk("bits 0..7", [](int i) { return i < 8; });
    mk("i%8==0", [](int i) { return i % 8 == 0; });
    mk("i%8<2", [](int i) { return i % 8 < 2; });
    mk("(i/8)%4==0", [](int i) { return (i / 8) % 4 == 0; });
    mk("i%32<8", [](int i) { return i % 32 < 8; });
    mk("bit 0", [](int i) { return i == 0; });
    mk("bit 1", [](int i) { return i == 1; });
    mk("bit 8", [](int i) { return i == 8; });
    mk("bit 9", [](int i) { return i == 9; });
    mk("bit 255", [](int i) { return i == 255; });
    mk("bits 128..255", [](int i) { return i >= 128; });
    for (auto& p : pats) {
        hipStream_t s = nullptr;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, words, p.m.data());
        if (e != hipSuccess) { printf("%-14s: create failed: %s\n", p.name.c_str(), hipGetErrorString(e)); continue; }
        hipMemsetAsync(d, 0xff, nb * 2 * 4, s);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, s);
        hipLaunchKernelGGL(k_census, dim3(nb), dim3(64), 0, s, d, 20000);
        hipEventRecord(b, s);
        hipStreamSynchronize(s);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(h.data(), d, nb * 2 * 4, hipMemcpyDeviceToHost);
        std::map<unsigned, std::set<unsigned>> per_xcc;          // xcc -> {se, sh, cu}
        std::set<unsigned> xraw;
        for (int i = 0; i < nb; i++) { const unsigned hw = h[2 * i], x = h[2 * i + 1]; xraw.insert(x); per_xcc[x & 15u].insert((hw >> 8) & 0xffu); }
        size_t tot = 0; for (auto& kv : per_xcc) tot += kv.second.size();
        printf("%-14s: %.3f ms, %zu CUs:", p.name.c_str(), ms, tot);
        for (auto& kv : per_xcc) printf(" x%u:%zu", kv.first, kv.second.size());
        printf("   raw xcc words:"); int k = 0; for (auto x : xraw) if (k++ < 4) printf(" %08x", x);
        if (tot <= 8) { printf("   cus:"); for (auto& kv : per_xcc) for (auto cu : kv.second) printf(" x%u/%02x", kv.first, cu); }
        printf("\n");
        hipStreamDestroy(s);
    }
    return 0;
}
