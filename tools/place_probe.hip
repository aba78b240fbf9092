// place_probe: is "where hipMalloc put it" a property a short probe can see?  (round 5, tools/runs/r5bd.sh)
// The scatter pass of the 9-mer path runs in one of two modes (830 / 1040 us per 32768-read batch) fixed by its slot's buffers (profiles/r05_summary.md).  Here: N
// allocations of the size of part[] at that batch size (1.3 GB), each timed under three access patterns --
//   lines   : every wavefront writes 64-byte lines at pseudo-random places of the whole buffer (the scatter pass' runs: 2e6 of them in flight)
//   streams : 32768 sequential streams 40 KB apart, one per wavefront, 512 bytes per step (the pass' evrec32 output)
//   sweep   : one grid-stride streaming store over the buffer
// -- median of 7 launches each.  A buffer is kept while the next one is allocated (the allocator cannot hand the same pages out again) and freed afterwards.
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/place_probe tools/place_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void k_lines(uint4* buf, size_t n_lines, int per_wave, uint32_t seed) {
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63, q = lane >> 2, part = lane & 3;          // 16 lines per wavefront instruction, 4 lanes (64 B) each
    uint64_t s = (wave * 0x9E3779B97F4A7C15ull) ^ seed;
    for (int i = 0; i < per_wave; i++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const uint64_t h = (s ^ (s >> 29)) * 0xBF58476D1CE4E5B9ull + (uint64_t)q * 0x94D049BB133111EBull;
        const size_t line = (size_t)((h >> 17) % n_lines);
        buf[line * 4 + part] = make_uint4((uint32_t)h, seed, (uint32_t)i, (uint32_t)lane);
    }
}
__global__ __launch_bounds__(256) void k_streams(uint4* buf, size_t stream_bytes, int steps, uint32_t seed) {
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    char* base = reinterpret_cast<char*>(buf) + wave * stream_bytes;
    for (int i = 0; i < steps; i++)
        if (lane < 32) *reinterpret_cast<uint4*>(base + (size_t)i * 512 + lane * 16) = make_uint4(seed, (uint32_t)i, (uint32_t)lane, 0u);
}
__global__ __launch_bounds__(256) void k_sweep(uint4* buf, size_t n16, uint32_t seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) buf[i] = make_uint4(seed, 1u, 2u, 3u);
}
template <typename F> static float med(F launch) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> t;
    for (int r = 0; r < 8; r++) {
        CK(hipEventRecord(a)); launch(r); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2] * 1e3f;
}
int main(int argc, char** argv) {
    const int n_alloc = argc > 1 ? atoi(argv[1]) : 12;
    const size_t bytes = (size_t)1310 << 20;
    void* filler = nullptr; CK(hipMalloc(&filler, (size_t)3 << 30));             // (something resident, as the genome is)
    uint4* prev = nullptr;
    printf("%-6s %10s %10s %10s   (us: 64-byte lines at random places, 256 MB; 32768 streams, 1.3 GB; one sweep, 1.3 GB)\n", "alloc", "lines", "streams", "sweep");
    for (int a = 0; a < n_alloc; a++) {
        uint4* buf; CK(hipMalloc((void**)&buf, bytes));
        if (prev) CK(hipFree(prev));
        const size_t n_lines = bytes / 64;
        const float t_lines = med([&](int r) { k_lines<<<16384, 256>>>(buf, n_lines, 4, 17u + r); });               // 65536 wavefronts x 4 x 16 lines x 64 B = 268 MB
        const float t_streams = med([&](int r) { k_streams<<<8192, 256>>>(buf, bytes / 32768 / 512 * 512, (int)(bytes / 32768 / 512), 3u + r); });
        const float t_sweep = med([&](int r) { k_sweep<<<4096, 256>>>(buf, bytes / 16, 5u + r); });
        printf("%-6d %10.1f %10.1f %10.1f\n", a, t_lines, t_streams, t_sweep);
        prev = buf;
    }
    return 0;
}
