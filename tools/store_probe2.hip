// store_probe2: what a wavefront's int16 output costs the vector-memory front end (TA / TCP) by store form (round 5).
// Every wavefront rewrites its own 8-KiB region REPS times (the footprint stays in L2: HBM does not bound the rate), as
//   0: global_store_short, lane-contiguous (128 B per instruction: k_samples_lean's form)
//   1: global_store_dword, 4-byte aligned (256 B per instruction)        2: the same, 2 bytes off
//   3: global_store_dwordx4, 16-byte aligned (1 KiB per instruction)     4: the same, 2 bytes off
//   5: raw buffer_store_dword through a descriptor whose range drops lane 0 (the edge masking the staged stores would use)
//   6: global_store_dwordx2 aligned (512 B per instruction)
// hipcc --offload-arch=gfx950 -O3 tools/store_probe2.hip -o tools/bin/store_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int REGION = 8192;
template <int MODE>
__global__ __launch_bounds__(256) void k_probe(char* base, int reps, uint32_t v) {
    const int lane = threadIdx.x & 63;
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    char* r = base + w * (REGION + 256) + 128;
    for (int it = 0; it < reps; it++) {
        if constexpr (MODE == 0) {
#pragma unroll 8
            for (int s = 0; s < REGION / 128; s++) *reinterpret_cast<uint16_t*>(r + 128 * s + 2 * lane) = (uint16_t)(v + s);
        } else if constexpr (MODE == 1 || MODE == 2) {
            char* q = r + (MODE == 2 ? 2 : 0);
#pragma unroll 8
            for (int s = 0; s < REGION / 256; s++) *reinterpret_cast<uint32_t*>(q + 256 * s + 4 * lane) = v + s;
        } else if constexpr (MODE == 3 || MODE == 4) {
            char* q = r + (MODE == 4 ? 2 : 0);
#pragma unroll 8
            for (int s = 0; s < REGION / 1024; s++) *reinterpret_cast<uint4*>(q + 1024 * s + 16 * lane) = make_uint4(v + s, v, v, v);
        } else if constexpr (MODE == 5) {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(r + 4, 0, REGION - 8, 0x00020000);
#pragma unroll 8
            for (int s = 0; s < REGION / 256; s++) __builtin_amdgcn_raw_buffer_store_b32(v + s, rs, 256 * s + 4 * lane - 4, 0, 0);
        } else {
#pragma unroll 8
            for (int s = 0; s < REGION / 512; s++) *reinterpret_cast<uint2*>(r + 512 * s + 8 * lane) = make_uint2(v + s, v);
        }
        v += 17;
    }
}
template <int MODE> static void run(char* d, int wgs, int reps, const char* what) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k_probe<MODE><<<wgs, 256>>>(d, 2, 1u);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int t = 0; t < 5; t++) {
        CK(hipEventRecord(a)); k_probe<MODE><<<wgs, 256>>>(d, reps, 3u + t); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    const double bytes = (double)wgs * 4 * REGION * reps;
    printf("mode %d  %-44s %8.3f ms  %7.2f TB/s  (%d workgroups x %d reps)\n", MODE, what, best, bytes / best / 1e9, wgs, reps);
}
int main(int argc, char** argv) {
    for (int wgs : {512, 2048, 65536}) {                                 // 2048 x 4 x 8 KiB = 64 MiB (L2 + MALL-resident), 65536: 2 GiB (streams to HBM)
        const int reps = wgs == 512 ? 256 : wgs == 2048 ? 64 : 2;
        char* d; CK(hipMalloc(&d, (size_t)wgs * 4 * (REGION + 256) + 4096));
        run<0>(d, wgs, reps, "store_short, 128 B per instruction");
        run<1>(d, wgs, reps, "store_dword aligned");
        run<2>(d, wgs, reps, "store_dword, 2 bytes off");
        run<3>(d, wgs, reps, "store_dwordx4 aligned");
        run<4>(d, wgs, reps, "store_dwordx4, 2 bytes off");
        run<5>(d, wgs, reps, "buffer_store_dword, range drops lane 0's");
        run<6>(d, wgs, reps, "store_dwordx2 aligned");
        CK(hipFree(d));
    }
    return 0;
}
