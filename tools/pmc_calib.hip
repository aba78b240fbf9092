// pmc_calib.hip -- known-byte access patterns for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in THIS code's
// access shapes (tools/pmc_calib.sh runs it under --pmc FETCH_SIZE and --pmc WRITE_SIZE and prints counter / known bytes).
//
//   cal_read16      16 B per lane, lanes contiguous                    (the guide's calibrated case: FETCH_SIZE reports 1/2)
//   cal_read32x     32 B per lane as two dwordx4, lanes 32 B apart     (k_samples_lean's evrec loads: 4 events x 8 B per lane)
//   cal_read8       8 B per lane, lanes contiguous                     (k_samples_lean's dwell load: 4 events x 2 B)
//   cal_read4       4 B per lane, lanes contiguous                     (k_part_hist / k_part_hand_ord: part[] records)
//   cal_slots       the slot array alone (16 B per lane)               (baseline of the two below)
//   cal_gather      slot array + state[slot]: the 4-B gather of k_samples_lean's set-up -- events of a link in chain order, their
//                   slots bucketed by a random 6-bit partition exactly as part[] is laid out (runs of a (link, partition) are
//                   consecutive slots: a 256-event item touches 64 partitions x ~4 consecutive words)
//   cal_gather_xcd  the same with the workgroup -> item map that keeps consecutive items on one XCD (blockIdx % 8 = XCD)
//   cal_rgather8    8 B per lane from a 2-MiB table at random places, 16 per lane (k_samples_lean's pore-table look-up)
//   cal_write2      int16 per lane, lanes contiguous (128 B per wave store) (the signal stores of k_samples_lean)
//   cal_write4      4 B per lane, lanes contiguous                     (k_part_hand_ord: state[])
//   cal_write16     16 B per lane                                      (k_store_probe)
//   cal_write_lines 64-B lines at scattered places, 16 lanes per line  (k_part_events<SCATTER>: part[] line by line)
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/pmc_calib tools/pmc_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void cal_read16(const uint4* __restrict__ a, size_t n, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint4 v = a[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void cal_read32x(const uint4* __restrict__ a, size_t n_lanes, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_lanes; i += (size_t)gridDim.x * 256) {
        const uint4 v = a[2 * i], w = a[2 * i + 1];
        acc ^= v.x ^ v.y ^ v.z ^ v.w ^ w.x ^ w.y ^ w.z ^ w.w;
    }
    if (acc == 0x12345u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void cal_read8(const uint2* __restrict__ a, size_t n, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint2 v = a[i]; acc ^= v.x ^ v.y; }
    if (acc == 0x12345u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void cal_read4(const uint32_t* __restrict__ a, size_t n, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= a[i];
    if (acc == 0x12345u) sink[0] = acc;
}
// one wavefront per item of 256 events, 4 consecutive events per lane; MODE 0: slots only; 1: + gather; XCD: item map
template <int MODE, bool XCD>
__global__ __launch_bounds__(256) void cal_gather(const uint4* __restrict__ slots, const uint32_t* __restrict__ state, const unsigned n_items, uint32_t* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned wg = blockIdx.x;
    if (XCD) { const unsigned per = (gridDim.x + 7) / 8; wg = (blockIdx.x & 7u) * per + (blockIdx.x >> 3); }
    const unsigned g = wg * 4 + wid;
    if (g >= n_items) return;
    const uint4 s = slots[(size_t)g * 64 + lane];
    uint32_t acc = s.x ^ s.y ^ s.z ^ s.w;
    if (MODE == 1) acc ^= state[s.x] ^ state[s.y] ^ state[s.z] ^ state[s.w];
    if (acc == 0x12345u) sink[0] = acc;
}
// 8 B per lane from a 2-MiB table at random places (k_samples_lean's pore-table look-up: the table lives in L2), 16 look-ups per lane
__global__ __launch_bounds__(256) void cal_rgather8(const uint2* __restrict__ tab, const unsigned n_items, uint32_t* __restrict__ sink) {
    const unsigned g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= n_items) return;
    uint32_t h = (g * 64 + (threadIdx.x & 63)) * 2654435761u, acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { h = h * 1664525u + 1013904223u; const uint2 v = tab[h >> 14]; acc ^= v.x + v.y; }
    if (acc == 0x12345u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void cal_write2(uint16_t* __restrict__ a, size_t n, uint32_t v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = (uint16_t)(v + i);
}
__global__ __launch_bounds__(256) void cal_write4(uint32_t* __restrict__ a, size_t n, uint32_t v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = v + (uint32_t)i;
}
__global__ __launch_bounds__(256) void cal_write16(uint4* __restrict__ a, size_t n, uint32_t v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = make_uint4(v, v + 1, v + 2, (uint32_t)i);
}
// every 16 lanes write one whole 64-B line; the lines of a wave-store are far apart (a multiplicative hash of the line index)
__global__ __launch_bounds__(256) void cal_write_lines(uint32_t* __restrict__ a, size_t n_lines, uint32_t v) {
    const size_t nq = n_lines;                                     // (a power of two)
    for (size_t q = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 4; q < nq; q += ((size_t)gridDim.x * 256) >> 4) {
        const size_t line = (q * 2654435761ull) & (nq - 1);        // odd multiplier: a permutation of the lines
        a[line * 16 + (threadIdx.x & 15)] = v + (uint32_t)q;
    }
}

int main(int argc, char** argv) {
    const size_t n_ev = (size_t)1 << 27;                           // events (134 M: a batch of the headline workload has 164 M)
    const int n_part = 64;
    const size_t link_ev = 20000;                                  // events per link
    const int reps = argc > 1 ? atoi(argv[1]) : 3;
    // slots: chain order -> partition-major, chain order inside a partition (a stable counting sort by a random partition)
    std::vector<uint8_t> part(n_ev);
    std::vector<uint32_t> slot(n_ev);
    {
        uint64_t s = 88172645463325252ull;
        std::vector<size_t> cnt(n_part + 1, 0);
        for (size_t e = 0; e < n_ev; e++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; part[e] = (uint8_t)((s >> 33) % n_part); cnt[part[e] + 1]++; }
        for (int p = 0; p < n_part; p++) cnt[p + 1] += cnt[p];
        for (size_t e = 0; e < n_ev; e++) slot[e] = (uint32_t)cnt[part[e]]++;
    }
    (void)link_ev;
    uint32_t *d_slot, *d_state, *d_sink, *d_out;
    const size_t bytes = n_ev * 4;
    CHK(hipMalloc(&d_slot, bytes)); CHK(hipMalloc(&d_state, bytes)); CHK(hipMalloc(&d_out, bytes)); CHK(hipMalloc(&d_sink, 256));
    CHK(hipMemcpy(d_slot, slot.data(), bytes, hipMemcpyHostToDevice));
    CHK(hipMemset(d_state, 1, bytes)); CHK(hipMemset(d_out, 0, bytes));
    const unsigned grid = 256 * 16;
    const unsigned n_items = (unsigned)(n_ev / 256), ggrid = (n_items + 3) / 4;
    printf("known bytes per launch: read16/read32x/read8/read4/slots %zu, gather = slots + %zu useful, write2/4/16/lines %zu\n", bytes, bytes, bytes);
    for (int r = 0; r < reps; r++) {
        hipLaunchKernelGGL(cal_read16, dim3(grid), dim3(256), 0, 0, (const uint4*)d_state, bytes / 16, d_sink);
        hipLaunchKernelGGL(cal_read32x, dim3(grid), dim3(256), 0, 0, (const uint4*)d_state, bytes / 32, d_sink);
        hipLaunchKernelGGL(cal_read8, dim3(grid), dim3(256), 0, 0, (const uint2*)d_state, bytes / 8, d_sink);
        hipLaunchKernelGGL(cal_read4, dim3(grid), dim3(256), 0, 0, (const uint32_t*)d_state, bytes / 4, d_sink);
        hipLaunchKernelGGL((cal_gather<0, false>), dim3(ggrid), dim3(256), 0, 0, (const uint4*)d_slot, d_state, n_items, d_sink);
        hipLaunchKernelGGL((cal_gather<1, false>), dim3(ggrid), dim3(256), 0, 0, (const uint4*)d_slot, d_state, n_items, d_sink);
        hipLaunchKernelGGL((cal_gather<1, true>), dim3(ggrid), dim3(256), 0, 0, (const uint4*)d_slot, d_state, n_items, d_sink);
        hipLaunchKernelGGL(cal_rgather8, dim3(ggrid / 16), dim3(256), 0, 0, (const uint2*)d_state, n_items / 16, d_sink);   // n_ev / 4 look-ups
        hipLaunchKernelGGL(cal_write2, dim3(grid), dim3(256), 0, 0, (uint16_t*)d_out, bytes / 2, (uint32_t)r);
        hipLaunchKernelGGL(cal_write4, dim3(grid), dim3(256), 0, 0, d_out, bytes / 4, (uint32_t)r);
        hipLaunchKernelGGL(cal_write16, dim3(grid), dim3(256), 0, 0, (uint4*)d_out, bytes / 16, (uint32_t)r);
        hipLaunchKernelGGL(cal_write_lines, dim3(grid), dim3(256), 0, 0, d_out, bytes / 64, (uint32_t)r);
        CHK(hipDeviceSynchronize());
    }
    // wall-clock rates (hipEvents), for orientation
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto timed = [&](const char* name, auto launch, double nbytes) {
        launch(); CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0, 0)); for (int i = 0; i < 5; i++) launch(); CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-16s %8.3f ms  %7.1f GB/s (known bytes)\n", name, ms / 5, nbytes / (ms / 5 * 1e-3) / 1e9);
    };
    timed("cal_read16", [&] { hipLaunchKernelGGL(cal_read16, dim3(grid), dim3(256), 0, 0, (const uint4*)d_state, bytes / 16, d_sink); }, (double)bytes);
    timed("cal_read4", [&] { hipLaunchKernelGGL(cal_read4, dim3(grid), dim3(256), 0, 0, (const uint32_t*)d_state, bytes / 4, d_sink); }, (double)bytes);
    timed("cal_slots", [&] { hipLaunchKernelGGL((cal_gather<0, false>), dim3(ggrid), dim3(256), 0, 0, (const uint4*)d_slot, d_state, n_items, d_sink); }, (double)bytes);
    timed("cal_gather", [&] { hipLaunchKernelGGL((cal_gather<1, false>), dim3(ggrid), dim3(256), 0, 0, (const uint4*)d_slot, d_state, n_items, d_sink); }, 2.0 * bytes);
    timed("cal_gather_xcd", [&] { hipLaunchKernelGGL((cal_gather<1, true>), dim3(ggrid), dim3(256), 0, 0, (const uint4*)d_slot, d_state, n_items, d_sink); }, 2.0 * bytes);
    timed("cal_rgather8", [&] { hipLaunchKernelGGL(cal_rgather8, dim3(ggrid / 16), dim3(256), 0, 0, (const uint2*)d_state, n_items / 16, d_sink); }, (double)(n_ev / 4) * 8);
    timed("cal_write2", [&] { hipLaunchKernelGGL(cal_write2, dim3(grid), dim3(256), 0, 0, (uint16_t*)d_out, bytes / 2, 1u); }, (double)bytes);
    timed("cal_write4", [&] { hipLaunchKernelGGL(cal_write4, dim3(grid), dim3(256), 0, 0, d_out, bytes / 4, 1u); }, (double)bytes);
    timed("cal_write16", [&] { hipLaunchKernelGGL(cal_write16, dim3(grid), dim3(256), 0, 0, (uint4*)d_out, bytes / 16, 1u); }, (double)bytes);
    timed("cal_write_lines", [&] { hipLaunchKernelGGL(cal_write_lines, dim3(grid), dim3(256), 0, 0, d_out, bytes / 64, 1u); }, (double)bytes);
    return 0;
}
