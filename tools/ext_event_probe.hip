// What do events attached to a launch (hipExtLaunchKernelGGL's startEvent / stopEvent) measure, next to recorded ones, and does
// hipExtAnyOrderLaunch let two kernels of one stream overlap?   hipcc --offload-arch=gfx950 -O2 tools/ext_event_probe.hip -o /tmp/ext_event_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>
__global__ void spin(unsigned long long cycles, unsigned int* sink) {
    const unsigned long long t0 = wall_clock64();
    unsigned int x = threadIdx.x;
    while (wall_clock64() - t0 < cycles) x = x * 1664525u + 1013904223u;
    if (x == 0xdeadbeefu) *sink = x;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned int* sink; CK(hipMalloc(&sink, 4));
    hipEvent_t e[10]; for (auto& x : e) CK(hipEventCreate(&x));
    // wall_clock64 runs at 100 MHz: 10000 ticks = 100 us
    const unsigned long long A = 20000, B = 5000, C = 10000;   // 200, 50, 100 us
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e[0], s));
        hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, e[1], e[2], 0, A, sink);
        hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, e[3], e[4], 0, B, sink);
        hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, e[5], e[6], 0, C, sink);
        CK(hipEventRecord(e[7], s));
        CK(hipStreamSynchronize(s));
        auto el = [&](int a, int b) { float ms = -1; hipError_t r = hipEventElapsedTime(&ms, e[a], e[b]); if (r != hipSuccess) { printf("(%d,%d): %s  ", a, b, hipGetErrorString(r)); return -1.f; } return ms * 1e3f; };
        printf("rep %d: kernels A 200, B 50, C 100 us.  rec0->rec7 %.1f | A.start->A.stop %.1f  B.start->B.stop %.1f  C.start->C.stop %.1f\n", rep, el(0, 7), el(1, 2), el(3, 4), el(5, 6));
        printf("   A.start->B.start %.1f  A.start->C.start %.1f  A.stop->C.stop %.1f  A.start->C.stop %.1f  A.stop->B.start %.1f\n", el(1, 3), el(1, 5), el(2, 6), el(1, 6), el(2, 3));
        printf("   rec0->A.start %.1f  rec0->A.stop %.1f  rec0->C.start %.1f  C.stop->rec7 %.1f  C.start->rec7 %.1f  A.start->rec7 %.1f\n", el(0, 1), el(0, 2), el(0, 5), el(6, 7), el(5, 7), el(1, 7));
    }
    // event-sync / query on a launch-attached stop event; a wait on another stream
    hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, e[8], 0, A, sink);
    printf("query right after launch: %s\n", hipGetErrorString(hipEventQuery(e[8])));
    CK(hipStreamWaitEvent(s2, e[8], 0));
    CK(hipEventRecord(e[9], s2));
    CK(hipEventSynchronize(e[9]));
    printf("query after the other stream waited for it: %s\n", hipGetErrorString(hipEventQuery(e[8])));
    CK(hipEventSynchronize(e[8]));
    // overlap: two kernels of 64 workgroups, in order / any order; and the cost of N recorded events between two kernels
    for (int mode = 0; mode < 2; mode++) {
        CK(hipStreamSynchronize(s));
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 20; i++) {
            hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, 0, C, sink);
            hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0, C, sink);
        }
        CK(hipStreamSynchronize(s));
        printf("40 kernels of 100 us, every second one %s: %.1f us\n", mode ? "any-order" : "in order", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    for (int nrec = 0; nrec <= 3; nrec++) {
        CK(hipStreamSynchronize(s));
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 200; i++) {
            hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, 500ull, sink);   // 5 us
            for (int r = 0; r < nrec; r++) CK(hipEventRecord(e[r], s));
        }
        CK(hipStreamSynchronize(s));
        printf("200 kernels of 5 us with %d recorded event(s) behind each: %.2f us per kernel\n", nrec, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200);
    }
    {
        CK(hipStreamSynchronize(s));
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 200; i++) hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, e[0], e[1], 0, 500ull, sink);
        CK(hipStreamSynchronize(s));
        printf("200 kernels of 5 us with a start and a stop event attached: %.2f us per kernel\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200);
    }
    return 0;
}
