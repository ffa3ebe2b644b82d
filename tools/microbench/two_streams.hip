// Microbenchmark: a step of three kernels -- F, then A and B which both depend on F only -- as one stream (F, A, B in
// order) against two streams (F, A on s0; B on s1 behind an event of F; the next F behind an event of B).  Kernels:
// F 5 us, A 10 us (16 workgroups), B 10 us (1,000 workgroups); 200 steps.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long ticks, int *sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0 && ticks == 1) sink[blockIdx.x] = 1;
}
int main() {
    int *sink; hipMalloc(&sink, 1 << 16);
    hipStream_t s0, s1; hipStreamCreate(&s0); hipStreamCreate(&s1);
    hipEvent_t e0, e1, ef, eb; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventCreateWithFlags(&ef, hipEventDisableTiming); hipEventCreateWithFlags(&eb, hipEventDisableTiming);
    const int steps = 200;
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0, s0);
            for (int i = 0; i < steps; ++i) {
                hipLaunchKernelGGL(spin, dim3(32), dim3(256), 0, s0, 500ull, sink);          // F
                if (mode == 0) {
                    hipLaunchKernelGGL(spin, dim3(16), dim3(256), 0, s0, 1000ull, sink);     // A
                    hipLaunchKernelGGL(spin, dim3(1000), dim3(256), 0, s0, 1000ull, sink);   // B
                } else {
                    hipEventRecord(ef, s0);
                    hipStreamWaitEvent(s1, ef, 0);
                    hipLaunchKernelGGL(spin, dim3(1000), dim3(256), 0, s1, 1000ull, sink);   // B beside A
                    hipEventRecord(eb, s1);
                    hipLaunchKernelGGL(spin, dim3(16), dim3(256), 0, s0, 1000ull, sink);     // A
                    hipStreamWaitEvent(s0, eb, 0);
                }
            }
            hipEventRecord(e1, s0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%s: %.2f us per step\n", mode == 0 ? "one stream (F, A, B)            " : "two streams (F, A | B beside A) ", best * 1e3 / steps);
    }
    return 0;
}
