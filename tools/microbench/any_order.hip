// Microbenchmark: do two kernels of one stream overlap when the second is launched with hipExtAnyOrderLaunch (no
// barrier bit on its dispatch packet)?  Each kernel is 64 workgroups that spin ~20 us.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(unsigned long long ticks, int *sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0 && ticks == 1) sink[blockIdx.x] = 1;
}
int main() {
    int *sink; hipMalloc(&sink, 4096);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0, st);
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, 2000ull, sink);            // ~20 us (100 MHz clock)
            if (mode == 0) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, 2000ull, sink);
            if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, 2000ull, sink);
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%s: %.1f us\n", mode == 0 ? "two kernels, in order      " : mode == 1 ? "second with AnyOrderLaunch " : "one kernel                 ", best * 1e3);
    }
    return 0;
}
