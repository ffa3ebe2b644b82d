// Microbenchmark: how fast does the chip start workgroups?  A kernel whose workgroups do one dependent load -> store
// (so each lives ~1-2 us), N workgroups of T threads, with V VGPRs reserved per lane (launch_bounds / asm clobbers).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int T, int REGS>
__global__ __launch_bounds__(T) void k(float *buf, int n) {
    float v[REGS];
    const size_t i = (size_t)blockIdx.x * T + threadIdx.x;
#pragma unroll
    for (int r = 0; r < REGS; ++r) v[r] = buf[(i + r * 64) % n];
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < REGS; ++r) s += v[r] * (float)(r + 1);
    buf[i % n] = s;
}
template <int T, int REGS>
void run(float *buf, int n, int wgs) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        k<T, REGS><<<wgs, T>>>(buf, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("threads %4d regs %3d workgroups %6d: %7.2f us  -> %6.1f workgroups/us, %7.1f waves/us\n", T, REGS, wgs, best * 1e3,
           wgs / (best * 1e3), wgs * (T / 64) / (best * 1e3));
}
int main() {
    const int n = 1 << 26;
    float *buf; hipMalloc(&buf, (size_t)n * 4); hipMemset(buf, 0, (size_t)n * 4);
    for (int wgs : {256, 1024, 2264, 6715, 20000, 60000}) {
        run<256, 4>(buf, n, wgs); run<256, 48>(buf, n, wgs); run<1024, 4>(buf, n, wgs / 4); run<64, 4>(buf, n, wgs * 4);
    }
    return 0;
}
