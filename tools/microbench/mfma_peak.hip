// Microbenchmark: sustained fp32 MFMA rate (v_mfma_f32_16x16x4_f32) with nothing else in the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int NT>
__global__ __launch_bounds__(NT) void k(float *out, int iters, float a, float b) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (threadIdx.x % 64 == 0 && blockIdx.x < 4) { unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4); out[1000000 + blockIdx.x * 16 + threadIdx.x / 64] = __uint_as_float(hw); }
}
int main(int argc, char **argv) {
    int wgs = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 2000, reps = argc > 3 ? atoi(argv[3]) : 5;
    float *out;
    hipMalloc(&out, (size_t)(1000000 + 64) * 4 + (size_t)wgs * 512 * 4);
    int nt = argc > 4 ? atoi(argv[4]) : 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        if (nt == 512) k<20, 512><<<wgs, 512>>>(out, iters, 1.f, 2.f); else k<20, 256><<<wgs, 256>>>(out, iters, 1.f, 2.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)wgs * (nt / 64) * iters * 20 * 2048.0;
        printf("wgs %d iters %d: %.3f ms  %.1f TFLOP/s\n", wgs, iters, ms, flops / ms * 1e-9);
    }
    unsigned hw[64];
    hipMemcpy(hw, out + 1000000, sizeof(hw), hipMemcpyDeviceToHost);
    for (int b = 0; b < 2; ++b) { printf("wg %d wave->simd:", b); for (int w = 0; w < nt / 64; ++w) printf(" %u", (hw[b * 16 + w] >> 4) & 3); printf("  cu:"); for (int w = 0; w < nt / 64; ++w) printf(" %u", (hw[b * 16 + w] >> 8) & 0xf); printf("\n"); }
    return 0;
}
