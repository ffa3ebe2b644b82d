// Microbenchmark: does LDS read traffic slow the fp32 MFMA stream, and do AGPR accumulators help?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NACC = 20, NRD = 12;
template <int MODE>   // 0: MFMA only; 1: + ds_read_b128 (VGPR acc); 2: + ds_read, AGPR acc via inline asm; 3: AGPR acc, no reads
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
    __shared__ f32x4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) {
        if (a0 == 0.f) {                                    // random operands (power: real data toggles the datapath)
            unsigned h = (i + 1) * 2654435761u + blockIdx.x * 40503u;
            f32x4 v;
            for (int k = 0; k < 4; ++k) { h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; v[k] = ((int)(h & 0xffff) - 32768) * (1.f / 32768.f); }
            lds[i] = v;
        } else lds[i] = (f32x4){a0, b0, a0, b0};
    }
    __syncthreads();
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ops[NRD];
#pragma unroll
    for (int r = 0; r < NRD; ++r) ops[r] = lds[(threadIdx.x + r * 64) & 1023];
    for (int it = 0; it < iters; ++it) {
        f32x4 nxt[NRD];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                const float a = ops[i & 1][kk], b = ops[2 + (i >> 1)][kk];
                if (MODE >= 2) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            }
            if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int r = kk * 3; r < kk * 3 + 3; ++r) nxt[r] = lds[(threadIdx.x + r * 64 + it) & 1023];
            }
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int r = 0; r < NRD; ++r) ops[r] = nxt[r];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(float *out, int wgs, int iters, float a0) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        k<MODE><<<wgs, 256>>>(out, iters, a0, 2.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)wgs * 4 * iters * 80 * 2048.0;
        if (r == 2) printf("a0 %.0f mode %d wgs %d: %.3f ms  %.1f TFLOP/s\n", a0, MODE, wgs, ms, flops / ms * 1e-9);
    }
}
int main(int argc, char **argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 500;
    float *out;
    hipMalloc(&out, (size_t)1024 * 256 * 4);
    for (float a0 : {1.f, 0.f}) for (int wgs : {256, 512}) { run<0>(out, wgs, iters, a0); run<1>(out, wgs, iters, a0); }
    return 0;
}
