// Microbenchmark: what do the memory instructions of a GEMM K-step cost the fp32 MFMA stream, in SHADER CYCLES
// (s_memtime) and in wall time (s_memrealtime, 100 MHz) separately -- i.e. issue cost vs clock (DVFS) cost.
//   SHAPE 16: 20 accumulators of v_mfma_f32_16x16x4_f32, 80 MFMAs (32 cycles each) per iteration
//   SHAPE 32:  5 accumulators of v_mfma_f32_32x32x2_f32, 40 MFMAs (64 cycles each) per iteration
// per iteration and wave: NRD ds_read_b128 (operands, consumed by the next iteration's MFMAs), NWR ds_write_b128,
// NLD global_load_dwordx4 of 16 rows x 64 B (the staging gather's shape) feeding the writes.
// NT = 256: one wave per SIMD, 512: two.  One workgroup per CU (grid = 256), random operand data.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int NRD, int NWR, int NLD, int NT>
__global__ __launch_bounds__(NT) void k(float *out, const float *src, unsigned long long *stamps, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *lds = reinterpret_cast<f32x4 *>(smem);                    // 4096 x 16 B = 64 KB
    for (int i = threadIdx.x; i < 4096; i += NT) {
        unsigned h = (i + 1) * 2654435761u + blockIdx.x * 40503u;
        f32x4 v;
        for (int c = 0; c < 4; ++c) { h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; v[c] = ((int)(h & 0xffff) - 32768) * (1.f / 32768.f); }
        lds[i] = v;
    }
    __syncthreads();
    constexpr int NACC = SHAPE == 16 ? 20 : 5, NOPS = 12;
    f32x4 acc16[SHAPE == 16 ? NACC : 1];
    f32x16 acc32[SHAPE == 32 ? NACC : 1];
    for (int i = 0; i < (SHAPE == 16 ? NACC : 1); ++i) acc16[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < (SHAPE == 32 ? NACC : 1); ++i)
        for (int c = 0; c < 16; ++c) acc32[i][c] = 0.f;
    f32x4 ops[NOPS], stg[NLD > 0 ? NLD : 1];
    const int lane = threadIdx.x & 63, lrow = lane & 15, q = lane >> 4;
    const int rbase = ((threadIdx.x >> 6) * 512 + lrow * 6 + q) & 4095;       // conflict-free b128 pattern (stride 24 floats)
#pragma unroll
    for (int r = 0; r < NOPS; ++r) ops[r] = lds[(rbase + r * 96) & 4095];
    const float *gp = src + (size_t)blockIdx.x * 65536 + (size_t)(threadIdx.x >> 2) * 300 + (threadIdx.x & 3) * 4;
#pragma unroll
    for (int r = 0; r < (NLD > 0 ? NLD : 1); ++r) stg[r] = (f32x4){1.f, 2.f, 3.f, 4.f};
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        f32x4 nxt[NOPS];
#pragma unroll
        for (int r = 0; r < NOPS; ++r) nxt[r] = ops[r];
        // memory instructions of this K-step (placed by the schedule below)
#pragma unroll
        for (int r = 0; r < NWR; ++r) lds[(2048 + threadIdx.x * 2 + r * 1024 + (it & 1) * 512) & 4095] = stg[NLD > 0 ? r % NLD : 0];
#pragma unroll
        for (int r = 0; r < NLD; ++r) stg[r] = *reinterpret_cast<const f32x4 *>(gp + ((it * 16 + r * 19200) & 65535));
#pragma unroll
        for (int r = 0; r < NRD; ++r) nxt[r % NOPS] = lds[(rbase + r * 96 + it * 4) & 2047];
        constexpr int NM = SHAPE == 16 ? 80 : 40;
        if constexpr (SHAPE == 16) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < NACC; ++i)
                    acc16[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ops[i & 1][kk], ops[2 + (i >> 1)][kk], acc16[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int i = 0; i < NACC; ++i)
                    acc32[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ops[kk >> 2][kk & 3], ops[2 + i * 2 + (kk >> 2)][kk & 3], acc32[i], 0, 0, 0);
        }
        constexpr int NMEM = NRD + NWR + NLD;
        if constexpr (NMEM > 0) {
            constexpr int PER = NM / NMEM > 0 ? NM / NMEM : 1;
#pragma unroll
            for (int i = 0; i < NWR; ++i) { __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, PER, 0); }
#pragma unroll
            for (int i = 0; i < NLD; ++i) { __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, PER, 0); }
#pragma unroll
            for (int i = 0; i < NRD; ++i) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, PER, 0); }
        }
#pragma unroll
        for (int r = 0; r < NOPS; ++r) ops[r] = nxt[r];
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < (SHAPE == 16 ? NACC : 1); ++i) s += acc16[i][0] + acc16[i][1] + acc16[i][2] + acc16[i][3];
    for (int i = 0; i < (SHAPE == 32 ? NACC : 1); ++i)
        for (int c = 0; c < 16; ++c) s += acc32[i][c];
    for (int r = 0; r < (NLD > 0 ? NLD : 1); ++r) s += stg[r][0];
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = w1 - w0; }
}

template <int SHAPE, int NRD, int NWR, int NLD, int NT>
void run(float *out, const float *src, unsigned long long *stamps, int iters) {
    const int wgs = 256;
    auto kern = k<SHAPE, NRD, NWR, NLD, NT>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    std::vector<unsigned long long> h(wgs * 2);
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0);
        kern<<<wgs, NT, 65536>>>(out, src, stamps, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    hipMemcpy(h.data(), stamps, wgs * 16, hipMemcpyDeviceToHost);
    std::vector<double> cyc, wall;
    for (int i = 0; i < wgs; ++i) { cyc.push_back((double)h[i * 2]); wall.push_back((double)h[i * 2 + 1] * 10.0); }   // wall in ns
    std::sort(cyc.begin(), cyc.end()); std::sort(wall.begin(), wall.end());
    const double c = cyc[wgs / 2] / iters, w = wall[wgs / 2] / iters;
    const int waves = NT / 256;
    const double pipe = (SHAPE == 16 ? 80 * 32 : 40 * 64) * waves;                  // MFMA pipe cycles per SIMD and iteration
    const double flops = (double)wgs * (NT / 64) * iters * 80 * 2048.0;
    printf("shape %2d waves/SIMD %d rd %2d wr %d ld %d: %7.1f cyc/iter (pipe %4.0f: %5.1f %% busy)  %6.1f ns/iter  clock %4.0f MHz  kernel %.3f ms %6.1f TFLOP/s\n",
           SHAPE, waves, NRD, NWR, NLD, c, pipe, 100.0 * pipe / c, w, c / w * 1000.0, best, flops / best * 1e-9);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 300;
    float *out, *src;
    unsigned long long *stamps;
    hipMalloc(&out, (size_t)256 * 512 * 4);
    hipMalloc(&src, (size_t)(256 * 65536 + 65536) * 4);
    hipMemset(src, 0, (size_t)(256 * 65536 + 65536) * 4);
    hipMalloc(&stamps, 256 * 16);
#define ROW(S, NT)                                     \
    run<S, 0, 0, 0, NT>(out, src, stamps, iters);      \
    run<S, 6, 0, 0, NT>(out, src, stamps, iters);      \
    run<S, 9, 0, 0, NT>(out, src, stamps, iters);      \
    run<S, 12, 0, 0, NT>(out, src, stamps, iters);     \
    run<S, 18, 0, 0, NT>(out, src, stamps, iters);     \
    run<S, 12, 4, 0, NT>(out, src, stamps, iters);     \
    run<S, 12, 0, 4, NT>(out, src, stamps, iters);     \
    run<S, 12, 4, 4, NT>(out, src, stamps, iters);     \
    run<S, 9, 4, 4, NT>(out, src, stamps, iters);
    ROW(16, 512) ROW(16, 256) ROW(32, 512) ROW(32, 256)
    return 0;
}
