// What does the chip deliver for the gather-add-max launch's access pattern with NOTHING else in the kernel?
// Every half-wave reads 400 contiguous bytes (25 lanes x 16 B) of a random row of a [rows x 1216 B] table that the
// previous launch wrote (36 MB at cfg3: L2 / Infinity-Cache resident; 390 MB: the HBM point), DEPTH independent loads
// per lane in flight, 1,024 workgroups of 256 threads resident at once (the product launch's shape).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/l2_gather.hip -o l2_gather && ./l2_gather
// Prints GB/s of requested bytes and 128-byte lines touched per ns for uniform and Zipf-distributed rows.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int STRIDE = 304;   // floats per row (1,216 B)

template <int DEPTH>
__global__ __launch_bounds__(256) void gather(const float *__restrict__ tab, const int *__restrict__ rows, int per_worker, float *sink) {
    const int worker = (blockIdx.x * 256 + threadIdx.x) >> 5, wl = threadIdx.x & 31;
    const int *mine = rows + (size_t)worker * per_worker;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (wl < 25) {
        for (int k = 0; k < per_worker; k += DEPTH) {
            f32x4 v[DEPTH][3];
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const float *r = tab + (size_t)mine[k + u] * STRIDE + wl * 4;
#pragma unroll
                for (int t = 0; t < 3; ++t) v[u][t] = *reinterpret_cast<const f32x4 *>(r + t * 100);
            }
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) acc += (v[u][0] + v[u][1]) + v[u][2];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = 1.f;
}

int main(int argc, char **argv) {
    const int nrows = argc > 1 ? atoi(argv[1]) : 29547;
    const int wgs = 1024, workers = wgs * 8, per_worker = 28;           // 28 tokens per worker: four rounds of 7
    float *tab, *sink; int *rows;
    hipMalloc(&tab, (size_t)nrows * STRIDE * 4); hipMemset(tab, 0, (size_t)nrows * STRIDE * 4);
    hipMalloc(&sink, 4); hipMalloc(&rows, (size_t)workers * per_worker * 4);
    std::vector<double> cdf(nrows);
    double s = 0; for (int i = 0; i < nrows; ++i) { s += 1.0 / (i + 1); cdf[i] = s; }
    std::vector<int> perm(nrows); for (int i = 0; i < nrows; ++i) perm[i] = i;
    srand(1); for (int i = nrows - 1; i > 0; --i) { int j = rand() % (i + 1); int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
    for (int dist = 0; dist < 2; ++dist) {
        std::vector<int> h((size_t)workers * per_worker);
        for (auto &x : h) {
            if (dist == 0) x = rand() % nrows;
            else { double u = (rand() / (double)RAND_MAX) * s; int lo = 0, hi = nrows - 1; while (lo < hi) { int m = (lo + hi) / 2; if (cdf[m] < u) lo = m + 1; else hi = m; } x = perm[lo]; }
        }
        hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(tab, 0, (size_t)nrows * STRIDE * 4);             // (the table is rewritten before every launch, like the GEMM does)
            hipEventRecord(e0);
            gather<7><<<wgs, 256>>>(tab, rows, per_worker, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)workers * per_worker * 1200.0;
            printf("%s rows, %d rows (%.0f MB): %.1f us, %.0f GB/s requested, %.1f lines/ns (10 lines per row)\n", dist ? "zipf   " : "uniform",
                   nrows, nrows * 1216.0 / 1e6, ms * 1e3, bytes / ms / 1e6, workers * (double)per_worker * 10 / (ms * 1e6));
        }
    }
    return 0;
}
