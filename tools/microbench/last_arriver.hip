// Microbenchmark: "last-arriving workgroup consumes what the others produced" across XCDs without a
// device-scope fence.  Producers write their values with agent-scope relaxed atomic stores (sc1:
// write-through past the XCD's L2), wait for the stores to be acknowledged (s_waitcnt vmcnt(0)),
// then bump a per-group counter with a relaxed agent-scope atomic; the workgroup that sees the last
// count reads every producer's values with agent-scope atomic loads and checks them.  MODE 1 does
// the same with plain stores + __threadfence() (the textbook form) for the cost comparison; MODE 2
// (plain loads in the consumer) and MODE 3 (plain stores in the producers) are negative controls:
// they must show stale values if the test can see them at all.
//   ./last_arriver <groups> <producers per group> <iterations> <mode>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int VALS = 256;   // floats per producer (one per thread)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *buf, unsigned *cnt, unsigned *errors, int per_group, unsigned iter) {
    const int g = blockIdx.x / per_group, p = blockIdx.x % per_group;
    float *mine = buf + ((size_t)g * per_group + p) * VALS;
    const float v = (float)(iter * 131u + blockIdx.x * 7u + threadIdx.x);
    __shared__ unsigned last;
    if (MODE != 1) {
        if (MODE == 3) mine[threadIdx.x] = v;               // negative control: plain (write-back) store
        else __hip_atomic_store(mine + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0x0f70 & ~0x3f | 0);      // vmcnt(0) (gfx9 encoding: vmcnt low bits 3:0 and 15:14)
        __syncthreads();
        if (threadIdx.x == 0)
            last = __hip_atomic_fetch_add(cnt + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        mine[threadIdx.x] = v;
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) last = atomicAdd(cnt + g, 1u);
    }
    __syncthreads();
    if (last != (iter + 1) * per_group - 1) return;          // not the last of this iteration (counter never reset)
    unsigned bad = 0;
    for (int q = 0; q < per_group; ++q) {
        const float *src = buf + ((size_t)g * per_group + q) * VALS;
        const float want = (float)(iter * 131u + (g * per_group + q) * 7u + threadIdx.x);
        float got;
        if (MODE == 0 || MODE == 3) got = __hip_atomic_load(src + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 2) got = src[threadIdx.x];          // negative control: plain (cacheable) load
        else { __threadfence(); got = src[threadIdx.x]; }
        bad += got != want;
    }
    if (bad) atomicAdd(errors, bad);
}

int main(int argc, char **argv) {
    const int groups = argc > 1 ? atoi(argv[1]) : 128, per = argc > 2 ? atoi(argv[2]) : 8;
    const int iters = argc > 3 ? atoi(argv[3]) : 2000, mode = argc > 4 ? atoi(argv[4]) : 0;
    float *buf; unsigned *cnt, *err;
    hipMalloc(&buf, (size_t)groups * per * VALS * 4);
    hipMalloc(&cnt, groups * 4); hipMalloc(&err, 4);
    hipMemset(buf, 0, (size_t)groups * per * VALS * 4); hipMemset(cnt, 0, groups * 4); hipMemset(err, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) k<0><<<groups * per, 256>>>(buf, cnt, err, per, (unsigned)it);
        else if (mode == 1) k<1><<<groups * per, 256>>>(buf, cnt, err, per, (unsigned)it);
        else if (mode == 2) k<2><<<groups * per, 256>>>(buf, cnt, err, per, (unsigned)it);
        else k<3><<<groups * per, 256>>>(buf, cnt, err, per, (unsigned)it);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned h; hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost);
    printf("mode %d: %d groups x %d producers, %d launches: %.2f us per launch, %u stale values\n", mode, groups, per, iters,
           1000.f * ms / iters, h);
    return 0;
}
