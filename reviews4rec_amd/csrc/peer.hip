// Device-side gradient exchange over peer-mapped buffers (data parallel, SURVEY.md 8e C1').
//
// The reference is single-process (main.py:407); this is the exchange step of the data-parallel form of its
// training step (main.py:56-60: mean loss -> backward -> optimizer.step on the SUM of the ranks' gradients).
// With RCCL the 0.73 MB gradient bucket of DeepCoNN costs a collective launch and its protocol's hops; on an
// xGMI mesh every GPU can write every other GPU's memory directly, so the exchange can be ONE kernel per rank:
//   r4r_peer_segment_*  one FINE-GRAINED allocation per rank (hipExtMallocWithFlags: the flags are polled while
//                   peers write them, which coarse-grained memory only promises at kernel boundaries), exported /
//                   imported as a 64-byte IPC handle (hipIpcGetMemHandle / hipIpcOpenMemHandle);
//   r4r_peer_push   copies this rank's flat gradient into slot `rank` of EVERY rank's gathered buffer, fences at
//                   system scope, and the last workgroup to finish raises this rank's flag (= the step's epoch) in
//                   every rank's flag array -- and, given `wait_flags`, stays to wait for the peers' flags in MY
//                   array (lane r spins, bounded, on rank r's flag): the whole exchange is then ONE launch;
//   r4r_peer_wait   the wait alone (one wave), for callers that put work between the push and the wait;
//   r4r_adam_gathered (adam.hip) then sums the `world` slots in rank order inside the optimiser launch: identical
//                   bits on every rank.
// Two gathered buffers alternate by epoch parity: a rank can run at most one step ahead of its peers (it waits for
// their step-k flags before its step-k update), and a peer's step-k read is stream-ordered before its step-(k+1)
// push, so the buffer a fast rank writes for step k+1 is never the one a slow rank still reads for step k.
// Nothing here assumes a placement: correctness comes from the system-scope release (push) / acquire (wait).
#include "common.h"
#include "peer_device.h"
#include <string.h>

#define R4R_HIP(expr)                                                        \
    do {                                                                     \
        hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess) {                                              \
            r4r::set_error("%s: %s", #expr, hipGetErrorString(e_));          \
            return R4R_ERR_LAUNCH;                                           \
        }                                                                    \
    } while (0)

namespace r4r {

constexpr int PEER_THREADS = 256;

struct PeerPush {
    const float *src;
    float *dst[PEER_MAX_WORLD];        // rank r's gathered buffer of this epoch's parity (r == rank: my own)
    unsigned *flags[PEER_MAX_WORLD];   // rank r's flag array [world]
    unsigned *arrive;                  // my arrival counter (device memory, zeroed by the host once)
    const unsigned *wait_flags;        // MY flag array: the last workgroup waits on it after raising the flags (null: no wait)
    unsigned *timed_out;
    unsigned long long max_ticks;
    int64_t n4;                        // float4 elements
    int rank, world;
    unsigned epoch;
};

typedef float peer_f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(PEER_THREADS) void peer_push_kernel(PeerPush a) {
    const int64_t stride = (int64_t)gridDim.x * PEER_THREADS;
    const peer_f32x4 *src = reinterpret_cast<const peer_f32x4 *>(a.src);
    for (int64_t i = (int64_t)blockIdx.x * PEER_THREADS + threadIdx.x; i < a.n4; i += stride) {
        const peer_f32x4 v = src[i];
        for (int r = 0; r < a.world; ++r)
            reinterpret_cast<peer_f32x4 *>(a.dst[r])[(int64_t)a.rank * a.n4 + i] = v;
    }
    // every thread's stores reach the peers' memories before this workgroup counts as arrived
    __threadfence_system();
    __syncthreads();
    __shared__ unsigned last;
    if (threadIdx.x == 0) {
        const unsigned k = __hip_atomic_fetch_add(a.arrive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = (k == gridDim.x - 1);
    }
    __syncthreads();
    if (last && threadIdx.x < (unsigned)a.world) {
        if (threadIdx.x == 0) __hip_atomic_store(a.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next push
        __hip_atomic_store(a.flags[threadIdx.x] + a.rank, a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (a.wait_flags) {
            peer_wait_lane(a.wait_flags, threadIdx.x, a.epoch, a.timed_out, a.max_ticks);
            __threadfence_system();
        }
    }
}

__global__ __launch_bounds__(64) void peer_wait_kernel(const unsigned *flags, int world, unsigned epoch, unsigned *timed_out,
                                                       unsigned long long max_ticks) {
    if ((int)threadIdx.x < world) peer_wait_lane(flags, threadIdx.x, epoch, timed_out, max_ticks);
    __threadfence_system();
}

}  // namespace r4r

using namespace r4r;

extern "C" int r4r_peer_segment_create(int64_t bytes, void **ptr, uint8_t *handle) {
    R4R_REQUIRE(ptr && handle && bytes > 0, "peer_segment_create: bad arguments");
    void *p = nullptr;
    R4R_HIP(hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained));
    hipError_t e = hipMemset(p, 0, (size_t)bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) { (void)hipFree(p); R4R_HIP(e); }   // (reports e)
    static_assert(sizeof(h) == 64, "IPC handle size");
    memcpy(handle, &h, sizeof(h));
    *ptr = p;
    return 0;
}

extern "C" int r4r_peer_segment_open(const uint8_t *handle, void **ptr) {
    R4R_REQUIRE(ptr && handle, "peer_segment_open: null pointer");
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    R4R_HIP(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
    return 0;
}

extern "C" int r4r_peer_segment_close(void *ptr) {
    if (ptr) R4R_HIP(hipIpcCloseMemHandle(ptr));
    return 0;
}

extern "C" int r4r_peer_segment_destroy(void *ptr) {
    if (ptr) R4R_HIP(hipFree(ptr));
    return 0;
}

extern "C" int r4r_peer_push(const float *src, int64_t numel, const uint64_t *peer_dst, const uint64_t *peer_flags,
                             uint32_t *arrive, int rank, int world, uint32_t epoch, const uint32_t *wait_flags,
                             uint32_t *timed_out, double timeout_s, void *stream) {
    R4R_REQUIRE(src && peer_dst && peer_flags && arrive, "peer_push: null pointer");
    R4R_REQUIRE(world >= 1 && world <= PEER_MAX_WORLD && rank >= 0 && rank < world, "peer_push: rank %d of %d (<= %d ranks)", rank,
                world, PEER_MAX_WORLD);
    R4R_REQUIRE(numel >= 0 && numel % 4 == 0, "peer_push: numel %lld must be a multiple of 4", (long long)numel);
    R4R_REQUIRE(!wait_flags || (timed_out && timeout_s > 0 && timeout_s <= 60), "peer_push: waiting needs timed_out and a timeout in (0, 60] s");
    PeerPush a;
    a.wait_flags = wait_flags; a.timed_out = timed_out; a.max_ticks = (unsigned long long)((wait_flags ? timeout_s : 0.0) * 1e8);
    a.src = src; a.n4 = numel / 4; a.rank = rank; a.world = world; a.epoch = epoch; a.arrive = arrive;
    for (int r = 0; r < PEER_MAX_WORLD; ++r) {
        a.dst[r] = reinterpret_cast<float *>(peer_dst[r < world ? r : 0]);
        a.flags[r] = reinterpret_cast<unsigned *>(peer_flags[r < world ? r : 0]);
        R4R_REQUIRE(a.dst[r] && a.flags[r] && (reinterpret_cast<uintptr_t>(a.dst[r]) & 15) == 0, "peer_push: bad peer buffer %d", r);
    }
    R4R_REQUIRE((reinterpret_cast<uintptr_t>(src) & 15) == 0, "peer_push: source must be 16-byte aligned");
    int64_t blocks = cdiv(a.n4, PEER_THREADS);            // one float4 per thread: every load of the copy in one round trip
    if (blocks < 1) blocks = 1;
    if (blocks > 512) blocks = 512;
    peer_push_kernel<<<(unsigned)blocks, PEER_THREADS, 0, as_stream(stream)>>>(a);
    return check_launch("peer_push");
}

extern "C" int r4r_peer_wait(const uint32_t *flags, int world, uint32_t epoch, uint32_t *timed_out, double timeout_s, void *stream) {
    R4R_REQUIRE(flags && timed_out, "peer_wait: null pointer");
    R4R_REQUIRE(world >= 1 && world <= PEER_MAX_WORLD, "peer_wait: %d ranks (<= %d)", world, PEER_MAX_WORLD);
    R4R_REQUIRE(timeout_s > 0 && timeout_s <= 60, "peer_wait: timeout %.3f s outside (0, 60]", timeout_s);
    peer_wait_kernel<<<1, 64, 0, as_stream(stream)>>>(flags, world, epoch, timed_out, (unsigned long long)(timeout_s * 1e8));
    return check_launch("peer_wait");
}
