// Device side of the peer-mapped exchanges (peer.hip, mf_engine.hip): the bounded wait on a rank's flag array.
#pragma once
#include "common.h"

namespace r4r {

constexpr int PEER_MAX_WORLD = 16;

// lane r of the calling wave waits for rank r.  Bounded: a peer that never arrives sets *timed_out instead of hanging the GPU.
// ACQ = false: the flag is read with relaxed system-scope loads and NO cache invalidation follows -- for callers whose
// every later read of the exchanged data goes to the fine-grained (uncached) segment itself, which no cache holds: an
// acquire's `buffer_inv sc0 sc1` per polling wave cost a 4,000-workgroup launch 60 us (profiles/r05_negatives.txt).
template <bool ACQ = true>
__device__ __forceinline__ void peer_wait_lane(const unsigned *flags, int r, unsigned epoch, unsigned *timed_out,
                                               unsigned long long max_ticks) {
    const unsigned long long t0 = wall_clock64();           // 100 MHz
    for (;;) {
        const unsigned f = ACQ ? __hip_atomic_load(flags + r, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)
                               : __hip_atomic_load(flags + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int)(f - epoch) >= 0) break;
        if (__hip_atomic_load(timed_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;   // somebody already gave up: the step is lost
        if (wall_clock64() - t0 > max_ticks) { __hip_atomic_store(timed_out, 1u + (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        __builtin_amdgcn_s_sleep(8);
    }
}

}  // namespace r4r
