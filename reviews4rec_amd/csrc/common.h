// Internal helpers shared by the gfx950 kernels of libr4r_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/r4r.h"

namespace r4r {

void set_error(const char *fmt, ...);

// live kernel timing (capi.cpp); slots are the R4R_TIMING_* ids of include/r4r.h
bool timing_on();
void timing_begin(int id, hipStream_t st, void **token);
void timing_begin(int id, hipStream_t st, void **token, bool chain);
void timing_end(void *token, hipStream_t st);

bool timing_kernel_events(int id, hipEvent_t *start, hipEvent_t *stop);

struct ScopedTiming {
    void *tok = nullptr;
    hipStream_t st;
    // chain: the span starts at the end event of the span recorded last (nothing was launched in between)
    ScopedTiming(int id, hipStream_t s, bool chain = false) : st(s) { if (timing_on()) timing_begin(id, s, &tok, chain); }
    ~ScopedTiming() { if (tok) timing_end(tok, st); }
};

// Launch `kernel`; when timing slot `id` is on, with the dispatch packet's own start / stop timestamps (capi.cpp).
template <class K, class... A>
inline void launch_timed(int id, K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st, A... args) {
    hipEvent_t e0, e1;
    if (timing_on() && timing_kernel_events(id, &e0, &e1))
        hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, st, e0, e1, 0, args...);
    else
        hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
}

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return R4R_ERR_LAUNCH;
    }
    return R4R_OK;
}

#define R4R_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            r4r::set_error(__VA_ARGS__);  \
            return R4R_ERR_ARG;           \
        }                                 \
    } while (0)

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int WAVE = 64;

// Sum over the 64 lanes, result in every lane.  DPP moves inside the 16-lane rows (four VALU
// instructions), then the four row totals are read as scalars and added in a fixed order -- a
// ds_bpermute butterfly (what __shfl_xor compiles to) costs ~100 cycles per step and made the
// reductions of the head kernels microseconds long.
__device__ __forceinline__ float wave_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}

// Maximum over the 64 lanes, result in every lane (same shape as wave_sum).
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false)));
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// Lanes of ONE wave handing values to one another through LDS without a workgroup barrier: a wave's LDS operations
// execute in order, so what is needed is only that hipcc keeps the writes before the reads (a wavefront-scope fence
// costs no instruction).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Philox4x32-10, first word of the draw for counter `ctr` (the fused steps draw one word per
// element; smallops.hip keeps the four-word form for the stand-alone dropout op)
__device__ __forceinline__ uint32_t philox_first_word(uint64_t ctr, uint64_t seed) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0u, c3 = 0u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Kernel arguments read WHERE THEY ARE USED.  hipcc loads every by-value kernel argument in the kernel's first block
// (the kernarg segment is known dereferenceable and invariant, so every s_load is hoisted there), keeps all of them
// live in scalar registers for the whole kernel and spills what does not fit into the lanes of vector registers -- a
// v_writelane / v_readlane pair around every use (tools/isa_scan.py: 128-330 spilled scalars and up to 1,400 lane
// operations in the role kernels, whose ~70 arguments are mostly ANOTHER role's).  kernel_args<T>() is the kernel's one
// argument struct (it must be the kernel's FIRST parameter) seen through a pointer the optimiser knows nothing about:
// each field becomes an s_load next to its use, inside the role's own branch.
template <class T>
__device__ __forceinline__ const T &kernel_args() {
    typedef const T __attribute__((address_space(4))) *kernarg_ptr;
    kernarg_ptr p = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const T *)p;
}


}  // namespace r4r
