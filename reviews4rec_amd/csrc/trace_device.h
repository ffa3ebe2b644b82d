// Stage timeline of a one-workgroup-per-rating kernel (make trace, tools/head_trace.py): under
// -DR4R_TRACE thread 0 of each workgroup writes s_memrealtime (100 MHz) at each HEAD_STAMP(k),
// 32 words per workgroup.  Never compiled into the product library.
#pragma once
#ifdef R4R_TRACE
#define HEAD_TRACE_DEFINE(setter)                                                                      \
    static __device__ unsigned long long *g_head_trace = nullptr;                                      \
    extern "C" int setter(void *buf) {                                                                 \
        return hipMemcpyToSymbol(HIP_SYMBOL(g_head_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;   \
    }
#define HEAD_STAMP(k)                                                                                  \
    if (g_head_trace && threadIdx.x == 0) g_head_trace[(size_t)blockIdx.x * 32 + (k)] = wall_clock64();
#else
#define HEAD_TRACE_DEFINE(setter)
#define HEAD_STAMP(k)
#endif

#ifdef R4R_TRACE
// Role timeline of a backward launch (tools/head_trace.py --backward): 4 words per workgroup --
// start, end (s_memrealtime, 100 MHz), z-slice
static __device__ unsigned long long *g_bwd_trace = nullptr;
#define BWD_TRACE_DEFINE(setter)                                                                       \
    extern "C" int setter(void *buf) {                                                                 \
        return hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;    \
    }
#define BWD_STAMP(k, val)                                                                              \
    if (g_bwd_trace && threadIdx.x == 0)                                                               \
        g_bwd_trace[((size_t)blockIdx.z * gridDim.y * gridDim.x + blockIdx.y * gridDim.x + blockIdx.x) * 4 + (k)] = (val);
#else
#define BWD_TRACE_DEFINE(setter)
#define BWD_STAMP(k, val)
#endif

