// Error reporting and version for libr4r_hip.so (see include/r4r.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/r4r.h"

namespace r4r {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace r4r

extern "C" int r4r_version(void) { return 1; }
extern "C" const char *r4r_last_error(void) { return r4r::g_err; }

// ---------------------------------------------------------------------------
// Optional live kernel timing (bench.py's roofline leg): when enabled, the
// instrumented launch sites bracket their DOMINANT kernel with hipEvents on the
// launch stream.  r4r_timing_read() synchronises those events and accumulates.
// Off by default: zero overhead on the normal path.
// ---------------------------------------------------------------------------
namespace r4r {
struct TimedSpan { int id; hipEvent_t a, b; bool a_shared; };   // a_shared: `a` is the previous span's `b` (recycled once)
static unsigned g_timing_mask = 0;             // bit i set -> slot i is instrumented
static std::vector<TimedSpan> g_spans;         // spans recorded since the last read
static size_t g_last_ended = 0;                // 1-based index of the span whose end event was recorded last
static std::vector<hipEvent_t> g_pool;         // recycled events: no create/destroy per launch
static double g_total_ms[R4R_TIMING_SLOTS];
static long long g_count[R4R_TIMING_SLOTS];

bool timing_on() { return g_timing_mask != 0; }

static hipEvent_t take_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    // Timing-only events: no system-scope fence when the event completes (hipEventRecord's default release writes
    // the L2 back -- 35 MB of projected rows between the GEMM and the gather that reads them -- which made an
    // instrumented step ~30 us longer than the steps it samples).  R4R_TIMING_SYSTEM_FENCE=1 restores the default.
    static const unsigned flags = getenv("R4R_TIMING_SYSTEM_FENCE") ? hipEventDefault : hipEventDisableSystemFence;
    hipEvent_t e;
    (void)hipEventCreateWithFlags(&e, flags);
    return e;
}

// chain: this span starts where the span recorded LAST ended -- same stream, nothing launched in between (the caller
// says so) -- and takes that span's end event as its start: one event record less per instrumented step (a record is a
// barrier packet in the queue: several microseconds of bubble each).
void timing_begin(int id, hipStream_t st, void **token, bool chain) {
    *token = nullptr;
    if (!(g_timing_mask & (1u << id))) return;
    TimedSpan s;
    s.id = id;
    // (only behind a span of ANOTHER slot that has ended and is the last one recorded: with the neighbour's slot
    // masked off the last span is an older launch's, and chaining to it would time everything in between)
    s.a_shared = chain && !g_spans.empty() && g_spans.back().id != id && g_last_ended == g_spans.size();
    s.a = s.a_shared ? g_spans.back().b : take_event();
    s.b = take_event();
    if (!s.a_shared) (void)hipEventRecord(s.a, st);
    g_spans.push_back(s);
    *token = reinterpret_cast<void *>(g_spans.size());
}
void timing_begin(int id, hipStream_t st, void **token) { timing_begin(id, st, token, false); }

// A span measured by the kernel's OWN dispatch packet (hipExtLaunchKernel's start / stop events): no event records --
// no barrier packets, no release fences -- around the kernel, so an instrumented step costs what a plain one does
// (two hipEventRecords around the projection GEMM made the sampled step ~30 us longer: 1.5 % of the driver's 20-step
// region).  -> false when the slot is not instrumented (launch the plain way).
bool timing_kernel_events(int id, hipEvent_t *start, hipEvent_t *stop) {
    if (!(g_timing_mask & (1u << id))) return false;
    TimedSpan s;
    s.id = id;
    s.a_shared = false;
    s.a = take_event();
    s.b = take_event();
    g_spans.push_back(s);
    g_last_ended = g_spans.size();                          // (a chained span may start at this kernel's stop event)
    *start = s.a;
    *stop = s.b;
    return true;
}

void timing_end(void *token, hipStream_t st) {
    const size_t i = reinterpret_cast<size_t>(token) - 1;
    (void)hipEventRecord(g_spans[i].b, st);
    g_last_ended = i + 1;
}
}  // namespace r4r

extern "C" int r4r_timing_enable(int mask) {
    r4r::g_timing_mask = (unsigned)mask;
    return R4R_OK;
}

extern "C" int r4r_timing_read(int slot, double *total_ms, int64_t *count, int reset) {
    if (slot < 0 || slot >= R4R_TIMING_SLOTS || !total_ms || !count) {
        r4r::set_error("timing_read: bad arguments");
        return R4R_ERR_ARG;
    }
    for (auto &s : r4r::g_spans) {
        (void)hipEventSynchronize(s.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            r4r::g_total_ms[s.id] += ms;
            r4r::g_count[s.id] += 1;
        }
        if (!s.a_shared) r4r::g_pool.push_back(s.a);
        r4r::g_pool.push_back(s.b);
    }
    r4r::g_spans.clear();
    r4r::g_last_ended = 0;
    *total_ms = r4r::g_total_ms[slot];
    *count = r4r::g_count[slot];
    if (reset) {
        memset(r4r::g_total_ms, 0, sizeof(r4r::g_total_ms));
        memset(r4r::g_count, 0, sizeof(r4r::g_count));
    }
    return R4R_OK;
}
