// Token state of the project-then-gather conv (project.hip): mark the tokens a batch uses,
// compact them into dense rows.  Per-workgroup bodies, shared by the stand-alone kernels
// (project.hip) and by the fused DeepCoNN step (engine.hip), which runs them for the NEXT
// batch inside the backward / reduce launches of the current one.
#pragma once
#include "textcnn.h"

namespace r4r {

struct TokenTower {
    const int64_t *idx;          // [N, T]
    int *flags, *slot, *list, *count;   // as in ProjTower
};
struct TokenArgs {
    TokenTower t[MAX_TOWERS];
    int64_t N, V;
    int T, ntower;
};

// ---- mark: block `blk` of `nblk` blocks of `nthreads` threads, all towers
__device__ __forceinline__ void token_mark_block(const TokenArgs &a, int blk, int nblk, int nthreads) {
    const int64_t total = a.N * a.T;
    for (int t = 0; t < a.ntower; ++t) {
        const int64_t *__restrict__ idx = a.t[t].idx;
        int *__restrict__ flags = a.t[t].flags;
        for (int64_t i = (int64_t)blk * nthreads + threadIdx.x; i < total; i += (int64_t)nblk * nthreads)
            flags[idx[i]] = 1;
    }
}

// ---- compact: slot[v] = dense row id of token v (or -1), list[row] = v, count.
// A workgroup of NW waves owns 256 NW G consecutive tokens (G int4 of flags per thread, all
// loaded up front), scans its flags in LDS and reserves a contiguous row range with ONE
// atomicAdd on the tower's counter (G > 1 keeps the number of atomics on that one address in
// the low hundreds at V = 1M).  The order in which workgroups reserve ranges varies from run to
// run, but any token <-> row bijection gives bit-identical results downstream (a projected row
// depends only on its own token), so no global scan or sort is needed.  Flags are cleared as
// they are consumed; `count` is reset by the gather kernel: both are all-zero between uses.
template <int NW, int G = 1>
__device__ __forceinline__ void token_compact_block(const TokenTower &tw, int64_t V, int blk) {
    __shared__ int wsum[G * NW];
    __shared__ int base_row;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t g0 = (int64_t)blk * (64 * NW * G) + tid;  // int4 group g0 + 64 NW g: tokens 4 gi .. 4 gi + 3
    const int64_t ngroups = (V + 3) / 4;                    // flags / slot buffers are padded to 4
    int4 f[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int64_t gi = g0 + (int64_t)g * (64 * NW);
        f[g] = make_int4(0, 0, 0, 0);
        if (gi < ngroups) f[g] = reinterpret_cast<int4 *>(tw.flags)[gi];
    }
    // exclusive prefix of the counts inside the workgroup: wave scans + (G NW)-entry LDS scan
    int cnt[G], incl[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        cnt[g] = f[g].x + f[g].y + f[g].z + f[g].w;
        incl[g] = cnt[g];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl[g], off);
            if (lane >= off) incl[g] += up;
        }
        if (lane == 63) wsum[g * NW + wave] = incl[g];
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int w = 0; w < G * NW; ++w) { const int c = wsum[w]; wsum[w] = run; run += c; }
        base_row = run ? atomicAdd(tw.count, run) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int64_t gi = g0 + (int64_t)g * (64 * NW);
        if (gi >= ngroups) continue;
        int at = base_row + wsum[g * NW + wave] + incl[g] - cnt[g];
        const int fl[4] = {f[g].x, f[g].y, f[g].z, f[g].w};
        int sl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sl[k] = -1;
            if (fl[k]) { sl[k] = at; tw.list[at] = (int)(gi * 4 + k); ++at; }
        }
        *reinterpret_cast<int4 *>(tw.slot + gi * 4) = make_int4(sl[0], sl[1], sl[2], sl[3]);
        if (cnt[g]) reinterpret_cast<int4 *>(tw.flags)[gi] = make_int4(0, 0, 0, 0);
    }
}
// Groups per thread inside the fused reduce launches: 8 for a large vocabulary (V = 1M: 123 instead of 977
// atomics per tower on one counter, reduce launch 30 -> 13 us), 1 otherwise (at V = 50k seven workgroups of
// eight loads each were 0.9 us SLOWER than 49 of one).
constexpr int COMPACT_BIG_G = 8;
constexpr int64_t COMPACT_BIG_V = 262144;
static inline int compact_groups(int64_t V) { return V > COMPACT_BIG_V ? COMPACT_BIG_G : 1; }
template <int NW>
__device__ __forceinline__ void token_compact_auto(const TokenTower &tw, int64_t V, int blk) {
    if (V > COMPACT_BIG_V) token_compact_block<NW, COMPACT_BIG_G>(tw, V, blk);   // uniform
    else token_compact_block<NW, 1>(tw, V, blk);
}

static inline TokenArgs make_token_args(int64_t V, const ProjTower *tw, int ntower, int64_t N, int T) {
    TokenArgs a;
    for (int k = 0; k < MAX_TOWERS; ++k) {
        const ProjTower &s = tw[k < ntower ? k : 0];
        a.t[k].idx = s.idx; a.t[k].flags = s.flags; a.t[k].slot = s.slot; a.t[k].list = s.list; a.t[k].count = s.count;
    }
    a.N = N; a.V = V; a.T = T; a.ntower = ntower;
    return a;
}

}  // namespace r4r
