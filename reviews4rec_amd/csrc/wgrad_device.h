// Argmax-sparse conv weight gradient: per-workgroup body, shared by the stand-alone kernel
// (textcnn.hip) and the fused DeepCoNN backward launch (engine.hip).
#pragma once
#include "textcnn.h"

namespace r4r {

typedef float wg_f32x4 __attribute__((ext_vector_type(4)));

constexpr int WG_THREADS = 256;

struct WgradArgs {
    WgradTower t[MAX_TOWERS];
    const float *table;
    int64_t N;
    int T, E, F, per_split, nsplit;
    int64_t table_bytes = 0;           // bytes of `table` (0: unknown -- wgrad_block keeps its per-row loads)
};

constexpr int WG_CHUNK = 16;       // documents resolved per phase-1 round
constexpr int WG_FLIGHT = 8;       // rows in flight per column slot (4 / 8 / 16, at 8 / 6 / 4 waves per SIMD: all within 1 % at cfg3)

// One pass over the split's documents for the 2 x 256 float4 columns from v0 on of the [3][E] window (each thread owns
// up to 2 of them).  FIRST: this pass also sums the bias gradient.
template <bool FIRST>
__device__ __forceinline__ void wgrad_block_pass(const WgradArgs &a, int f, int s, int tower, int v0,
                                                 long (&s_off)[WG_CHUNK][3], float (&s_g)[WG_CHUNK]) {
    const WgradTower &tw = a.t[tower];
    const float *__restrict__ table = a.table;
    const int64_t *__restrict__ idx = tw.idx;
    const float *__restrict__ gp = tw.g_pooled;
    const int *__restrict__ argmax = tw.argmax;
    const int T = a.T, E = a.E, F = a.F;
    const int64_t n0 = (int64_t)s * a.per_split;
    const int64_t n1 = min(a.N, n0 + (int64_t)a.per_split);
    const int nvec = 3 * E / 4;
    const int tid = threadIdx.x;
    wg_f32x4 acc[2] = {(wg_f32x4){0.f, 0.f, 0.f, 0.f}, (wg_f32x4){0.f, 0.f, 0.f, 0.f}};
    int vj[2], ve[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int v = v0 + tid + k * WG_THREADS;
        vj[k] = (v * 4) / E;
        ve[k] = v * 4 - vj[k] * E;
    }
    float sb = 0.f;
    // The rows come through a buffer resource over the table: a slot without a contribution asks for an offset past the
    // end, which the hardware answers with zeros WITHOUT a memory access -- so all 16 rows of a round are requested
    // back to back with no branch between them (as `if (off >= 0) acc += g * row` hipcc had put a branch and a full
    // vmcnt(0) around every single row: sixteen dependent round trips per round, the whole launch), and the skipped
    // rows stay free (requesting them for real, as wgrad_block_packed does, cost +4 us at this width).
    // (tables of 4 GB and more keep the per-row form: 32-bit buffer offsets)
    const bool buf_ok = a.table_bytes > 0 && a.table_bytes < (1ll << 32);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(table), 0, buf_ok ? (int)(unsigned)a.table_bytes : 0, 0x00020000);
    for (int64_t c0 = n0; c0 < n1; c0 += WG_CHUNK) {
        const int nd = (int)min((int64_t)WG_CHUNK, n1 - c0);
        __syncthreads();
        if (tid < nd * 3) {
            const int d = tid / 3, j = tid - d * 3;
            const int64_t n = c0 + d;
            const int p = argmax[n * F + f];
            const float gv = gp[n * F + f];                 // (with the argmax, not behind it)
            const int t = p - 2 + j;
            long off = -1;
            if (p >= 0 && t >= 0 && t < T) off = (long)idx[n * T + t] * E;
            s_off[d][j] = off;
            if (j == 0) s_g[d] = (p >= 0) ? gv : 0.f;
        }
        __syncthreads();
        if (buf_ok) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (v0 + tid + k * WG_THREADS < nvec) {
#pragma unroll
                    for (int d0 = 0; d0 < WG_CHUNK; d0 += WG_FLIGHT) {
                        wg_f32x4 row[WG_FLIGHT];
                        bool on[WG_FLIGHT];
#pragma unroll
                        for (int u = 0; u < WG_FLIGHT; ++u) {
                            const int d = d0 + u;
                            const long off = d < nd ? s_off[d][vj[k]] : -1;
                            on[u] = off >= 0;
                            const unsigned bo = on[u] ? (unsigned)((off + ve[k]) * 4) : 0xfffffff0u;
                            row[u] = __builtin_bit_cast(wg_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)bo, 0, 0));
                        }
#pragma unroll
                        for (int u = 0; u < WG_FLIGHT; ++u) {  // document order, the same additions as the per-row form
                            const wg_f32x4 sum = acc[k] + s_g[d0 + u < nd ? d0 + u : 0] * row[u];
                            acc[k] = on[u] ? sum : acc[k];
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (v0 + tid + k * WG_THREADS < nvec) {
#pragma unroll 4
                    for (int d = 0; d < nd; ++d) {
                        const long off = s_off[d][vj[k]];
                        if (off >= 0) acc[k] += s_g[d] * *reinterpret_cast<const wg_f32x4 *>(table + off + ve[k]);
                    }
                }
            }
        }
        if (FIRST && tid == 0)
            for (int d = 0; d < nd; ++d) sb += s_g[d];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int v = v0 + tid + k * WG_THREADS;
        if (v < nvec) *reinterpret_cast<wg_f32x4 *>(tw.part_w + ((size_t)s * F + f) * 3 * E + v * 4) = acc[k];
    }
    if (FIRST && tid == 0) tw.part_b[(size_t)s * F + f] = sb;
}

__device__ __forceinline__ void wgrad_block(const WgradArgs &a, int f, int s, int tower) {
    // Phase 1 resolves, for a chunk of documents at once, the dependent chain
    // argmax -> position -> token id -> table row offset (one lane per (document, tap), so
    // the chain's latency is paid once per chunk, not once per document); phase 2 streams
    // the resolved rows with independent float4 loads.
    // One pass takes up to 3 E / 4 = 512 float4 columns (word_embed_size <= 680: every configuration of BASELINE.json, the
    // code path and the code of rounds 1-4); wider windows (the reference puts no bound on word_embed_size,
    // hyper_params.py:64) take another pass per 512 columns, repeating the resolve phase instead of costing every launch
    // registers.  Per column the additions are the same, in the same document order, whatever the pass.
    __shared__ long s_off[WG_CHUNK][3];     // table row offset in floats, -1 = no contribution
    __shared__ float s_g[WG_CHUNK];
    wgrad_block_pass<true>(a, f, s, tower, 0, s_off, s_g);
    const int nvec = 3 * a.E / 4;
    for (int v0 = 2 * WG_THREADS; v0 < nvec; v0 += 2 * WG_THREADS) wgrad_block_pass<false>(a, f, s, tower, v0, s_off, s_g);
}


// Narrow windows (3E/4 <= 64 float4, i.e. E <= 85: the reference's default word_embed_size 64):
// one WAVE per filter, four filters per workgroup -- `wgrad_block` would leave three waves of
// four idle in its streaming phase.  Same per-(filter, split) summation order, so the same bits.
// grid.x = ceil(F / 4) for this form.
constexpr int WG_SUPER = 96;  // documents whose (argmax -> token -> row offset) chains are resolved TOGETHER
static_assert(WG_SUPER % WG_CHUNK == 0, "whole streaming rounds per resolved super-chunk");
__device__ __forceinline__ void wgrad_block_packed(const WgradArgs &a, int fgroup, int s, int tower) {
    __shared__ long p_off[4][WG_SUPER][3];
    __shared__ float p_g[4][WG_SUPER];
    const WgradTower &tw = a.t[tower];
    const float *__restrict__ table = a.table;
    const int64_t *__restrict__ idx = tw.idx;
    const float *__restrict__ gp = tw.g_pooled;
    const int *__restrict__ argmax = tw.argmax;
    const int T = a.T, E = a.E, F = a.F;
    const int64_t n0 = (int64_t)s * a.per_split;
    const int64_t n1 = min(a.N, n0 + (int64_t)a.per_split);
    const int nvec = 3 * E / 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int f = fgroup * 4 + wave;
    const bool live = f < F;
    wg_f32x4 acc = (wg_f32x4){0.f, 0.f, 0.f, 0.f};
    const int vj = (lane * 4) / E, ve = lane * 4 - vj * E;
    float sb = 0.f;
    // A split is 80 documents at NARRE's 1,280 reviews per tower: resolved 16 at a time (argmax -> token id -> rows:
    // three dependent round trips per round, five rounds) the launch was latency, not bandwidth -- 18 us where the
    // rows themselves are 197 MB out of L2.  Now the chains of up to WG_SUPER documents are resolved together (every
    // argmax / gradient read of the super-chunk in one round trip, every token id in the next), then the rows stream
    // 16 per round trip as before: seven round trips per split instead of fifteen.  Same per-(filter, split) order
    // of additions: the same bits.
    constexpr int NRES = (WG_SUPER * 3 + 63) / 64;          // (document, tap) items per lane
    for (int64_t c0 = n0; c0 < n1; c0 += WG_SUPER) {
        const int nd = (int)min((int64_t)WG_SUPER, n1 - c0);
        __syncthreads();
        if (live) {
            int pv[NRES];
            float gv[NRES];
#pragma unroll
            for (int k = 0; k < NRES; ++k) {                // every argmax and gradient of the super-chunk: one round trip
                const int i = lane + 64 * k, d = i / 3;
                const int64_t n = c0 + (d < nd ? d : 0);
                pv[k] = argmax[n * F + f];
                gv[k] = gp[n * F + f];
                if (d >= nd) pv[k] = -1;
            }
            long off[NRES], tokv[NRES];
            bool onv[NRES];
#pragma unroll
            for (int k = 0; k < NRES; ++k) {                // every token id: the next
                const int i = lane + 64 * k, d = i / 3, j = i - d * 3;
                const int64_t n = c0 + (d < nd ? d : 0);
                const int t = pv[k] - 2 + j;
                onv[k] = pv[k] >= 0 && t >= 0 && t < T;
                tokv[k] = idx[n * T + (onv[k] ? t : 0)];
            }
            // (all of them requested before the first is used: hipcc had sunk each load into the branch of its `on`, with a
            // full vmcnt(0) behind it -- five dependent round trips per super-chunk instead of one)
#pragma unroll
            for (int k = 0; k < NRES; ++k) asm volatile("" : "+v"(tokv[k]));
#pragma unroll
            for (int k = 0; k < NRES; ++k) off[k] = onv[k] ? tokv[k] * E : -1;
#pragma unroll
            for (int k = 0; k < NRES; ++k) {
                const int i = lane + 64 * k, d = i / 3, j = i - d * 3;
                if (d < WG_SUPER) {
                    p_off[wave][d][j] = off[k];
                    if (j == 0) p_g[wave][d] = pv[k] >= 0 ? gv[k] : 0.f;
                }
            }
        }
        __syncthreads();
        // all WG_CHUNK rows of a round are requested before the first is used (unconditional loads -- a
        // slot without a contribution re-reads row 0 -- and selects instead of branches, in document
        // order: one memory round trip per round instead of four, same bits): NARRE's backward launch
        // 46.9 -> 34.6 us
        if (live && lane < nvec) {
            for (int r0 = 0; r0 < nd; r0 += WG_CHUNK) {
                wg_f32x4 v[WG_CHUNK];
#pragma unroll
                for (int d = 0; d < WG_CHUNK; ++d) {
                    const long off = p_off[wave][r0 + d][vj];
                    v[d] = *reinterpret_cast<const wg_f32x4 *>(table + (off < 0 ? 0 : off) + ve);
                }
#pragma unroll
                for (int d = 0; d < WG_CHUNK; ++d) {
                    const wg_f32x4 sum = acc + p_g[wave][r0 + d] * v[d];
                    acc = p_off[wave][r0 + d][vj] >= 0 ? sum : acc;
                }
            }
        }
        if (live && lane == 0)
            for (int d = 0; d < nd; ++d) sb += p_g[wave][d];
    }
    if (live && lane < nvec) *reinterpret_cast<wg_f32x4 *>(tw.part_w + ((size_t)s * F + f) * 3 * E + lane * 4) = acc;
    if (live && lane == 0) tw.part_b[(size_t)s * F + f] = sb;
}

// Second stage: element i of tower `tw`'s [F*3E weight | F bias] gradient = sum of the nsplit
// partials in a fixed order (deterministic).
constexpr int WG_MAX_SPLITS = 16;      // textcnn_wgrad_splits never returns more

// Element i of the concatenated [F*3E | F] gradient: every partial is requested before the first is added (a
// loop with a run-time trip count made them dependent round trips), summed in split order like before.  Writes
// the element, returns it and its address (nullptr past the end).
__device__ __forceinline__ float wgrad_reduce_elem(const WgradArgs &a, int tower, int i, float *&dst) {
    const WgradTower &tw = a.t[tower];
    const int nw = a.F * 3 * a.E;
    const float *src;
    size_t stride;
    if (i < nw) { src = tw.part_w + i; stride = (size_t)nw; dst = tw.d_w + i; }
    else if (i < nw + a.F) { src = tw.part_b + (i - nw); stride = (size_t)a.F; dst = tw.d_b + (i - nw); }
    else { dst = nullptr; return 0.f; }
    float v[WG_MAX_SPLITS];
#pragma unroll
    for (int k = 0; k < WG_MAX_SPLITS; ++k) v[k] = src[(size_t)(k < a.nsplit ? k : 0) * stride];   // unconditional loads
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < WG_MAX_SPLITS; ++k)
        if (k < a.nsplit) s += v[k];                        // uniform
    *dst = s;
    return s;
}

__device__ __forceinline__ void wgrad_reduce_block(const WgradArgs &a, int tower, int blk) {
    float *dst;
    (void)wgrad_reduce_elem(a, tower, blk * (int)blockDim.x + (int)threadIdx.x, dst);
}

}  // namespace r4r
