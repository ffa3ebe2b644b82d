// Launch roles shared by the fused review-model steps (narre_engine.hip, deepconnpp_engine.hip,
// transnet_engine.hip): the column sums that turn per-rating head-gradient rows into the flat
// gradient, the backward launch (argmax-sparse conv wgrad of the towers + column sums + the next
// batch's token marks + NARRE's ID-table Adam), the reduce launch (wgrad partials -> gradient,
// Adam on the dense parameters, next batch's token compaction).  Header-only: every step file
// instantiates its own copies.
#pragma once
#include <stdlib.h>

#include "adam_device.h"
#include "rows_device.h"
#include "textcnn.h"
#include "tokens_device.h"
#include "wgrad_device.h"

namespace r4r {

constexpr int NF = 100;                // conv filters (common_pytorch_models.py:11)
constexpr int NR_MAX_L = 32, NR_MAX_R = 32;        // hard limits; the kernels are instantiated for <= 16 and <= 32

// ---- 4: column sums of the [B, NHP] matrix (fixed order) -> flat gradient; + running SE
struct ColSum {
    const float *part, *se;
    float *flat_g, *sse_accum;
    int64_t B;
    int nhp, col0_lo, col0_n, col1_lo;
    const float *aux = nullptr;        // TransNet: [B, 3] per-rating (target prediction, its SE, ||s_ir - t_ir||^2)
    float inv_denom = 0.f;             //           sse_accum[1], [2] += this batch's MEAN of aux columns 1, 2
};
constexpr int CS_ROWS = 16, CS_COLS = 16;
__device__ __forceinline__ void colsum_block(const ColSum &c, int blk) {
    __shared__ float red[CS_ROWS][CS_COLS];
    const int ox = threadIdx.x & (CS_COLS - 1), rg = threadIdx.x / CS_COLS;
    const int col = blk * CS_COLS + ox;                     // column nhp = the SE accumulator
    float s = 0.f;
    if (col < c.nhp) for (int64_t b = rg; b < c.B; b += CS_ROWS) s += c.part[(size_t)b * c.nhp + col];
    else if (col == c.nhp) for (int64_t b = rg; b < c.B; b += CS_ROWS) s += c.se[b];
    else if (c.aux && col <= c.nhp + 2) for (int64_t b = rg; b < c.B; b += CS_ROWS) s += c.aux[b * 3 + (col - c.nhp)];
    red[rg][ox] = s;
    __syncthreads();
    if (rg == 0 && col <= c.nhp + (c.aux ? 2 : 0)) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < CS_ROWS; ++r) t += red[r][ox];
        if (col == c.nhp) { if (c.sse_accum) c.sse_accum[0] += t; }
        else if (col > c.nhp) { if (c.sse_accum) c.sse_accum[col - c.nhp] += t * c.inv_denom; }
        else c.flat_g[col < c.col0_n ? c.col0_lo + col : c.col1_lo + (col - c.col0_n)] = t;
    }
}

// ---- Adam over the ID tables and bias vectors (a role of the backward launch: it depends only on
// the head kernel, and its ~18 us of latency-bound work hides behind the weight gradient), two
// kinds of workgroup
//   sweep workgroups  stream every element; a row NO rating touched (tag != this step) gets the
//                     gradient-zero update, a touched row is left alone
//   entry waves       one wave per compact entry k.  It scans the entry ids once (64 lanes wide);
//                     if k is the first entry of its row it sums that row's entries in ascending
//                     order (deterministic) and applies the update to the table row (all
//                     entries) and to the row's bias element (the self entries, the first B)
// (scalar kernel arguments only: an argument array indexed by the workgroup's slot is copied to
// scratch by hipcc, mf_engine.hip)
constexpr int NROW_CHUNK = 2048, NROW_THREADS = 256;
constexpr int NROW_MAX_ENTRIES = 4096;   // B (1 + R): an entry wave keeps every entry id in registers
struct RowSweep {
    float *p0, *p1, *p2, *p3, *m0, *m1, *m2, *m3, *v0, *v1, *v2, *v3;   // user table, item table, user bias, item bias
    int64_t n0, n1, n2, n3;
    int cb1, cb2, cb3, cb_entries;
    const int64_t *gid0, *gid1;        // entry ids of the user / item table
    const float *grow0, *grow1;        // entry rows [entries, L]
    const float *g;                    // [B]: bias entries are the first B table entries (the self rows)
    const int *tag0, *tag1;
    int64_t entries, B;
    int L, now;
    AdamScalars s;
};
// MW: 64-bit words of a lane's hit mask (entries <= 4096 MW: 1 in the fused single-process launch,
// 4 in the stand-alone data-parallel launch); `sid`: LDS for the entry ids (entries ints).
template <int ML, int MW>
__device__ __forceinline__ void narre_rows_block(const RowSweep &w, int bx, int *sid) {
    if (bx >= w.cb_entries) {
        // ---- entry waves: 4 per workgroup, all of one table (user table's groups first)
        const int lane = threadIdx.x & 63;
        const int groups = (int)((w.entries + 3) / 4);
        int gi = bx - w.cb_entries;
        const int t = gi >= groups;
        if (t) gi -= groups;
        const int64_t *ids = t ? w.gid1 : w.gid0;
        const float *rows = t ? w.grow1 : w.grow0;
        for (int64_t j = threadIdx.x; j < w.entries; j += NROW_THREADS) sid[j] = (int)ids[j];   // one round trip
        __syncthreads();
        const int64_t k = (int64_t)gi * 4 + (threadIdx.x >> 6);
        if (k >= w.entries) return;                         // whole wave
        const int row = sid[k];
        if (row < 0) return;                                // a padded entry (gathered ragged shards): whole wave
        const int L = w.L;
        const int nch = (int)((w.entries + 63) / 64);
        // phase 1: is k the first entry of its row?  (scan of the ids in LDS, 64 at a time)
        bool first = true;
        for (int c = 0; c < nch && first; ++c) {
            const int j = c * 64 + lane;
            const unsigned long long mask = __ballot(j < w.entries && sid[j] == row);
            if (mask && (int64_t)c * 64 + (__ffsll((long long)mask) - 1) < k) first = false;
            if (mask && (int64_t)c * 64 + 63 >= k) break;   // reached k's own chunk: nothing earlier matched
        }
        if (!first) return;                                 // an earlier entry owns this row (uniform)
        // phase 2 (one wave per DISTINCT row): every lane adds up the rows of ITS hits, chunk by
        // chunk in ascending order, then one fixed butterfly per column combines the 64 lanes -- a
        // fixed order, so the result is deterministic (a butterfly per chunk made a row with
        // hundreds of entries a 70 us chain of cross-lane permutes)
        float rv[ML];
#pragma unroll
        for (int col = 0; col < ML; ++col) rv[col] = 0.f;
        float gv = 0.f;
        unsigned long long mine[MW];                        // bit c of word c / 64: entry c*64 + lane is a hit (nch <= 64 MW)
#pragma unroll
        for (int q = 0; q < MW; ++q) mine[q] = 0;
        for (int c = (int)(k / 64); c < nch; ++c) {         // (no hit before k's chunk: k is the first)
            const int64_t j = (int64_t)c * 64 + lane;
            if (j < w.entries && sid[j] == row) {
#pragma unroll
                for (int q = 0; q < MW; ++q)
                    if ((c >> 6) == q) mine[q] |= 1ull << (c & 63);
            }
        }
        auto any = [&]() { unsigned long long o = 0;
#pragma unroll
            for (int q = 0; q < MW; ++q) o |= mine[q];
            return o != 0; };
        auto pop = [&]() -> int {                           // the lane's lowest remaining hit (ascending order), -1 if none
#pragma unroll
            for (int q = 0; q < MW; ++q)
                if (mine[q]) { const int c = __ffsll((long long)mine[q]) - 1; mine[q] &= mine[q] - 1; return q * 64 + c; }
            return -1;
        };
        while (__ballot(any())) {                           // four of a lane's hits per round, their loads together
            int cs[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) cs[u] = pop();
            float tmp[4][ML], tg[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t j = (int64_t)(cs[u] < 0 ? 0 : cs[u]) * 64 + lane;
#pragma unroll
                for (int col = 0; col < ML; ++col)
                    tmp[u][col] = (cs[u] >= 0 && col < L) ? rows[j * L + col] : 0.f;
                tg[u] = (cs[u] >= 0 && j < w.B) ? w.g[j] : 0.f;         // only the self entries carry a bias gradient
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {                   // ascending entry order within the lane
#pragma unroll
                for (int col = 0; col < ML; ++col) rv[col] += tmp[u][col];
                gv += tg[u];
            }
        }
        float acc = 0.f;                                    // lane < L: column `lane` of the table row
#pragma unroll
        for (int col = 0; col < ML; ++col) {
            if (col < L) {                                  // uniform
                const float sum = wave_sum(rv[col]);
                if (lane == col) acc = sum;
            }
        }
        const float accb = wave_sum(gv);
        if (lane < L) {
            float *p = (t ? w.p1 : w.p0) + (int64_t)row * L + lane, *m = (t ? w.m1 : w.m0) + (int64_t)row * L + lane,
                  *v = (t ? w.v1 : w.v0) + (int64_t)row * L + lane;
            float P = *p, M = *m, V = *v;
            adam_elem(P, acc, M, V, w.s);
            *p = P; *m = M; *v = V;
        }
        if (lane == 0) {                                    // the row's bias element (gradient zero if no self entry)
            float *p = (t ? w.p3 : w.p2) + row, *m = (t ? w.m3 : w.m2) + row, *v = (t ? w.v3 : w.v2) + row;
            float P = *p, M = *m, V = *v;
            adam_elem(P, accb, M, V, w.s);
            *p = P; *m = M; *v = V;
        }
        return;
    }
    // ---- sweep workgroups
    const int t = (bx >= w.cb1) + (bx >= w.cb2) + (bx >= w.cb3);
    float *bp = w.p0, *bm = w.m0, *bv = w.v0;
    int64_t numel = w.n0;
    int cb = 0;
    if (t == 1) { bp = w.p1; bm = w.m1; bv = w.v1; numel = w.n1; cb = w.cb1; }
    else if (t == 2) { bp = w.p2; bm = w.m2; bv = w.v2; numel = w.n2; cb = w.cb2; }
    else if (t == 3) { bp = w.p3; bm = w.m3; bv = w.v3; numel = w.n3; cb = w.cb3; }
    const unsigned W = t < 2 ? (unsigned)w.L : 1u;
    const int *tag = (t == 0 || t == 2) ? w.tag0 : w.tag1;
    const int64_t start = (int64_t)(bx - cb) * NROW_CHUNK;
    int64_t cnt = numel - start;
    if (cnt > NROW_CHUNK) cnt = NROW_CHUNK;
    const int64_t row0 = start / W;                         // one 64-bit division per workgroup
    const unsigned col0 = (unsigned)(start - row0 * W);
    constexpr int PER = NROW_CHUNK / NROW_THREADS;
    float P[PER], M[PER], V[PER];
    int T[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {                         // all loads of the thread before any use
        const int64_t i = threadIdx.x + (int64_t)u * NROW_THREADS;
        const int64_t ii = i < cnt ? i : 0;
        const int64_t row = row0 + (col0 + (unsigned)ii) / W;
        P[u] = bp[start + ii]; M[u] = bm[start + ii]; V[u] = bv[start + ii];
        T[u] = tag[row];
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int64_t i = threadIdx.x + (int64_t)u * NROW_THREADS;
        if (i < cnt && T[u] != w.now) {
            adam_elem(P[u], 0.f, M[u], V[u], w.s);
            bp[start + i] = P[u]; bm[start + i] = M[u]; bv[start + i] = V[u];
        }
    }
}

// ML: 0 = no ID-table role (DeepCoNN++), else the rows role's template argument.  z-slices: the ID
// tables (ML > 0), the `ntower` towers' weight gradients, the head-parameter column sums, the next
// batch's token marks (if announced).
template <int ML>
__global__ __launch_bounds__(WG_THREADS) void narre_backward_kernel(WgradArgs w, ColSum c, int cs_blocks, TokenArgs nx,
                                                                    int packed, RowSweep rows, int row_blocks, int ntower) {
    const int blk0 = blockIdx.y * gridDim.x + blockIdx.x, nblk = gridDim.x * gridDim.y;
    // the ID-table role is dispatched FIRST (slice 0 when present): its owners' dependent chains are
    // the longest thing in the launch, the weight-gradient workgroups fill in around them
    const int z = (int)blockIdx.z - (ML > 0 ? 1 : 0);
    if (z < 0) {
        if constexpr (ML > 0) {
            __shared__ int rows_sid[NROW_MAX_ENTRIES];
            // entry workgroups first (the owner of a popular row is the longest), then the sweep
            for (int blk = blk0; blk < row_blocks; blk += nblk) {
                const int ne = row_blocks - rows.cb_entries;
                narre_rows_block<ML, 1>(rows, blk < ne ? rows.cb_entries + blk : blk - ne, rows_sid);
                __syncthreads();
            }
        }
    } else if (z < ntower) {
        if (packed) wgrad_block_packed(w, blockIdx.x, blockIdx.y, z);   // grid.x = ceil(F / 4)
        else wgrad_block(w, blockIdx.x, blockIdx.y, z);
    } else if (z == ntower) {
        for (int blk = blk0; blk < cs_blocks; blk += nblk) {
            colsum_block(c, blk);
            __syncthreads();
        }
    } else {
        token_mark_block(nx, blk0, nblk, WG_THREADS);
    }
}

// The ID-table role as a launch of its own (data parallel: the entries of ALL ranks, gathered): up to
// 16,384 entries per table, their ids in dynamic LDS.
constexpr int NROW_DP_WORDS = 4, NROW_DP_MAX_ENTRIES = 64 * 64 * NROW_DP_WORDS;
template <int ML>
static __global__ __launch_bounds__(NROW_THREADS) void narre_rows_kernel(RowSweep w) {
    extern __shared__ int rows_sid_dyn[];
    narre_rows_block<ML, NROW_DP_WORDS>(w, (int)blockIdx.x, rows_sid_dyn);
}

// ---- 5: wgrad partial reduce + Adam on the dense parameters + next batch's compaction
constexpr int NRED_THREADS = 256;
struct DenseAdam {
    float *p, *m, *v;
    const float *g;
    int64_t lo0, hi0, lo1, hi1;
    AdamScalars s;
    int on;
};
static __global__ __launch_bounds__(NRED_THREADS) void narre_reduce_kernel(WgradArgs w, int red_blocks, int comp_blocks,
                                                                    TokenArgs nx, DenseAdam opt) {
    const int bx = blockIdx.x;
    if (bx < red_blocks) {
        wgrad_reduce_block(w, blockIdx.y, bx);
        if (opt.on) {
            const WgradTower &tw = w.t[blockIdx.y];
            const int nw = w.F * 3 * w.E;
            const int i = bx * NRED_THREADS + threadIdx.x;
            const float *gp = i < nw ? tw.d_w + i : (i < nw + w.F ? tw.d_b + (i - nw) : nullptr);
            if (gp) {
                const int64_t o = gp - opt.g;
                float P = opt.p[o], M = opt.m[o], V = opt.v[o];
                adam_elem(P, *gp, M, V, opt.s);
                opt.p[o] = P; opt.m[o] = M; opt.v[o] = V;
            }
        }
    } else if (bx < red_blocks + comp_blocks) {
        token_compact_block<NRED_THREADS / 64, COMPACT_G>(nx.t[blockIdx.y], nx.V, bx - red_blocks);
    } else {
        const int64_t base = blockIdx.y == 0 ? opt.lo0 : opt.lo1, end = blockIdx.y == 0 ? opt.hi0 : (blockIdx.y == 1 ? opt.hi1 : opt.lo1);
        const int64_t o = base + (int64_t)(bx - red_blocks - comp_blocks) * NRED_THREADS + threadIdx.x;
        if (o < end) {
            float P = opt.p[o], M = opt.m[o], V = opt.v[o];
            adam_elem(P, opt.g[o], M, V, opt.s);
            opt.p[o] = P; opt.m[o] = M; opt.v[o] = V;
        }
    }
}

}  // namespace r4r
