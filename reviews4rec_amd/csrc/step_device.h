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
#include "trace_device.h"
#include "wgrad_device.h"

namespace r4r {

constexpr int NF = 100;                // conv filters (common_pytorch_models.py:11)
constexpr int NR_MAX_L = 32, NR_MAX_R = 32;        // hard limits of the fused ID-table ROW role (a row in registers); instantiated for <= 16 and <= 32
constexpr int HEAD_MAX_L = 64;                       // DeepCoNN++'s and TransNet's heads (LDS arrays by the template's ML; their ID parts go through the MF sweeps)

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
    // (the running sums are read now, not after the reduction: the launch's last block was a round trip longer
    // than every other one -- its tail)
    const bool acc_col = rg == 0 && c.sse_accum && col >= c.nhp && col <= c.nhp + (c.aux ? 2 : 0);
    const float prev = acc_col ? c.sse_accum[col - c.nhp] : 0.f;
    // ONE loop for the three kinds of column (a matrix column, the SE vector, an aux column): as three loops the
    // last block's waves ran them one after the other -- 8.3 us where every other block takes 4.6
    const float *src = nullptr;
    size_t stride = 0;
    if (col < c.nhp) { src = c.part + col; stride = (size_t)c.nhp; }
    else if (col == c.nhp) { src = c.se; stride = 1; }
    else if (c.aux && col <= c.nhp + 2) { src = c.aux + (col - c.nhp); stride = 3; }
    float s = 0.f;
    if (src) {
#pragma unroll 8                                             // (eight of a thread's loads in flight: B = 128 is one round)
        for (int64_t b = rg; b < c.B; b += CS_ROWS) s += src[(size_t)b * stride];
    }
    red[rg][ox] = s;
    __syncthreads();
    if (rg == 0 && col <= c.nhp + (c.aux ? 2 : 0)) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < CS_ROWS; ++r) t += red[r][ox];
        if (col == c.nhp) { if (c.sse_accum) c.sse_accum[0] = prev + t; }
        else if (col > c.nhp) { if (c.sse_accum) c.sse_accum[col - c.nhp] = prev + t * c.inv_denom; }
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
    int sweep_elsewhere;               // the sweep workgroups ride in the reduce launch: the backward launch runs the entry waves only
    // BLK form (data parallel, r4r_narre_rows_apply_blocks): the entries live in the ranks' gathered blocks (rows_device.h:
    // mf_block(E_pad, L)) -- entry e is entry e % E_pad of block e / E_pad, `blk_units` 4-byte units apart; gid32_* / grow* / g
    // point into block 0
    const int *gid32_0 = nullptr, *gid32_1 = nullptr;
    int64_t E_pad = 0, blk_units = 0;
};
// offset (4-byte units) of entry e's field of `width` units per entry
template <bool BLK>
__device__ __forceinline__ int64_t nrow_eoff(const RowSweep &w, int64_t e, int width) {
    if constexpr (BLK) {
        const int r = (int)((unsigned)e / (unsigned)w.E_pad);
        return (int64_t)r * w.blk_units + (e - (int64_t)r * w.E_pad) * width;
    } else {
        return e * width;
    }
}
// ---- sweep workgroup `bx` (< cb_entries) of the ID tables / bias vectors: rows no rating touched
__device__ __forceinline__ void narre_sweep_block(const RowSweep &w, int bx) {
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
            adam_elem_fast(P[u], 0.f, M[u], V[u], w.s);
            bp[start + i] = P[u]; bm[start + i] = M[u]; bv[start + i] = V[u];
        }
    }
}

constexpr int NROW_EPW = 2;              // entries per wave of an entry workgroup (16 entries share one load of the ids)
// MW: 64-bit words of a lane's hit mask (entries <= 4096 MW: 1 in the fused single-process launch,
// 4 in the stand-alone data-parallel launch); `sid`: LDS for the entry ids (entries ints).
template <int ML, int MW, bool BLK = false>
__device__ __forceinline__ void narre_rows_block(const RowSweep &w, int bx, int *sid) {
    if (bx >= w.cb_entries) {
        // ---- entry waves: 4 per workgroup, NROW_EPW entries each (interleaved: the owners of popular rows
        // are early entries), all of one table (user table's groups first)
        const int lane = threadIdx.x & 63;
        const int groups = (int)((w.entries + 4 * NROW_EPW - 1) / (4 * NROW_EPW));
        int gi = bx - w.cb_entries;
        const int t = gi >= groups;
        if (t) gi -= groups;
        const int64_t *ids = t ? w.gid1 : w.gid0;
        const float *rows = t ? w.grow1 : w.grow0;
        // eight ids per thread requested together (clamped, unconditional; as a plain loop every iteration was a load, a
        // full wait and an LDS write: six dependent round trips for NARRE's 1,408 entries, in front of everything else)
        constexpr int SB = 8;
        for (int64_t j0 = threadIdx.x; j0 < w.entries; j0 += SB * NROW_THREADS) {
            int idv[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int64_t j = min(j0 + (int64_t)u * NROW_THREADS, w.entries - 1);
                if constexpr (BLK) idv[u] = (t ? w.gid32_1 : w.gid32_0)[nrow_eoff<true>(w, j, 1)];
                else idv[u] = (int)ids[j];
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) asm volatile("" : "+v"(idv[u]));
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int64_t j = j0 + (int64_t)u * NROW_THREADS;
                if (j < w.entries) sid[j] = idv[u];
            }
        }
        __syncthreads();
        const int L = w.L;
        const int nch = (int)((w.entries + 63) / 64);
        float *tp = t ? w.p1 : w.p0, *tm = t ? w.m1 : w.m0, *tv = t ? w.v1 : w.v0;      // the table ...
        float *bp = t ? w.p3 : w.p2, *bm = t ? w.m3 : w.m2, *bv = t ? w.v3 : w.v2;      // ... and its bias vector
        // Two entries at a time: the LDS-only work of both first (is the entry the first of its row? which
        // entries hit the row?), then EVERY global read of the pair in one round trip -- the rows' parameters
        // and moments (their addresses need only the row) next to each lane's first hit -- then the sums.
        // (One entry at a time was two dependent round trips per owner: 6 us each, 30 us for a wave of 4.)
        constexpr int PAIR = 2;
        static_assert(NROW_EPW % PAIR == 0, "entries per wave come in pairs");
        for (int e0 = 0; e0 < NROW_EPW; e0 += PAIR) {
            int rowq[PAIR];
            bool own[PAIR];
            unsigned long long mine[PAIR][MW];              // bit c of word c / 64: entry c*64 + lane is a hit (nch <= 64 MW)
#pragma unroll
            for (int q = 0; q < PAIR; ++q) {
                const int64_t k = (int64_t)gi * 4 * NROW_EPW + (e0 + q) * 4 + (threadIdx.x >> 6);
                own[q] = false;
                rowq[q] = 0;
#pragma unroll
                for (int m = 0; m < MW; ++m) mine[q][m] = 0;
                const int row = k < w.entries ? sid[k] : -1;    // -1: past the end, or a padded entry (gathered ragged shards)
                // phase 1: is k the first entry of its row?  (scan of the ids in LDS, 64 at a time)
                bool first = row >= 0;                      // (everything here is uniform over the wave)
                for (int c = 0; c < nch && first; ++c) {
                    const int jj = c * 64 + lane;
                    const unsigned long long mask = __ballot(jj < w.entries && sid[jj] == row);
                    if (mask && (int64_t)c * 64 + (__ffsll((long long)mask) - 1) < k) first = false;
                    if (mask && (int64_t)c * 64 + 63 >= k) break;   // reached k's own chunk: nothing earlier matched
                }
                if (first) {                                // else an earlier entry owns this row
                    own[q] = true;
                    rowq[q] = row;
                    for (int c = (int)(k / 64); c < nch; ++c) { // (no hit before k's chunk: k is the first)
                        const int64_t jj = (int64_t)c * 64 + lane;
                        if (jj < w.entries && sid[jj] == row) {
#pragma unroll
                            for (int m = 0; m < MW; ++m)
                                if ((c >> 6) == m) mine[q][m] |= 1ull << (c & 63);
                        }
                    }
                }
            }
            auto any = [&](int q) { unsigned long long o = 0;
#pragma unroll
                for (int m = 0; m < MW; ++m) o |= mine[q][m];
                return o != 0; };
            auto pop = [&](int q) -> int {                  // the lane's lowest remaining hit (ascending order), -1 if none
#pragma unroll
                for (int m = 0; m < MW; ++m)
                    if (mine[q][m]) { const int c = __ffsll((long long)mine[q][m]) - 1; mine[q][m] &= mine[q][m] - 1; return m * 64 + c; }
                return -1;
            };
            // one round trip: unconditional loads at clamped addresses (row 0 / entry 0 for a non-owner: unused)
            float P[PAIR], M[PAIR], V[PAIR], Pb[PAIR], Mb[PAIR], Vb[PAIR], h0[PAIR][ML], g0[PAIR];
            int c0[PAIR];
#pragma unroll
            for (int q = 0; q < PAIR; ++q) {
                const int64_t o = (int64_t)rowq[q] * L + (lane < L ? lane : L - 1);
                P[q] = tp[o]; M[q] = tm[o]; V[q] = tv[o];
                Pb[q] = bp[rowq[q]]; Mb[q] = bm[rowq[q]]; Vb[q] = bv[rowq[q]];
                c0[q] = pop(q);
                int64_t jj = (int64_t)(c0[q] < 0 ? 0 : c0[q]) * 64 + lane;
                if (jj >= w.entries) jj = w.entries - 1;
                const int64_t ro = nrow_eoff<BLK>(w, jj, L);
#pragma unroll
                for (int col = 0; col < ML; ++col) h0[q][col] = rows[ro + (col < L ? col : L - 1)];
                g0[q] = w.g[nrow_eoff<BLK>(w, jj < w.B ? jj : 0, 1)];
            }
#pragma unroll
            for (int q = 0; q < PAIR; ++q) if (own[q]) {    // uniform
                // phase 2 (one wave per DISTINCT row): every lane adds up the rows of ITS hits, chunk by
                // chunk in ascending order, then one fixed butterfly per column combines the 64 lanes -- a
                // fixed order, so the result is deterministic (a butterfly per chunk made a row with
                // hundreds of entries a 70 us chain of cross-lane permutes)
                float rv[ML];
                const int64_t j0 = (int64_t)(c0[q] < 0 ? 0 : c0[q]) * 64 + lane;
#pragma unroll
                for (int col = 0; col < ML; ++col) rv[col] = 0.f + ((c0[q] >= 0 && col < L) ? h0[q][col] : 0.f);
                float gv = 0.f + ((c0[q] >= 0 && j0 < w.B) ? g0[q] : 0.f);     // only the self entries carry a bias gradient
                while (__ballot(any(q))) {                  // a lane's further hits, two per round, their loads together
                    constexpr int HR = 2;         // (registers: the pair's first hits are still live)
                    int cs[HR];
#pragma unroll
                    for (int u = 0; u < HR; ++u) cs[u] = pop(q);
                    float tmp[HR][ML], tg[HR];
#pragma unroll
                    for (int u = 0; u < HR; ++u) {
                        const int64_t jj = (int64_t)(cs[u] < 0 ? 0 : cs[u]) * 64 + lane;
                        const int64_t ro = cs[u] >= 0 ? nrow_eoff<BLK>(w, jj, L) : 0;
#pragma unroll
                        for (int col = 0; col < ML; ++col)
                            tmp[u][col] = (cs[u] >= 0 && col < L) ? rows[ro + col] : 0.f;
                        tg[u] = (cs[u] >= 0 && jj < w.B) ? w.g[nrow_eoff<BLK>(w, jj, 1)] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < HR; ++u) {          // ascending entry order within the lane
#pragma unroll
                        for (int col = 0; col < ML; ++col) rv[col] += tmp[u][col];
                        gv += tg[u];
                    }
                }
                float acc = 0.f;                            // lane < L: column `lane` of the table row
#pragma unroll
                for (int col = 0; col < ML; ++col) {
                    if (col < L) {                          // uniform
                        const float sum = wave_sum(rv[col]);
                        if (lane == col) acc = sum;
                    }
                }
                const float accb = wave_sum(gv);
                const int row = rowq[q];
                if (lane < L) {
                    const int64_t o = (int64_t)row * L + lane;
                    float p1 = P[q], m1 = M[q], v1 = V[q];
                    adam_elem_fast(p1, acc, m1, v1, w.s);
                    tp[o] = p1; tm[o] = m1; tv[o] = v1;
                }
                if (lane == 0) {                            // the row's bias element (gradient zero if no self entry)
                    float p1 = Pb[q], m1 = Mb[q], v1 = Vb[q];
                    adam_elem_fast(p1, accb, m1, v1, w.s);
                    bp[row] = p1; bm[row] = m1; bv[row] = v1;
                }
            }
        }
        return;
    }
    narre_sweep_block(w, bx);
}

// ML: 0 = no ID-table role (DeepCoNN++), else the rows role's template argument.  z-slices: the ID
// tables (ML > 0), the `ntower` towers' weight gradients, the head-parameter column sums, the next
// batch's token marks (if announced).
// z-slices the head-parameter column sums take: one block per workgroup
__host__ __device__ inline int backward_cs_slices(int cs_blocks, int wgs_per_slice) {
    return (cs_blocks + wgs_per_slice - 1) / wgs_per_slice;
}

// WIDE: the generic wgrad (3E/4 > 64 float4: wgrad_block) instead of the packed one-wave-per-filter form.  Without
// an ID-table role that variant fits 64 VGPRs -- 8 waves per SIMD, every workgroup of a DeepCoNN++ launch resident
// (13.7 -> 11.0 us); the packed form spills at that cap (its workgroups went 4.6 -> 7.2 us) and keeps 4.
// ONE argument struct, read through kernel_args (common.h): each role loads its own fields inside its own branch --
// as eight by-value arguments all ~90 scalars were loaded up front and 160 of NARRE's lived spilled in vector lanes
// (775 v_readlane / v_writelane; tools/isa_scan.py).
struct BackwardArgs {
    WgradArgs w;
    ColSum c;
    int cs_blocks;
    TokenArgs nx;
    int packed;
    RowSweep rows;
    int row_blocks, ntower;
};
template <int ML, bool WIDE = false>
__global__ __launch_bounds__(WG_THREADS, ML > 16 ? 2 : (ML == 0 && WIDE ? 8 : 4)) void narre_backward_kernel(BackwardArgs by_value) {
    // (without the ID-table role the arguments fit the scalar registers -- 15-20 spills -- and loading them up front is
    // 0.3 us faster than at their uses: same-box A/B, profiles/r06_ab_kernargs.txt)
    const BackwardArgs &A = ML > 0 ? kernel_args<BackwardArgs>() : by_value;
    const WgradArgs &w = A.w;
    const ColSum &c = A.c;
    const TokenArgs &nx = A.nx;
    const RowSweep &rows = A.rows;
    const int cs_blocks = A.cs_blocks, packed = A.packed, row_blocks = A.row_blocks, ntower = A.ntower;
    const int blk0 = blockIdx.y * gridDim.x + blockIdx.x, nblk = gridDim.x * gridDim.y;
    BWD_STAMP(0, wall_clock64())
    BWD_STAMP(2, (unsigned long long)blockIdx.z + 1)
    // the ID-table role is dispatched FIRST (slice 0 when present): its owners' dependent chains are
    // the longest thing in the launch, the weight-gradient workgroups fill in around them
    const int z = (int)blockIdx.z - (ML > 0 ? 1 : 0);
    if (z < 0) {
        if constexpr (ML > 0) {
            __shared__ int rows_sid[NROW_MAX_ENTRIES];
            // entry workgroups first (the owner of a popular row is the longest), then the sweep
            const int ne = row_blocks - rows.cb_entries;
            for (int blk = blk0; blk < (rows.sweep_elsewhere ? ne : row_blocks); blk += nblk) {
                narre_rows_block<ML, 1>(rows, blk < ne ? rows.cb_entries + blk : blk - ne, rows_sid);
                __syncthreads();
            }
        }
    } else if (z < ntower) {
        if constexpr (WIDE) wgrad_block(w, blockIdx.x, blockIdx.y, z);
        else if (packed) wgrad_block_packed(w, blockIdx.x, blockIdx.y, z);   // grid.x = ceil(F / 4)
        else wgrad_block(w, blockIdx.x, blockIdx.y, z);
    } else if (z < ntower + backward_cs_slices(cs_blocks, nblk)) {
        // (one block per workgroup, over as many slices as that takes: TransNet's 220 blocks on a 200-workgroup
        // slice made 20 workgroups take two in a row -- the launch's tail, 10.9 us instead of 5.6)
        const int blk = (z - ntower) * nblk + blk0;
        if (blk < cs_blocks) colsum_block(c, blk);
    } else {
        token_mark_block(nx, blk0, nblk, WG_THREADS);
    }
#ifdef R4R_TRACE
    __syncthreads();                                        // the workgroup's end, not thread 0's
#endif
    BWD_STAMP(1, wall_clock64())
}

// The ID-table role as a launch of its own (data parallel: the entries of ALL ranks, gathered): up to
// 16,384 entries per table, their ids in dynamic LDS.
constexpr int NROW_DP_WORDS = 4, NROW_DP_MAX_ENTRIES = 64 * 64 * NROW_DP_WORDS;
template <int ML, bool BLK = false>
static __global__ __launch_bounds__(NROW_THREADS) void narre_rows_kernel(RowSweep) {
    const RowSweep &w = kernel_args<RowSweep>();
    extern __shared__ int rows_sid_dyn[];
    narre_rows_block<ML, NROW_DP_WORDS, BLK>(w, (int)blockIdx.x, rows_sid_dyn);
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
                                                                    TokenArgs nx, DenseAdam opt, int opt_blocks = 0,
                                                                    RowSweep rows = RowSweep{}) {
    const int bx = blockIdx.x;
    if (rows.sweep_elsewhere && bx >= red_blocks + comp_blocks + opt_blocks) {
        // NARRE: the ID tables' untouched rows (the touched ones were updated by the backward launch's
        // entry waves; the two sets are disjoint), two sweep workgroups per grid column
        const int sb = (bx - red_blocks - comp_blocks - opt_blocks) * (int)gridDim.y + (int)blockIdx.y;
        if (sb < rows.cb_entries) narre_sweep_block(rows, sb);
        return;
    }
    if (bx < red_blocks) {
        // (parameter + moments requested with the partials, Adam fed from the register: engine.hip)
        const WgradTower &tw = w.t[blockIdx.y];
        const int nw = w.F * 3 * w.E;
        const int i = bx * NRED_THREADS + threadIdx.x;
        const float *gp0 = i < nw ? tw.d_w + i : (i < nw + w.F ? tw.d_b + (i - nw) : nullptr);
        const int64_t o = gp0 ? gp0 - opt.g : 0;
        float P = 0.f, M = 0.f, V = 0.f;
        if (opt.on && gp0) { P = opt.p[o]; M = opt.m[o]; V = opt.v[o]; }
        float *dst;
        const float g = wgrad_reduce_elem(w, blockIdx.y, i, dst);
        if (opt.on && dst) {
            adam_elem(P, g, M, V, opt.s);
            opt.p[o] = P; opt.m[o] = M; opt.v[o] = V;
        }
    } else if (bx < red_blocks + comp_blocks) {
        token_compact_auto<NRED_THREADS / 64>(nx.t[blockIdx.y], nx.V, bx - red_blocks);
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
