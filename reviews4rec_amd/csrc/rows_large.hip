// Dense-Adam update of the rows of ONE ID table (+ its bias vector) that a step's compact entries name, for ANY
// number of entries -- the path behind the fused steps' entry-count caps (narre_rows_block keeps every entry id in
// LDS and a hit mask in registers: <= 4,096 entries fused, <= 16,384 stand-alone; a data-parallel NARRE step at a
// global batch of 8,192 gathers 90,112 per table).
//
// Reference behaviour restated: torch.optim.Adam updates EVERY row of an nn.Embedding every step (main.py:94-96,
// NARRE.py:20-23,85-107); a row's gradient is the sum of the gradient rows of all entries that name it.  Rows no entry
// names are the caller's tagged sweep (step_device.h: narre_sweep_block); this launch chain owns the named rows.
//
// Determinism (replicas must hold the same bits): no floating-point atomics and no order that depends
// on scheduling.  Entries are split by a STABLE counting sort into RL_G buckets by row id (integer atomics on
// counters only; positions come from prefix sums), so that inside a bucket entries keep their ascending order; an
// entry wave then scans only its bucket -- O(n^2 / RL_G) id comparisons for the launch instead of O(n^2):
//   1 rl_hist     one wave per 64 entries: bucket histogram of the block
//   2 rl_scan     one workgroup per bucket: exclusive prefix of the blocks' counts; bucket totals
//   3 rl_layout   one workgroup: bucket begins, entry-group begins
//   4 rl_scatter  one wave per 64 entries: position = bucket begin + blocks before + same-bucket lanes before
//   5 rl_apply    one wave per entry: if no earlier entry of the bucket names its row, the wave OWNS the row -- every
//                 lane adds up the gradient rows of ITS hits (chunk by chunk, ascending), one fixed cross-lane sum
//                 per column, then the row's Adam update (and the bias element's)
// Same arithmetic as the fused entry waves (adam_elem_fast: every path that touches an ID-table row uses it).
#include <stdlib.h>

#include "common.h"
#include "adam_device.h"

namespace r4r {

constexpr int RL_G = 256;                  // buckets (row id mod RL_G); bucket RL_G = padding entries (id < 0)
constexpr int RL_NB = RL_G + 1;
constexpr int RL_EPW = 4, RL_EPG = 4 * RL_EPW;   // entries per wave / per 256-thread workgroup of rl_apply
constexpr int RL_WL = 32;                  // columns per pass of an owner wave (wider rows: several passes)

struct RlArgs {
    const int64_t *ids; const float *grads, *gbias;
    int64_t n;
    int W, nblk;
    float *p, *m, *v, *bp, *bm, *bv;
    int64_t rows;
    int *perm, *bid, *hist, *bucket_begin, *group_begin;     // scratch
    AdamScalars s;
};

__device__ __forceinline__ int rl_bucket(int64_t id, int64_t rows) { return (id < 0 || id >= rows) ? RL_G : (int)(id & (RL_G - 1)); }

__global__ __launch_bounds__(256) void rl_hist_kernel(RlArgs a) {
    __shared__ int h[4][RL_NB];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int blk = blockIdx.x * 4 + w;
    for (int i = lane; i < RL_NB; i += 64) h[w][i] = 0;
    __syncthreads();
    const int64_t e = (int64_t)blk * 64 + lane;
    if (blk < a.nblk && e < a.n) atomicAdd(&h[w][rl_bucket(a.ids[e], a.rows)], 1);       // (counts: order-independent)
    __syncthreads();
    if (blk < a.nblk)
        for (int i = lane; i < RL_NB; i += 64) a.hist[(size_t)blk * RL_NB + i] = h[w][i];
}

// bucket b = blockIdx.x: hist[blk][b] <- number of bucket-b entries in blocks before blk; bucket_begin[b] <- total
__global__ __launch_bounds__(256) void rl_scan_kernel(RlArgs a) {
    __shared__ int part[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int per = (a.nblk + 255) / 256;                   // consecutive blocks per thread
    const int lo = tid * per, hi = min(a.nblk, lo + per);
    int sum = 0;
    for (int k = lo; k < hi; ++k) sum += a.hist[(size_t)k * RL_NB + b];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {                                          // 256 values: a serial scan is a microsecond
        int run = 0;
        for (int k = 0; k < 256; ++k) { const int t = part[k]; part[k] = run; run += t; }
        a.bucket_begin[b] = run;                             // (total for now: rl_layout turns the totals into begins)
    }
    __syncthreads();
    int run = part[tid];
    for (int k = lo; k < hi; ++k) {
        const int t = a.hist[(size_t)k * RL_NB + b];
        a.hist[(size_t)k * RL_NB + b] = run;
        run += t;
    }
}

__global__ void rl_layout_kernel(RlArgs a) {
    if (threadIdx.x != 0) return;
    int run = 0, groups = 0;
    for (int b = 0; b < RL_NB; ++b) {
        const int t = a.bucket_begin[b];
        a.bucket_begin[b] = run;
        a.group_begin[b] = groups;
        run += t;
        if (b < RL_G) groups += (t + RL_EPG - 1) / RL_EPG;   // the padding bucket gets no entry waves
    }
    a.bucket_begin[RL_NB] = run;
    a.group_begin[RL_NB] = groups;
}

__global__ __launch_bounds__(256) void rl_scatter_kernel(RlArgs a) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int blk = blockIdx.x * 4 + w;
    if (blk >= a.nblk) return;
    const int64_t e = (int64_t)blk * 64 + lane;
    const bool live = e < a.n;
    const int64_t id = live ? a.ids[e] : -1;
    const int b = live ? rl_bucket(id, a.rows) : -1;
    // lanes of the same bucket, in lane order (= ascending entry order): one round per distinct bucket of the block
    int rank = 0;
    unsigned long long todo = __ballot(live);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int lb = __builtin_amdgcn_readlane(b, leader);
        const unsigned long long same = __ballot(live && b == lb);
        if (b == lb) rank = __popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    if (live) {
        const int dst = a.bucket_begin[b] + a.hist[(size_t)blk * RL_NB + b] + rank;
        a.perm[dst] = (int)e;
        a.bid[dst] = b == RL_G ? -1 : (int)id;
    }
}

__global__ __launch_bounds__(256) void rl_apply_kernel(RlArgs) {
    const RlArgs &a = kernel_args<RlArgs>();                // (fields loaded at their uses: common.h)
    __shared__ int s_bucket;
    const int g = blockIdx.x;
    if (g >= a.group_begin[RL_NB]) return;                  // (the grid is the host's upper bound)
    if (threadIdx.x == 0) {                                 // the bucket of this entry group: last b with group_begin[b] <= g
        int lo = 0, hi = RL_G;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.group_begin[mid] <= g) lo = mid; else hi = mid; }
        s_bucket = lo;
    }
    __syncthreads();
    const int b = s_bucket;
    const int bb = a.bucket_begin[b], be = a.bucket_begin[b + 1];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int W = a.W;
    for (int q = 0; q < RL_EPW; ++q) {
        const int j = bb + (g - a.group_begin[b]) * RL_EPG + q * 4 + w;      // interleaved over the four waves
        if (j >= be) continue;                               // uniform over the wave
        const int row = a.bid[j];
        // is an earlier entry of the bucket the owner of this row?
        bool owner = true;
        for (int c0 = bb; c0 < j && owner; c0 += 64) {
            const int jj = c0 + lane;
            if (__ballot(jj < j && a.bid[jj] == row)) owner = false;
        }
        if (!owner) continue;
        // the row's parameters and moments: requested now, used after the sums
        const bool has_bias = a.bp != nullptr;
        float Pb = 0.f, Mb = 0.f, Vb = 0.f;
        if (has_bias) { Pb = a.bp[row]; Mb = a.bm[row]; Vb = a.bv[row]; }
        float gb = 0.f;
        for (int cb = 0; cb < W; cb += RL_WL) {              // column passes of RL_WL
            const int nc = min(RL_WL, W - cb);
            const int64_t o = (int64_t)row * W + cb + (lane < nc ? lane : 0);
            float P = a.p[o], M = a.m[o], V = a.v[o];
            float rv[RL_WL];
#pragma unroll
            for (int col = 0; col < RL_WL; ++col) rv[col] = 0.f;
            for (int c0 = bb + ((j - bb) & ~63); c0 < be; c0 += 128) {       // (no hit before j: j is the first)
                // two chunks per round, their loads together
                int ent[2];
                bool hit[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int jj = c0 + 64 * u + lane;
                    hit[u] = jj >= j && jj < be && a.bid[jj] == row;
                    ent[u] = hit[u] ? a.perm[jj] : 0;
                }
                if (!__ballot(hit[0] || hit[1])) continue;
                float tmp[2][RL_WL], tg[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
#pragma unroll
                    for (int col = 0; col < RL_WL; ++col)
                        tmp[u][col] = (hit[u] && col < nc) ? a.grads[(int64_t)ent[u] * W + cb + col] : 0.f;
                    tg[u] = (hit[u] && cb == 0 && a.gbias) ? a.gbias[ent[u]] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {                // ascending entry order within the lane
#pragma unroll
                    for (int col = 0; col < RL_WL; ++col) rv[col] += tmp[u][col];
                    gb += tg[u];
                }
            }
            float acc = 0.f;                                 // lane < nc: column cb + lane of the row
#pragma unroll
            for (int col = 0; col < RL_WL; ++col) {
                if (col < nc) {                              // uniform
                    const float sum = wave_sum(rv[col]);
                    if (lane == col) acc = sum;
                }
            }
            if (lane < nc) {
                adam_elem_fast(P, acc, M, V, a.s);
                a.p[o] = P; a.m[o] = M; a.v[o] = V;
            }
        }
        if (has_bias) {
            const float accb = wave_sum(gb);
            if (lane == 0) {                                 // gradient zero if no entry of the row carries one
                adam_elem_fast(Pb, accb, Mb, Vb, a.s);
                a.bp[row] = Pb; a.bm[row] = Mb; a.bv[row] = Vb;
            }
        }
    }
}

static size_t rl_align(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace r4r

using namespace r4r;

extern "C" size_t r4r_rows_large_ws_bytes(int64_t entries) {
    if (entries < 0) return 0;
    const size_t nblk = (size_t)((entries + 63) / 64);
    return 2 * rl_align((size_t)entries * 4) + rl_align(nblk * RL_NB * 4) + 2 * rl_align((RL_NB + 1) * 4) + 256;
}

extern "C" int r4r_rows_apply_large(const int64_t *ids, const float *grads, const float *gbias, int64_t entries, int W,
                                    float *p, float *m, float *v, float *bp, float *bm, float *bv, int64_t rows,
                                    void *scratch, size_t scratch_bytes,
                                    float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                                    void *stream) {
    R4R_REQUIRE(ids && grads && p && m && v && scratch, "rows_apply_large: null pointer");
    R4R_REQUIRE(!bp == !bm && !bp == !bv, "rows_apply_large: the bias vector and its two moments go together");
    R4R_REQUIRE(!gbias || bp, "rows_apply_large: bias gradients without a bias vector");
    R4R_REQUIRE(entries >= 0 && entries < (1ll << 31) - 64 && rows > 0 && rows < (1ll << 31), "rows_apply_large: bad sizes");
    R4R_REQUIRE(W >= 1 && W <= 1024, "rows_apply_large: row width %d outside 1..1024", W);
    R4R_REQUIRE(adam_step >= 1, "rows_apply_large: adam_step must be >= 1");
    if (scratch_bytes < r4r_rows_large_ws_bytes(entries)) {
        set_error("rows_apply_large: scratch %zu < %zu bytes", scratch_bytes, r4r_rows_large_ws_bytes(entries));
        return R4R_ERR_WORKSPACE;
    }
    if (entries == 0) return R4R_OK;
    hipStream_t st = as_stream(stream);
    RlArgs a;
    a.ids = ids; a.grads = grads; a.gbias = gbias; a.n = entries; a.W = W;
    a.nblk = (int)((entries + 63) / 64);
    a.p = p; a.m = m; a.v = v; a.bp = bp; a.bm = bm; a.bv = bv; a.rows = rows;
    char *q = static_cast<char *>(scratch);
    auto take = [&](size_t nbytes) { char *r = q; q += rl_align(nbytes); return reinterpret_cast<int *>(r); };
    a.perm = take((size_t)entries * 4);
    a.bid = take((size_t)entries * 4);
    a.hist = take((size_t)a.nblk * RL_NB * 4);
    a.bucket_begin = take((RL_NB + 1) * 4);
    a.group_begin = take((RL_NB + 1) * 4);
    a.s = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    const unsigned wblocks = (unsigned)((a.nblk + 3) / 4);
    rl_hist_kernel<<<wblocks, 256, 0, st>>>(a);
    rl_scan_kernel<<<RL_NB, 256, 0, st>>>(a);
    rl_layout_kernel<<<1, 64, 0, st>>>(a);
    rl_scatter_kernel<<<wblocks, 256, 0, st>>>(a);
    const int64_t groups_max = entries / RL_EPG + RL_G;      // sum over buckets of ceil(count / RL_EPG) <= this
    rl_apply_kernel<<<(unsigned)groups_max, 256, 0, st>>>(a);
    return check_launch("rows_apply_large");
}
