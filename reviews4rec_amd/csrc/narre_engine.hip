// Fused native training step for NARRE (pytorch_models/NARRE.py:10-124): per rating, R reviews
// of W words on the user side and on the item side.  One C call = forward, loss (loss.py:7-11),
// backward and the dense Adam update (main.py:56-60,94-96) in five launches:
//
//   1+2  token compaction (rides on the previous step when the loop announces the next batch),
//        projection GEMM + gather-add-max over the 2 x B*R review documents (project.hip), or the
//        direct gather-fused conv for small launches (textcnn.hip)
//   3    narre_head_kernel   one workgroup per rating: pool finish, TextCNN's FC + dropout per
//                            review, both attention scorers + softmax (NARRE.py:53-64), the ID
//                            vectors, the interaction, `final`, the bias head, SE -- and the whole
//                            backward of that down to d/d pooled, with the rating's contribution
//                            to every head parameter gradient written as ONE row of a [B, NHP]
//                            matrix and its ID-table gradient rows kept compact
//   4    backward launch     argmax-sparse conv wgrad of both towers (wgrad_device.h), the column
//                            sums of the [B, NHP] matrix in fixed order, next batch's token marks,
//                            and the Adam sweep over the two ID tables and the two bias vectors
//                            (narre_rows_block): rows no rating touched have gradient zero (never
//                            materialised), touched rows sum their compact entries in ascending
//                            order (deterministic)
//   5    reduce launch       wgrad partials -> gradient, Adam on every dense parameter, next
//                            batch's token compaction
//
// The op-by-op path issues ~130 launches for the same step.
#include "step_device.h"
#include "trace_device.h"

namespace r4r {

constexpr int NHEAD_THREADS = 512;   // threads of the per-rating head workgroup
constexpr int NARRE_MAX_L = 64, NARRE_MAX_R = 64;  // the head's widest instantiation (the fused ID-table role: NR_MAX_L / NR_MAX_R = 32)

// flat dense-parameter layout (21 slots); slots 0,1 / 4,5 are the conv weight + bias of the towers
enum { NP_UCW = 0, NP_UCB, NP_UFW, NP_UFB, NP_ICW, NP_ICB, NP_IFW, NP_IFB,
       NP_AUW0, NP_AUB0, NP_AUW3, NP_AUB3, NP_AIW0, NP_AIB0, NP_AIW3, NP_AIB3,
       NP_F1W, NP_F1B, NP_F3W, NP_F3B, NP_GB, NP_COUNT };

struct NLayout { int64_t off[NP_COUNT], size[NP_COUNT], total; };

static NLayout narre_layout(int E, int L) {
    NLayout lay;
    const int64_t sz[NP_COUNT] = {(int64_t)NF * 3 * E, NF, (int64_t)L * NF, L, (int64_t)NF * 3 * E, NF, (int64_t)L * NF, L,
                                  (int64_t)L * 2 * L, L, L, 1, (int64_t)L * 2 * L, L, L, 1,
                                  (int64_t)L * L, L, L, 1, 1};
    int64_t o = 0;
    for (int i = 0; i < NP_COUNT; ++i) {
        lay.off[i] = o;
        lay.size[i] = sz[i];
        o += (sz[i] + 31) & ~(int64_t)31;          // 128-byte aligned slots (whole cache lines for the GEMM's weight-row pieces); pad floats stay 0 forever
    }
    lay.total = o;
    return lay;
}

// Per-rating head-gradient row: the non-conv dense parameters in flat-layout order, i.e. flat
// offsets [off[UFW], off[ICW]) followed by [off[IFW], total).  `hp_index` maps a flat offset of a
// head parameter to its column.
struct HeadCols { int64_t lo0, hi0, lo1, hi1; int n; };
static HeadCols head_cols(const NLayout &lay) {
    HeadCols h;
    h.lo0 = lay.off[NP_UFW]; h.hi0 = lay.off[NP_ICW]; h.lo1 = lay.off[NP_IFW]; h.hi1 = lay.total;
    h.n = (int)((h.hi0 - h.lo0) + (h.hi1 - h.lo1));
    return h;
}

struct NarreHead {
    const float *pmax[2]; const int *parg[2];       // conv partials [N, tiles, NP], N = B*R
    const float *flat_p;                            // dense parameters (layout above)
    int off[NP_COUNT];                              // flat offsets (fit int: < 2^31 floats)
    int col0_lo, col0_n, col1_lo;                   // head-column mapping: off in [lo0, lo0+n0) -> off-lo0, else n0 + off - lo1
    const float *emb[2], *bias[2];                  // user / item embedding tables [rows, L], bias vectors
    const int64_t *self_id[2];                      // uid, iid [B]
    const int64_t *other_id[2];                     // side 0 (user tower): reviewed_items [B,R] -> ITEM table
                                                    // side 1 (item tower): users_who_reviewed [B,R] -> USER table
    const float *y;
    float *pooled[2]; int *argmax[2]; float *g_pooled[2];   // [N, 100]
    float *part;                                    // [B, NHP]
    float *grow[2]; int64_t *gid[2];                // compact rows / ids of table t: [B(1+R), L], [B(1+R)]
    float *g;                                       // [B] d mean(SE) / d pred
    int *tag[2];                                    // row tags of the user / item side
    float *mult;                                    // [B, 4RL + 3L] dropout multipliers
    float *pred, *se;
    int64_t B;
    int R, L, tiles, nhp, training, want_grad, now;
    float p_drop, inv_denom;
    uint64_t seed, offset;
};

__device__ __forceinline__ int head_col(const NarreHead &a, int flat_off) {
    return flat_off < a.col0_lo + a.col0_n ? flat_off - a.col0_lo : a.col0_n + flat_off - a.col1_lo;
}

// One workgroup per rating.  Dynamic LDS, carved for the actual R and L.
// MR / ML: compile-time caps of R / L (register arrays and unrolled loops are sized by them:
// the <= 16 instantiation is 10 % faster on the default shape than the <= 32 one); NT: threads.
// The kernel is bound by vector-ALU issue (index arithmetic around ~2000-element loops), with one
// workgroup per CU on half the CUs -- so more waves per rating, not fewer instructions per
// memory access, is what shortens it.
// A pointer array of the kernel arguments indexed by a run-time side makes every lane FETCH the pointer from the
// argument segment -- a memory round trip in front of the access it serves (and, the loads returning in order, behind
// everything requested before it).  Selecting between the two constant-index elements is two scalar registers and a
// v_cndmask.
#define SEL2(arr, s) ((s) ? (arr)[1] : (arr)[0])
HEAD_TRACE_DEFINE(r4r_debug_narre_head_trace)
BWD_TRACE_DEFINE(r4r_debug_narre_bwd_trace)
template <int MR, int ML, int NT>
__global__ __launch_bounds__(NT) void narre_head_kernel(NarreHead by_value) {
    // (fields loaded at their uses: common.h -- 88 spilled scalars before, 0 now; the 64 x 64 instantiation is at its
    // 256-VGPR cap and puts the address arithmetic of lazy loads into scratch memory: it keeps the up-front form)
    const NarreHead &a = ML <= 32 ? kernel_args<NarreHead>() : by_value;
    HEAD_STAMP(0)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int R = a.R, L = a.L, RL = R * L, L2 = 2 * L;
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x;
    // ---- LDS carve
    float *P = sm;                          // [2][R][100]   pooled conv features
    float *fcw = P + 2 * R * NF;            // [2][L][101]
    float *W0 = fcw + 2 * L * (NF + 1);     // [2][L][2L+1]
    float *F1 = W0 + 2 * L * (L2 + 1);      // [L][L+1]
    float *x = F1 + L * (L + 1);            // [2][R][L]     TextCNN outputs after dropout
    float *xm = x + 2 * RL;                 // [2][R][L]     their dropout multipliers
    float *o = xm + 2 * RL;                 // [2][R][L]     the other side's ID vectors
    float *h = o + 2 * RL;                  // [2][R][L]     scorer hidden (after relu, before dropout)
    float *hm = h + 2 * RL;                 // [2][R][L]     scorer dropout multipliers
    float *dz = hm + 2 * RL;                // [2][R][L]     scratch: dhpre, then dz
    float *sc = dz + 2 * RL;                // [2][R]        scores -> attention weights
    float *da = sc + 2 * R;                 // [2][R]        d attention -> d score
    float *sv = da + 2 * R;                 // small vectors, 12 x [2][L] slots below
    float *fcb = sv, *b0 = sv + L2, *w3 = sv + 2 * L2, *ev = sv + 3 * L2, *evm = sv + 4 * L2, *v = sv + 5 * L2,
          *dv = sv + 6 * L2, *cdv = sv + 7 * L2 /* cd [L], cm [L] */, *fv = sv + 8 * L2 /* f1b [L], F3 [L] */,
          *fh = sv + 9 * L2 /* fh [L], dfpre [L] */, *misc = sv + 10 * L2;   // misc: b3[2], f3b, gb, ub, ib, g, dot[2]
    const float *fp = a.flat_p;
    // Index decompositions without integer division by a runtime value (~40 instructions each, and
    // the kernel does hundreds per thread): a side is a compare (there are two), NF is a compile-time
    // constant, and x / d for x <= 6400 is exact as (int)((x + 0.5f) * (1.f / d)).
    const float invL = 1.f / (float)L, invL2 = 1.f / (float)L2;
    auto qd = [](int x, float inv) { return (int)(((float)x + 0.5f) * inv); };
    const float keep = 1.f / (1.f - a.p_drop);
    const bool drop = a.training && a.p_drop > 0.f;
    const int ND = 4 * RL + 3 * L;                          // dropout draws per rating
    auto draw = [&](int k) -> float {                       // multiplier of draw k of this rating
        if (!drop) return 1.f;
        const uint32_t r = philox_first_word(a.offset + (uint64_t)(b * ND + k), a.seed);
        const float m = ((float)(r >> 8) * (1.0f / 16777216.0f) >= a.p_drop) ? keep : 0.f;
        if (a.mult) a.mult[b * ND + k] = m;
        return m;
    };
    if (!drop && a.mult) for (int k = tid; k < ND; k += NT) a.mult[b * ND + k] = 1.f;

    // ---- S0: weights -> LDS, pool finish, ID vectors.  The three big reads -- the pooling
    // partials, the FC matrices, the scorer matrices -- are issued into registers before anything
    // waits (a load -> LDS-store loop is one memory round trip per iteration: 8 + 8 + 2 of them)
    // (the kernel is bound by vector-ALU issue: the 2,000-element partial and FC-weight blocks move as 16-byte units
    // -- NF / 4 = 25 filter quads per row -- a quarter of the loads, index computations and stores)
    constexpr int NQ4 = NF / 4;
    typedef float hq4 __attribute__((ext_vector_type(4)));
    typedef int hi4 __attribute__((ext_vector_type(4)));
    constexpr int PREG = (2 * MR * NQ4 + NT - 1) / NT, WREG = (2 * ML * NQ4 + NT - 1) / NT,
                  AREG = (2 * ML * 2 * ML + NT - 1) / NT;
    hq4 pv[PREG], wv[WREG];
    float av[AREG];
    hi4 pa[PREG];
    const bool one_tile = a.tiles == 1;
    // The ids go out FIRST: the ID vectors need them (a second, dependent round trip), and loads return in order -- with
    // the ids the oldest requests, waiting for them does not wait for the big reads behind them, and the second round
    // trip runs under the first.
    constexpr int FREG = (ML * ML + NT - 1) / NT, OREG = (2 * MR * ML + NT - 1) / NT;
    float f1v[FREG], ov[OREG];
    int64_t oid[OREG];
#pragma unroll
    for (int u = 0; u < OREG; ++u) {
        oid[u] = 0;
        if (NT * u < 2 * RL) {
            const int i = min(tid + NT * u, 2 * RL - 1);
            const int s = i >= RL, rem = i - s * RL, r = qd(rem, invL);
            oid[u] = SEL2(a.other_id, s)[b * R + r];
        }
    }
    const int ti = min(tid, L2 - 1), ts = ti >= L, tl = ti - ts * L;          // this thread's (side, l) for the [2][L] vectors
    const int64_t sid_r = SEL2(a.self_id, ts)[b];
    const int64_t sid0 = a.self_id[0][b], sid1 = a.self_id[1][b];
#pragma unroll
    for (int u = 0; u < PREG; ++u) {
        pv[u] = (hq4){0.f, 0.f, 0.f, 0.f}; pa[u] = (hi4){0, 0, 0, 0};
        if (one_tile && NT * u < 2 * R * NQ4) {            // uniform: rounds past the end cost nothing
            const int i = min(tid + NT * u, 2 * R * NQ4 - 1);
            const int s = i >= R * NQ4, rem = i - s * R * NQ4, rr = rem / NQ4, q4 = rem - rr * NQ4;
            const size_t q = ((size_t)(b * R + rr) * a.tiles) * NP + 4 * q4;
            pv[u] = *reinterpret_cast<const hq4 *>(SEL2(a.pmax, s) + q);
            pa[u] = *reinterpret_cast<const hi4 *>(SEL2(a.parg, s) + q);
        }
    }
#pragma unroll
    for (int u = 0; u < WREG; ++u) {
        wv[u] = (hq4){0.f, 0.f, 0.f, 0.f};
        if (NT * u < 2 * L * NQ4) {
            const int i = min(tid + NT * u, 2 * L * NQ4 - 1);
            const int s = i >= L * NQ4;
            wv[u] = *reinterpret_cast<const hq4 *>(fp + (s ? a.off[NP_IFW] : a.off[NP_UFW]) + 4 * (i - s * L * NQ4));
        }
    }
#pragma unroll
    for (int u = 0; u < AREG; ++u) {
        av[u] = 0.f;
        if (NT * u < 2 * L * L2) {
            const int i = min(tid + NT * u, 2 * L * L2 - 1);
            const int s = i >= L * L2;
            av[u] = fp[(s ? a.off[NP_AIW0] : a.off[NP_AUW0]) + i - s * L * L2];
        }
    }
    // the small reads ride in the same round trip
#pragma unroll
    for (int u = 0; u < FREG; ++u) f1v[u] = (NT * u < L * L) ? fp[a.off[NP_F1W] + min(tid + NT * u, L * L - 1)] : 0.f;
    const float fcb_r = fp[(ts ? a.off[NP_IFB] : a.off[NP_UFB]) + tl], b0_r = fp[(ts ? a.off[NP_AIB0] : a.off[NP_AUB0]) + tl],
                w3_r = fp[(ts ? a.off[NP_AIW3] : a.off[NP_AUW3]) + tl];
    const float f1b_r = fp[a.off[NP_F1B] + min(tid, L - 1)], f3w_r = fp[a.off[NP_F3W] + min(tid, L - 1)];
    const float m0 = fp[a.off[NP_AUB3]], m1 = fp[a.off[NP_AIB3]], m2 = fp[a.off[NP_F3B]], m3 = fp[a.off[NP_GB]];
    const float y_r = a.y ? a.y[b] : 0.f;                   // (the rating rides in the first round trip: wave 0's S8 does not wait for it)
    // round 2: the reads that depend on ids
#pragma unroll
    for (int u = 0; u < OREG; ++u) {
        ov[u] = 0.f;
        if (NT * u < 2 * RL) {
            const int i = min(tid + NT * u, 2 * RL - 1);
            const int s = i >= RL, rem = i - s * RL, r = qd(rem, invL), l = rem - r * L;
            ov[u] = SEL2(a.emb, 1 - s)[oid[u] * L + l];
        }
    }
    const float ev_r = SEL2(a.emb, ts)[sid_r * L + tl];
    const float ub_r = a.bias[0][sid0], ib_r = a.bias[1][sid1];
#pragma unroll
    for (int u = 0; u < WREG; ++u) {
        const int i = tid + NT * u;
        if (i < 2 * L * NQ4) {
            const int s = i >= L * NQ4, r = i - s * L * NQ4, l = r / NQ4;
            float *dst = fcw + (s * L + l) * (NF + 1) + 4 * (r - l * NQ4);      // (rows of NF + 1 floats: four 4-byte writes)
#pragma unroll
            for (int c = 0; c < 4; ++c) dst[c] = wv[u][c];
        }
    }
#pragma unroll
    for (int u = 0; u < AREG; ++u) {
        const int i = tid + NT * u;
        if (i < 2 * L * L2) {
            const int s = i >= L * L2, r = i - s * L * L2, k = qd(r, invL2);
            W0[(s * L + k) * (L2 + 1) + r - k * L2] = av[u];
        }
    }
#pragma unroll
    for (int u = 0; u < FREG; ++u) {
        const int i = tid + NT * u;
        if (i < L * L) { const int k = qd(i, invL); F1[k * (L + 1) + i - k * L] = f1v[u]; }
    }
    if (tid < L2) {
        fcb[tid] = fcb_r; b0[tid] = b0_r; w3[tid] = w3_r;
        evm[tid] = draw(4 * RL + ts * L + tl);
    }
    if (tid < L) { fv[tid] = f1b_r; fv[L + tid] = f3w_r; }
    if (tid == 0) { misc[0] = m0; misc[1] = m1; misc[2] = m2; misc[3] = m3; }
    if (one_tile) {                                         // pool finish of a one-tile document: relu + argmax
#pragma unroll
        for (int u = 0; u < PREG; ++u) {
            const int i = tid + NT * u;
            if (i < 2 * R * NQ4) {
                const int s = i >= R * NQ4, rem = i - s * R * NQ4, rr = rem / NQ4, q4 = rem - rr * NQ4;
                const int64_t n = b * R + rr;
                hq4 best = pv[u];
                hi4 bp = pa[u];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (!(best[c] > 0.f)) { best[c] = 0.f; bp[c] = -1; }
                *reinterpret_cast<hq4 *>(P + (s * R + rr) * NF + 4 * q4) = best;
                *reinterpret_cast<hq4 *>(SEL2(a.pooled, s) + n * NF + 4 * q4) = best;
                *reinterpret_cast<hi4 *>(SEL2(a.argmax, s) + n * NF + 4 * q4) = bp;
            }
        }
    } else {
        for (int i = tid; i < 2 * R * NF; i += NT) {       // pool finish: max over tiles, relu, first argmax
            const int s = i >= R * NF, rem = i - s * R * NF, rr = rem / NF, f = rem - rr * NF;
            const int64_t n = b * R + rr;
            float best = -INFINITY;
            int bp = -1;
            for (int k = 0; k < a.tiles; ++k) {
                const size_t q = ((size_t)n * a.tiles + k) * NP + f;
                const float val = SEL2(a.pmax, s)[q];
                if (val > best) { best = val; bp = SEL2(a.parg, s)[q]; }
            }
            if (!(best > 0.f)) { best = 0.f; bp = -1; }
            P[i] = best;
            SEL2(a.pooled, s)[n * NF + f] = best;
            SEL2(a.argmax, s)[n * NF + f] = bp;
        }
    }
    __syncthreads();
    HEAD_STAMP(1)
    // ---- S1: TextCNN FC + dropout per review (common_pytorch_models.py:35-37)
    for (int i = tid; i < 2 * RL; i += NT) {
        const int s = i >= RL, rem = i - s * RL, r = qd(rem, invL), l = rem - r * L;
        const float *pr = P + (s * R + r) * NF, *wr = fcw + (s * L + l) * (NF + 1);
        float acc = 0.f;
        for (int f = 0; f < NF; ++f) acc = fmaf(pr[f], wr[f], acc);
        const float m = draw(s * RL + r * L + l);
        xm[i] = m;
        x[i] = (acc + fcb[s * L + l]) * m;
    }
    // the round-2 reads (ID vectors, biases) land in LDS only now: S1 ran under their latency
    if (tid < L2) ev[tid] = ev_r;
    if (tid == 0) { misc[4] = ub_r; misc[5] = ib_r; }
#pragma unroll
    for (int u = 0; u < OREG; ++u) {                        // other side's ID vectors: side s reads table 1-s
        const int i = tid + NT * u;
        if (i < 2 * RL) o[i] = ov[u];
    }
    __syncthreads();
    HEAD_STAMP(2)
    // ---- S2: scorer hidden layer on [x ; other] (NARRE.py:55-58)
    for (int i = tid; i < 2 * RL; i += NT) {
        const int s = i >= RL, rem = i - s * RL, r = qd(rem, invL), k = rem - r * L;
        const float *wr = W0 + (s * L + k) * (L2 + 1), *xr = x + (s * R + r) * L, *orow = o + (s * R + r) * L;
        float acc = 0.f;
        for (int j = 0; j < L; ++j) acc = fmaf(xr[j], wr[j], acc);
        for (int j = 0; j < L; ++j) acc = fmaf(orow[j], wr[L + j], acc);
        acc += b0[s * L + k];
        h[i] = acc > 0.f ? acc : 0.f;
        hm[i] = draw(2 * RL + s * RL + r * L + k);
    }
    __syncthreads();
    HEAD_STAMP(3)
    // ---- S3 + S4 + S5 by ONE WAVE per side (wave s = side s), with no workgroup barrier between them.  A side's
    // scores, their softmax and the attended vector are 2 R + L values; as three barrier-separated stages of all 512
    // threads they cost 1.5 us of barrier round trips around a few hundred nanoseconds of work, and the nine stages
    // from here to the scorer's backward 6.6 us of the launch's 18 (tools/head_trace.py).  A wave's LDS operations
    // execute in order, so its lanes see one another's writes behind a wave-level fence; per output the operations
    // and their order are what they were.
    if (tid < 128) {
        const int s = tid >> 6, ln = tid & 63;
        float val = -INFINITY;
        if (ln < R) {
            const int i = s * R + ln;
            float acc = 0.f;
            for (int k = 0; k < L; ++k) acc = fmaf(h[i * L + k] * hm[i * L + k], w3[s * L + k], acc);
            val = acc + misc[s];
        }
        // softmax over the R reviews of the side (pads are not masked, like the reference); all 64 lanes take part in
        // the reductions: never reduce under a lane-divergent branch
        const float mx = wave_max(val);
        const float e = ln < R ? expf(val - mx) : 0.f;
        const float den = wave_sum(e);
        if (ln < R) sc[s * R + ln] = e / den;
        wave_lds_fence();
        if (ln < L) {                                       // attended review vector + the ID vector (NARRE.py:110-111)
            const int i = s * L + ln;
            float acc = 0.f;
            for (int r = 0; r < R; ++r) acc = fmaf(sc[s * R + r], x[(s * R + r) * L + ln], acc);
            v[i] = acc + ev[i] * evm[i];
        }
    }
    __syncthreads();
    HEAD_STAMP(4)
    // ---- S6 + S7 + S8 and the vector parts of B1 / B2 by wave 0: L-element vectors, each step reading what the one
    // before wrote.  The other waves meanwhile write the step's ID entries and row tags (nothing computed here feeds them).
    const int64_t nself = a.B;                              // entries [0, B): self rows, then B*R others
    float *prow = a.part + (size_t)b * a.nhp;
    if (tid < 64) {
        const int ln = tid;
        if (ln < L) {                                       // interaction + dropout (final.0)
            const float m = draw(4 * RL + 2 * L + ln);
            cdv[L + ln] = m;
            cdv[ln] = v[ln] * v[L + ln] * m;
        }
        wave_lds_fence();
        if (ln < L) {                                       // final.1 + relu
            float acc = 0.f;
            for (int l = 0; l < L; ++l) acc = fmaf(cdv[l], F1[ln * (L + 1) + l], acc);
            acc += fv[ln];
            fh[ln] = acc > 0.f ? acc : 0.f;
        }
        wave_lds_fence();
        if (ln == 0) {                                      // final.3, bias head, SE
            float acc = 0.f;
            for (int k = 0; k < L; ++k) acc = fmaf(fh[k], fv[L + k], acc);
            const float rating = acc + misc[2];
            const float pred = ((rating + misc[4]) + misc[5]) + misc[3];
            a.pred[b] = pred;
            float g0 = 0.f;
            if (a.y) {
                const float d = pred - y_r;
                a.se[b] = d * d;
                g0 = 2.f * d * a.inv_denom;
            }
            misc[6] = g0;
            if (a.want_grad) a.g[b] = g0;
        }
        if (a.want_grad) {                                  // uniform
            wave_lds_fence();
            const float g = misc[6];
            if (ln < L) {                                   // B1: final.3 / final.1 bias, d fpre
                prow[head_col(a, a.off[NP_F3W] + ln)] = g * fh[ln];
                const float d = fh[ln] > 0.f ? g * fv[L + ln] : 0.f;
                fh[L + ln] = d;
                prow[head_col(a, a.off[NP_F1B] + ln)] = d;
            }
            if (ln == 0) {
                prow[head_col(a, a.off[NP_F3B])] = g;
                prow[head_col(a, a.off[NP_GB])] = g;
            }
            wave_lds_fence();
            if (ln < L) {                                   // B2: d interaction -> d v
                float acc = 0.f;
                for (int k = 0; k < L; ++k) acc = fmaf(fh[L + k], F1[k * (L + 1) + ln], acc);
                const float dcat = acc * cdv[L + ln];
                dv[ln] = dcat * v[L + ln];
                dv[L + ln] = dcat * v[ln];
            }
        }
    } else if (a.want_grad) {
        const int t2 = tid - 64;
        if (t2 == 0) {
            // ids + tags of the self rows: user table entry b <- uid, item table entry b <- iid
            for (int s = 0; s < 2; ++s) {
                const int64_t id = SEL2(a.self_id, s)[b];
                SEL2(a.gid, s)[b] = id;
                SEL2(a.tag, s)[id] = a.now;
            }
        }
        for (int i = t2; i < 2 * R; i += NT - 64) {         // others: side s's ids index table 1-s
            const int s = i >= R, r = i - s * R;
            const int64_t id = SEL2(a.other_id, s)[b * R + r];
            SEL2(a.gid, 1 - s)[nself + b * R + r] = id;
            SEL2(a.tag, 1 - s)[id] = a.now;
        }
    }
    if (!a.want_grad) return;                               // uniform
    __syncthreads();
    HEAD_STAMP(5)
    // ---- B2 (final.1 weight) + B3: self ID rows (compact), d attention weights
    for (int i = tid; i < L * L; i += NT) { const int k = qd(i, invL); prow[head_col(a, a.off[NP_F1W] + i)] = fh[L + k] * cdv[i - k * L]; }
    for (int i = tid; i < L2; i += NT) {
        const int s = i >= L, l = i - s * L;
        SEL2(a.grow, s)[(size_t)b * L + l] = dv[i] * evm[i];
    }
    for (int i = tid; i < 2 * R; i += NT) {
        const int s = i >= R;
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc = fmaf(dv[s * L + l], x[i * L + l], acc);
        da[i] = acc;
    }
    __syncthreads();
    HEAD_STAMP(6)
    // ---- B4 + B5 by one wave per side: softmax backward, the scorer's output layer, d hidden
    if (tid < 128) {
        const int s = tid >> 6, ln = tid & 63;
        const float av = ln < R ? sc[s * R + ln] : 0.f, dav = ln < R ? da[s * R + ln] : 0.f;
        const float dot = wave_sum(av * dav);               // d score = a (d a - <a, d a>)
        if (ln < R) da[s * R + ln] = av * (dav - dot);
        wave_lds_fence();
        if (ln == 0) {
            float acc = 0.f;
            for (int r = 0; r < R; ++r) acc += da[s * R + r];
            prow[head_col(a, (s ? a.off[NP_AIB3] : a.off[NP_AUB3]))] = acc;
        }
        if (ln < L) {
            float acc = 0.f;
            for (int r = 0; r < R; ++r) acc = fmaf(da[s * R + r], h[(s * R + r) * L + ln] * hm[(s * R + r) * L + ln], acc);
            prow[head_col(a, (s ? a.off[NP_AIW3] : a.off[NP_AUW3]) + ln)] = acc;
        }
        for (int rem = ln; rem < RL; rem += 64) {           // d hpre of the side's R x L hidden units
            const int i = s * RL + rem, r = qd(rem, invL), k = rem - r * L;
            dz[i] = h[i] > 0.f ? da[s * R + r] * w3[s * L + k] * hm[i] : 0.f;
        }
    }
    __syncthreads();
    HEAD_STAMP(7)
    // ---- B6: scorer hidden layer gradients, d x (-> d z), d other (compact rows)
    for (int i = tid; i < L2; i += NT) {
        const int s = i >= L, k = i - s * L;
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc += dz[(s * R + r) * L + k];
        prow[head_col(a, (s ? a.off[NP_AIB0] : a.off[NP_AUB0]) + k)] = acc;
    }
    for (int i = tid; i < 2 * L * L2; i += NT) {
        const int s = i >= L * L2, rem = i - s * L * L2, k = qd(rem, invL2), j = rem - k * L2;
        float acc = 0.f;
        for (int r = 0; r < R; ++r) {
            const float c = j < L ? x[(s * R + r) * L + j] : o[(s * R + r) * L + j - L];
            acc = fmaf(dz[(s * R + r) * L + k], c, acc);
        }
        prow[head_col(a, (s ? a.off[NP_AIW0] : a.off[NP_AUW0]) + k * L2 + j)] = acc;
    }
    float dzv[(2 * MR * ML + NT - 1) / NT];        // d z of this thread's elements (kept over the barrier)
#pragma unroll
    for (int it = 0; it < (2 * MR * ML + NT - 1) / NT; ++it) {
        const int i = tid + NT * it;
        dzv[it] = 0.f;
        if (i < 2 * RL) {
            const int s = i >= RL, rem = i - s * RL, r = qd(rem, invL), j = rem - r * L;
            float ax = sc[s * R + r] * dv[s * L + j], ao = 0.f;
            for (int k = 0; k < L; ++k) {
                const float d = dz[(s * R + r) * L + k];
                ax = fmaf(d, W0[(s * L + k) * (L2 + 1) + j], ax);
                ao = fmaf(d, W0[(s * L + k) * (L2 + 1) + L + j], ao);
            }
            dzv[it] = ax * xm[i];
            SEL2(a.grow, 1 - s)[(size_t)(nself + b * R + r) * L + j] = ao;
        }
    }
    __syncthreads();
    HEAD_STAMP(8)
#pragma unroll
    for (int it = 0; it < (2 * MR * ML + NT - 1) / NT; ++it) {
        const int i = tid + NT * it;
        if (i < 2 * RL) dz[i] = dzv[it];
    }
    __syncthreads();
    HEAD_STAMP(9)
    // ---- B7: TextCNN FC gradients, d pooled
    for (int i = tid; i < L2; i += NT) {
        const int s = i >= L, l = i - s * L;
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc += dz[(s * R + r) * L + l];
        prow[head_col(a, (s ? a.off[NP_IFB] : a.off[NP_UFB]) + l)] = acc;
    }
    // Four consecutive filters per thread (NF / 4 = 25 quads): one dz read serves four products, the pooled features come
    // as one 16-byte LDS read, the results leave as one 16-byte store -- 2 (L + R) x 25 items, one pass of the 512 threads
    // at the default shape where the one-output-per-thread form took eight (3.5 us of the launch's 21).  Per output the
    // additions are the same, in the same order.
    constexpr int NQ = NF / 4;
    typedef float hf4 __attribute__((ext_vector_type(4)));
    for (int i = tid; i < 2 * L * NQ; i += NT) {
        const int s = i >= L * NQ, rem = i - s * L * NQ, l = rem / NQ, q = rem - l * NQ;
        hf4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 5
        for (int r = 0; r < R; ++r) {
            const float d = dz[(s * R + r) * L + l];
            const hf4 p4 = *reinterpret_cast<const hf4 *>(P + (s * R + r) * NF + 4 * q);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = fmaf(d, p4[c], acc[c]);
        }
        *reinterpret_cast<hf4 *>(prow + head_col(a, (s ? a.off[NP_IFW] : a.off[NP_UFW]) + l * NF + 4 * q)) = acc;
    }
    for (int i = tid; i < 2 * R * NQ; i += NT) {
        const int s = i >= R * NQ, rem = i - s * R * NQ, r = rem / NQ, q = rem - r * NQ;
        hf4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 5
        for (int l = 0; l < L; ++l) {
            const float d = dz[(s * R + r) * L + l];
            const float *wr = fcw + (s * L + l) * (NF + 1) + 4 * q;         // (rows of NF + 1 floats: four 4-byte reads)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = fmaf(d, wr[c], acc[c]);
        }
        *reinterpret_cast<hf4 *>(SEL2(a.g_pooled, s) + (b * R + r) * NF + 4 * q) = acc;
    }
    HEAD_STAMP(10)
}

static size_t narre_head_lds_bytes(int R, int L) {
    const size_t fl = (size_t)2 * R * NF + 2 * L * (NF + 1) + 2 * L * (2 * L + 1) + L * (L + 1) + 6 * 2 * R * L + 2 * 2 * R +
                      11 * 2 * L + 16;
    return fl * 4;
}

struct NarreWs {
    float *wp[2], *pmax[2]; int *parg[2];
    int *flags[2][2], *slot[2][2], *list[2][2], *count[2][2]; float *ptab[2];
    float *pooled[2]; int *argmax[2]; float *g_pooled[2];
    float *part_w[2], *part_b[2];
    int *tag[2];
    float *part, *grow[2], *g, *mult; int64_t *gid[2];
    size_t bytes;
};

static NarreWs narre_carve(void *ws, int64_t B, int R, int T, int E, int L, int64_t V, int64_t n_users, int64_t n_items) {
    NarreWs w;
    char *p = static_cast<char *>(ws);
    size_t o = 0;
    auto take = [&](size_t nbytes) { char *r = p ? p + o : nullptr; o += align256(nbytes); return r; };
    const int64_t N = B * R;
    const size_t tiles128 = (size_t)(T + 2 + 127) / 128;
    const int ns = textcnn_wgrad_splits(N);
    w.tag[0] = reinterpret_cast<int *>(take((size_t)n_users * 4));          // tags first: persistent, zeroed once
    w.tag[1] = reinterpret_cast<int *>(take((size_t)n_items * 4));
    for (int t = 0; t < 2; ++t)
        for (int bf = 0; bf < 2; ++bf) {                                    // token state: persistent too
            w.flags[bf][t] = reinterpret_cast<int *>(take((size_t)(V + 4) * 4));
            w.count[bf][t] = reinterpret_cast<int *>(take(256));
        }
    for (int t = 0; t < 2; ++t) {
        for (int bf = 0; bf < 2; ++bf) {
            w.slot[bf][t] = reinterpret_cast<int *>(take((size_t)(V + 4) * 4));
            w.list[bf][t] = reinterpret_cast<int *>(take((size_t)proj_row_capacity(N, T, V) * 4));
        }
        w.wp[t] = reinterpret_cast<float *>(take(textcnn_wp_floats(E) * 4));
        w.pmax[t] = reinterpret_cast<float *>(take((size_t)N * tiles128 * NP * 4));
        w.parg[t] = reinterpret_cast<int *>(take((size_t)N * tiles128 * NP * 4));
        w.pooled[t] = reinterpret_cast<float *>(take((size_t)N * NF * 4));
        w.argmax[t] = reinterpret_cast<int *>(take((size_t)N * NF * 4));
        w.g_pooled[t] = reinterpret_cast<float *>(take((size_t)N * NF * 4));
        w.part_w[t] = reinterpret_cast<float *>(take((size_t)ns * NF * 3 * E * 4));
        w.part_b[t] = reinterpret_cast<float *>(take((size_t)ns * NF * 4));
        w.ptab[t] = reinterpret_cast<float *>(take(proj_ptab_floats(N, T, V) * 4));
        w.grow[t] = reinterpret_cast<float *>(take((size_t)B * (1 + R) * L * 4));
        w.gid[t] = reinterpret_cast<int64_t *>(take((size_t)B * (1 + R) * 8));
    }
    const NLayout lay = narre_layout(E, L);
    w.part = reinterpret_cast<float *>(take((size_t)B * head_cols(lay).n * 4));
    w.g = reinterpret_cast<float *>(take((size_t)B * 4));
    w.mult = reinterpret_cast<float *>(take((size_t)B * (4 * R * L + 3 * L) * 4));
    w.bytes = o;
    return w;
}

}  // namespace r4r

using namespace r4r;

extern "C" int r4r_narre_nparam(void) { return NP_COUNT; }

extern "C" int r4r_narre_layout(int E, int L, int64_t *offsets, int64_t *sizes, int64_t *total) {
    R4R_REQUIRE(offsets && sizes && total, "narre_layout: null pointer");
    R4R_REQUIRE(E > 0 && L > 0, "narre_layout: bad sizes");
    const NLayout lay = narre_layout(E, L);
    for (int i = 0; i < NP_COUNT; ++i) { offsets[i] = lay.off[i]; sizes[i] = lay.size[i]; }
    *total = lay.total;
    return R4R_OK;
}

extern "C" size_t r4r_narre_ws_bytes(int64_t B, int R, int T, int E, int L, int64_t V, int64_t n_users, int64_t n_items) {
    if (B < 0 || R <= 0 || T <= 0 || E <= 0 || L <= 0 || V <= 0 || n_users <= 0 || n_items <= 0) return 0;
    return narre_carve(nullptr, B, R, T, E, L, V, n_users, n_items).bytes;
}

// which: 0 dropout multipliers [B, 4RL+3L]; 1 / 2 compact rows of the user / item table [B(1+R), L];
// 3 / 4 their ids (int64); 5 d loss / d pred [B]; 6 + 2 * tower + buffer: that token buffer's
// compaction counter (one int: zero it to discard a prepared-but-unused token state)
extern "C" size_t r4r_narre_ws_offset(int64_t B, int R, int T, int E, int L, int64_t V, int64_t n_users, int64_t n_items,
                                      int which) {
    const NarreWs w = narre_carve(reinterpret_cast<void *>(256), B, R, T, E, L, V, n_users, n_items);
    if (which >= 6 && which < 10)
        return (size_t)(reinterpret_cast<char *>(w.count[(which - 6) & 1][(which - 6) >> 1]) - reinterpret_cast<char *>(256));
    const char *q = which == 0 ? reinterpret_cast<char *>(w.mult) : which == 1 ? reinterpret_cast<char *>(w.grow[0])
                  : which == 2 ? reinterpret_cast<char *>(w.grow[1]) : which == 3 ? reinterpret_cast<char *>(w.gid[0])
                  : which == 4 ? reinterpret_cast<char *>(w.gid[1]) : reinterpret_cast<char *>(w.g);
    return (size_t)(q - reinterpret_cast<char *>(256));
}

extern "C" int r4r_narre_step(const float *table, int64_t V,
                              const int64_t *user_reviews, const int64_t *item_reviews,
                              const int64_t *reviewed_items, const int64_t *users_who_reviewed,
                              const int64_t *uid, const int64_t *iid, const float *y,
                              float *flat_p, float *flat_g, float *flat_m, float *flat_v,
                              const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                              int64_t n_users, int64_t n_items,
                              float *pred, float *se, float *sse_accum, void *ws, size_t ws_bytes,
                              int64_t B, int R, int T, int E, int L,
                              float dropout_p, int training, uint64_t seed, uint64_t offset, float inv_denom,
                              int conv_algo, int token_buffer, int tokens_ready,
                              const int64_t *next_user_reviews, const int64_t *next_item_reviews,
                              float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                              void *stream) {
    R4R_REQUIRE(table && user_reviews && item_reviews && reviewed_items && users_who_reviewed && uid && iid && flat_p &&
                rows_p && pred && ws, "narre_step: null pointer");
    R4R_REQUIRE(V > 0 && B >= 0 && T > 0 && n_users > 0 && n_items > 0, "narre_step: bad sizes");
    R4R_REQUIRE(R > 0 && R <= NARRE_MAX_R, "narre_step: narre_num_reviews %d outside 1..%d", R, NARRE_MAX_R);
    R4R_REQUIRE(L > 0 && L <= NARRE_MAX_L, "narre_step: latent_size %d outside 1..%d", L, NARRE_MAX_L);
    R4R_REQUIRE(E > 0 && E % 4 == 0, "narre_step: word_embed_size %d must be a positive multiple of 4", E);
    const bool train_step = flat_g != nullptr;
    // flat_m == NULL on a training step: gradients only (flat_g and the compact ID rows in the workspace) -- the
    // data-parallel form (r4r_adam_multi + r4r_narre_rows_apply after the exchange)
    const bool apply = flat_m != nullptr;
    // (the fused ID-table role keeps a row's L columns in registers: built for L <= 32; wider rows take the split step)
    R4R_REQUIRE(!(flat_g && apply) || (L <= NR_MAX_L && R <= NR_MAX_R), "narre_step: latent_size %d / narre_num_reviews %d "
                "beyond %d: run the step as gradients (flat_m = NULL) + r4r_adam_multi + r4r_narre_rows_apply_large", L, R, NR_MAX_L);
    R4R_REQUIRE(!train_step || (y && se && adam_step >= 1 && (!apply || (flat_v && rows_m && rows_v))),
                "narre_step: a training step needs ratings, se, gradient buffers and adam_step >= 1 (+ moments to update)");
    R4R_REQUIRE(!y || se, "narre_step: se buffer required when y is given");
    R4R_REQUIRE(!next_user_reviews == !next_item_reviews, "narre_step: next_user_reviews and next_item_reviews go together");
    R4R_REQUIRE(!next_user_reviews || train_step, "narre_step: the next batch's tokens ride on the backward launches");
    R4R_REQUIRE(token_buffer == 0 || token_buffer == 1, "narre_step: token_buffer must be 0 or 1");
    R4R_REQUIRE(adam_step < (1ll << 31), "narre_step: step tag overflow");
    // (the fused ID-table role keeps the entry ids in registers; a gradients-only step has no such role)
    R4R_REQUIRE(!train_step || !apply || B * (1 + R) <= NROW_MAX_ENTRIES, "narre_step: %lld ID entries per table > %d: run the "
                "step as gradients (flat_m = NULL) + r4r_adam_multi + r4r_narre_rows_apply[_large]",
                (long long)(B * (1 + R)), NROW_MAX_ENTRIES);
    R4R_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "narre_step: dropout %f outside [0,1)", (double)dropout_p);
    const int64_t N = B * R;
    R4R_REQUIRE(N * (int64_t)((T + 2 + 127) / 128) < (1ll << 31), "narre_step: grid too large");
    if (ws_bytes < r4r_narre_ws_bytes(B, R, T, E, L, V, n_users, n_items)) {
        set_error("narre_step: workspace %zu < %zu bytes", ws_bytes, r4r_narre_ws_bytes(B, R, T, E, L, V, n_users, n_items));
        return R4R_ERR_WORKSPACE;
    }
    if (B == 0) return R4R_OK;
    hipStream_t st = as_stream(stream);
    const NLayout lay = narre_layout(E, L);
    R4R_REQUIRE(lay.total < (1ll << 31), "narre_step: dense parameter buffer too large");
    const HeadCols hc = head_cols(lay);
    const NarreWs w = narre_carve(ws, B, R, T, E, L, V, n_users, n_items);
    const float *P[NP_COUNT];
    float *G[NP_COUNT];
    for (int i = 0; i < NP_COUNT; ++i) { P[i] = flat_p + lay.off[i]; G[i] = flat_g ? flat_g + lay.off[i] : nullptr; }

    // 1+2: both towers' review documents, one grid
    const int64_t *idx[2] = {user_reviews, item_reviews};
    const int algo = textcnn_pick_algo(conv_algo, N, T, E, NF);
    int tiles;
    if (algo == R4R_CONV_PROJECT) {
        ProjTower pt[2];
        for (int t = 0; t < 2; ++t) {
            pt[t].idx = idx[t];
            pt[t].conv_w = P[t ? NP_ICW : NP_UCW]; pt[t].conv_b = P[t ? NP_ICB : NP_UCB];
            pt[t].flags = w.flags[token_buffer][t]; pt[t].slot = w.slot[token_buffer][t];
            pt[t].list = w.list[token_buffer][t]; pt[t].count = w.count[token_buffer][t];
            pt[t].ptab = w.ptab[t]; pt[t].pmax = w.pmax[t]; pt[t].parg = w.parg[t]; pt[t].wimg = w.wp[t];
        }
        if (!tokens_ready)
            if (int rc = textcnn_proj_tokens_launch(V, pt, 2, N, T, /*zero_state=*/false, st)) return rc;
        if (int rc = textcnn_proj_compute_launch(table, V, pt, 2, N, T, E, NF, st)) return rc;
        tiles = proj_tiles(T);
    } else {
        FwdTower ft[2];
        for (int t = 0; t < 2; ++t) {
            ft[t].idx = idx[t];
            ft[t].conv_w = P[t ? NP_ICW : NP_UCW]; ft[t].conv_b = P[t ? NP_ICB : NP_UCB];
            ft[t].wp = w.wp[t]; ft[t].pmax = w.pmax[t]; ft[t].parg = w.parg[t];
        }
        if (int rc = textcnn_fwd_launch(table, ft, 2, N, T, E, NF, st)) return rc;
        tiles = textcnn_tiles(T);
    }

    // 3: head
    NarreHead h;
    for (int t = 0; t < 2; ++t) {
        h.pmax[t] = w.pmax[t]; h.parg[t] = w.parg[t];
        h.pooled[t] = w.pooled[t]; h.argmax[t] = w.argmax[t]; h.g_pooled[t] = w.g_pooled[t];
        h.emb[t] = reinterpret_cast<const float *>(rows_p[t]);
        h.bias[t] = reinterpret_cast<const float *>(rows_p[2 + t]);
        h.grow[t] = w.grow[t]; h.gid[t] = w.gid[t]; h.tag[t] = w.tag[t];
        R4R_REQUIRE(h.emb[t] && h.bias[t], "narre_step: null table / bias pointer");
    }
    h.self_id[0] = uid; h.self_id[1] = iid;
    h.other_id[0] = reviewed_items; h.other_id[1] = users_who_reviewed;
    h.flat_p = flat_p;
    for (int i = 0; i < NP_COUNT; ++i) h.off[i] = (int)lay.off[i];
    h.col0_lo = (int)hc.lo0; h.col0_n = (int)(hc.hi0 - hc.lo0); h.col1_lo = (int)hc.lo1;
    h.y = y; h.part = w.part; h.g = w.g; h.mult = w.mult; h.pred = pred; h.se = se;
    h.B = B; h.R = R; h.L = L; h.tiles = tiles; h.nhp = hc.n; h.training = training; h.want_grad = train_step;
    h.now = (int)adam_step; h.p_drop = dropout_p; h.inv_denom = inv_denom; h.seed = seed; h.offset = offset;
    const size_t lds = narre_head_lds_bytes(R, L);
    R4R_REQUIRE(lds <= 160 * 1024, "narre_step: R = %d, L = %d need %zu bytes of LDS", R, L, lds);
    const int cls = (R <= 16 && L <= 16) ? 0 : ((R <= 32 && L <= 32) ? 1 : 2);      // instantiation: caps 16 / 32 / 64
    static size_t lds_set[3] = {0, 0, 0};
    if (lds > lds_set[cls]) {
        const void *fn = cls == 0 ? reinterpret_cast<const void *>(narre_head_kernel<16, 16, NHEAD_THREADS>)
                       : cls == 1 ? reinterpret_cast<const void *>(narre_head_kernel<32, 32, NHEAD_THREADS>)
                                  : reinterpret_cast<const void *>(narre_head_kernel<64, 64, NHEAD_THREADS>);
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lds_set[cls] = lds;
    }
    if (cls == 0) narre_head_kernel<16, 16, NHEAD_THREADS><<<(unsigned)B, NHEAD_THREADS, lds, st>>>(h);
    else if (cls == 1) narre_head_kernel<32, 32, NHEAD_THREADS><<<(unsigned)B, NHEAD_THREADS, lds, st>>>(h);
    else narre_head_kernel<64, 64, NHEAD_THREADS><<<(unsigned)B, NHEAD_THREADS, lds, st>>>(h);     // (hyper_params.py:63,78: no bound)
    if (!train_step) return check_launch("narre_step(forward)");

    // 4: conv weight gradients + head-parameter column sums (+ next batch's token marks)
    WgradTower wt[2];
    WgradArgs wa;
    for (int t = 0; t < 2; ++t) {
        wt[t].idx = idx[t]; wt[t].g_pooled = w.g_pooled[t]; wt[t].argmax = w.argmax[t];
        wt[t].part_w = w.part_w[t]; wt[t].part_b = w.part_b[t];
        wt[t].d_w = G[t ? NP_ICW : NP_UCW]; wt[t].d_b = G[t ? NP_ICB : NP_UCB];
    }
    for (int k = 0; k < MAX_TOWERS; ++k) wa.t[k] = wt[k < 2 ? k : 0];
    wa.table = table; wa.N = N; wa.T = T; wa.E = E; wa.F = NF;
    wa.table_bytes = (int64_t)V * E * 4;                   // (the wide wgrad reads the rows through a buffer resource)
    wa.nsplit = textcnn_wgrad_splits(N);
    wa.per_split = (int)cdiv(N, wa.nsplit);
    ColSum cs;
    cs.part = w.part; cs.se = se; cs.flat_g = flat_g; cs.sse_accum = sse_accum; cs.B = B; cs.nhp = hc.n;
    cs.col0_lo = (int)hc.lo0; cs.col0_n = (int)(hc.hi0 - hc.lo0); cs.col1_lo = (int)hc.lo1;
    const int cs_blocks = (hc.n + 1 + CS_COLS - 1) / CS_COLS;
    const bool prefetch = next_user_reviews && algo == R4R_CONV_PROJECT;
    TokenArgs nx{};
    if (prefetch) {
        ProjTower nt[2];
        const int64_t *nidx[2] = {next_user_reviews, next_item_reviews};
        const int ob = token_buffer ^ 1;
        for (int t = 0; t < 2; ++t) {
            nt[t] = ProjTower{};
            nt[t].idx = nidx[t];
            nt[t].flags = w.flags[ob][t]; nt[t].slot = w.slot[ob][t]; nt[t].list = w.list[ob][t]; nt[t].count = w.count[ob][t];
        }
        nx = make_token_args(V, nt, 2, N, T);
    }
    // ID tables + bias vectors: a role of the backward launch
    RowSweep rs;
    float *rp[4], *rm[4], *rv[4];
    for (int k = 0; k < 4; ++k) {
        rp[k] = reinterpret_cast<float *>(rows_p[k]);
        rm[k] = apply ? reinterpret_cast<float *>(rows_m[k]) : nullptr;
        rv[k] = apply ? reinterpret_cast<float *>(rows_v[k]) : nullptr;
        R4R_REQUIRE(rp[k] && (!apply || (rm[k] && rv[k])), "narre_step: row tensor %d: null parameter / moment pointer", k);
    }
    rs.p0 = rp[0]; rs.p1 = rp[1]; rs.p2 = rp[2]; rs.p3 = rp[3];
    rs.m0 = rm[0]; rs.m1 = rm[1]; rs.m2 = rm[2]; rs.m3 = rm[3];
    rs.v0 = rv[0]; rs.v1 = rv[1]; rs.v2 = rv[2]; rs.v3 = rv[3];
    const int64_t numel[4] = {n_users * L, n_items * L, n_users, n_items};
    int64_t begin[5], chunks = 0;
    for (int k = 0; k < 4; ++k) { begin[k] = chunks; chunks += cdiv(numel[k], NROW_CHUNK); }
    rs.n0 = numel[0]; rs.n1 = numel[1]; rs.n2 = numel[2]; rs.n3 = numel[3];
    rs.cb1 = (int)begin[1]; rs.cb2 = (int)begin[2]; rs.cb3 = (int)begin[3]; rs.cb_entries = (int)chunks;
    const int sweep_blocks = (int)chunks;
    chunks += 2 * cdiv(B * (1 + R), 4 * NROW_EPW);          // the entry waves, 4 per workgroup, per table
    R4R_REQUIRE(chunks < (1ll << 31), "narre_step: too many chunks");
    rs.sweep_elsewhere = 1;                                 // entry waves here, the sweep in the reduce launch
    rs.gid0 = w.gid[0]; rs.gid1 = w.gid[1]; rs.grow0 = w.grow[0]; rs.grow1 = w.grow[1]; rs.g = w.g;
    rs.tag0 = w.tag[0]; rs.tag1 = w.tag[1]; rs.entries = B * (1 + R); rs.B = B; rs.L = L; rs.now = (int)adam_step;
    rs.s = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);

    const int packed = 3 * E / 4 <= 64;                     // narrow windows: one wave per filter
    const int gx = packed ? (NF + 3) / 4 : NF;              // slices: ID tables, two towers, the column sums, the marks
    const dim3 bgrid(gx, wa.nsplit, 3 + backward_cs_slices(cs_blocks, gx * wa.nsplit) + (prefetch ? 1 : 0));
    if (!apply)                                             // gradients only: no ID-table role in this launch
        narre_backward_kernel<0><<<dim3(bgrid.x, bgrid.y, bgrid.z - 1), WG_THREADS, 0, st>>>(BackwardArgs{wa, cs, cs_blocks, nx, packed, RowSweep{}, 0, 2});
    else if (L <= 16) narre_backward_kernel<16><<<bgrid, WG_THREADS, 0, st>>>(BackwardArgs{wa, cs, cs_blocks, nx, packed, rs, (int)chunks, 2});
    else narre_backward_kernel<32><<<bgrid, WG_THREADS, 0, st>>>(BackwardArgs{wa, cs, cs_blocks, nx, packed, rs, (int)chunks, 2});

    // 5: wgrad reduce + Adam on the dense parameters (+ next batch's compaction)
    const int red_blocks = (NF * 3 * E + NF + NRED_THREADS - 1) / NRED_THREADS;
    const int comp_blocks = prefetch ? (int)cdiv((V + 3) / 4, NRED_THREADS * compact_groups(V)) : 0;
    DenseAdam opt;
    opt.on = apply ? 1 : 0; opt.p = flat_p; opt.m = flat_m; opt.v = flat_v; opt.g = flat_g;
    opt.lo0 = hc.lo0; opt.hi0 = hc.hi0; opt.lo1 = hc.lo1; opt.hi1 = hc.hi1;
    opt.s = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    const int64_t longest = hc.hi0 - hc.lo0 > hc.hi1 - hc.lo1 ? hc.hi0 - hc.lo0 : hc.hi1 - hc.lo1;
    const int opt_blocks = apply ? (int)cdiv(longest, NRED_THREADS) : 0;
    if (apply)
        narre_reduce_kernel<<<dim3(red_blocks + comp_blocks + opt_blocks + (sweep_blocks + 1) / 2, 2), NRED_THREADS, 0, st>>>(
            wa, red_blocks, comp_blocks, nx, opt, opt_blocks, rs);
    else
        narre_reduce_kernel<<<dim3(red_blocks + comp_blocks + opt_blocks, 2), NRED_THREADS, 0, st>>>(wa, red_blocks, comp_blocks,
                                                                                                    nx, opt);

    return check_launch("narre_step");
}

// Data parallel: the ID tables + bias vectors from ALL ranks' compact entries (gathered by the caller
// in one fixed order; ids -1 pad ragged shards), after a gradients-only r4r_narre_step (flat_m ==
// NULL).  gid0 / gid1 [entries]: entry ids of the user / item table; grow0 / grow1 [entries, L]: their
// gradient rows; g_entry [entries]: an entry's bias gradient (d loss / d pred for a rating's own
// user / item entry, 0 for a neighbour entry).  `ws` and the shape arguments are the step's: the
// row tags live there.  entries <= 16384.
namespace r4r {
__global__ void narre_tag_rows_kernel(const int64_t *gid0, const int64_t *gid1, int64_t n, int *tag0, int *tag1, int now) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    if (gid0[e] >= 0) tag0[gid0[e]] = now;
    if (gid1[e] >= 0) tag1[gid1[e]] = now;
}
}  // namespace r4r

extern "C" int r4r_narre_rows_apply(const int64_t *gid0, const int64_t *gid1, const float *grow0, const float *grow1,
                                    const float *g_entry, int64_t entries,
                                    const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                                    int64_t n_users, int64_t n_items, void *ws, size_t ws_bytes,
                                    int64_t B, int R, int T, int E, int L, int64_t V,
                                    float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                                    void *stream) {
    R4R_REQUIRE(gid0 && gid1 && grow0 && grow1 && g_entry && rows_p && rows_m && rows_v && ws, "narre_rows_apply: null pointer");
    R4R_REQUIRE(entries >= 0 && entries <= NROW_DP_MAX_ENTRIES, "narre_rows_apply: %lld entries outside 0..%d",
                (long long)entries, NROW_DP_MAX_ENTRIES);
    R4R_REQUIRE(L > 0 && L <= NR_MAX_L, "narre_rows_apply: latent_size %d outside 1..%d", L, NR_MAX_L);
    R4R_REQUIRE(adam_step >= 1 && adam_step < (1ll << 31), "narre_rows_apply: bad adam_step");
    if (ws_bytes < r4r_narre_ws_bytes(B, R, T, E, L, V, n_users, n_items)) {
        set_error("narre_rows_apply: workspace %zu < %zu bytes", ws_bytes, r4r_narre_ws_bytes(B, R, T, E, L, V, n_users, n_items));
        return R4R_ERR_WORKSPACE;
    }
    if (entries == 0) return R4R_OK;
    hipStream_t st = as_stream(stream);
    const NarreWs w = narre_carve(ws, B, R, T, E, L, V, n_users, n_items);
    RowSweep rs;
    float *rp[4], *rm[4], *rv[4];
    for (int k = 0; k < 4; ++k) {
        rp[k] = reinterpret_cast<float *>(rows_p[k]); rm[k] = reinterpret_cast<float *>(rows_m[k]);
        rv[k] = reinterpret_cast<float *>(rows_v[k]);
        R4R_REQUIRE(rp[k] && rm[k] && rv[k], "narre_rows_apply: row tensor %d: null parameter / moment pointer", k);
    }
    rs.p0 = rp[0]; rs.p1 = rp[1]; rs.p2 = rp[2]; rs.p3 = rp[3];
    rs.m0 = rm[0]; rs.m1 = rm[1]; rs.m2 = rm[2]; rs.m3 = rm[3];
    rs.v0 = rv[0]; rs.v1 = rv[1]; rs.v2 = rv[2]; rs.v3 = rv[3];
    const int64_t numel[4] = {n_users * L, n_items * L, n_users, n_items};
    int64_t begin[5], chunks = 0;
    for (int k = 0; k < 4; ++k) { begin[k] = chunks; chunks += cdiv(numel[k], NROW_CHUNK); }
    rs.n0 = numel[0]; rs.n1 = numel[1]; rs.n2 = numel[2]; rs.n3 = numel[3];
    rs.cb1 = (int)begin[1]; rs.cb2 = (int)begin[2]; rs.cb3 = (int)begin[3]; rs.cb_entries = (int)chunks;
    chunks += 2 * cdiv(entries, 4 * NROW_EPW);
    R4R_REQUIRE(chunks < (1ll << 31), "narre_rows_apply: too many chunks");
    rs.sweep_elsewhere = 0;
    rs.gid0 = gid0; rs.gid1 = gid1; rs.grow0 = grow0; rs.grow1 = grow1; rs.g = g_entry;
    rs.tag0 = w.tag[0]; rs.tag1 = w.tag[1]; rs.entries = entries; rs.B = entries; rs.L = L; rs.now = (int)adam_step;
    rs.s = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    narre_tag_rows_kernel<<<(unsigned)cdiv(entries, 256), 256, 0, st>>>(gid0, gid1, entries, w.tag[0], w.tag[1], (int)adam_step);
    const size_t lds = (size_t)entries * sizeof(int);
    static size_t attr16 = 0, attr32 = 0;
    if (L <= 16) {
        if (attr16 < lds) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(narre_rows_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr16 = lds; }
        narre_rows_kernel<16><<<(unsigned)chunks, NROW_THREADS, lds, st>>>(rs);
    } else {
        if (attr32 < lds) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(narre_rows_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr32 = lds; }
        narre_rows_kernel<32><<<(unsigned)chunks, NROW_THREADS, lds, st>>>(rs);
    }
    return check_launch("narre_rows_apply");
}

// ---- the same update with ONE exchange and no glue launches: every rank packs its compact entries into one block
// (r4r_narre_dp_block: rows_device.h mf_block's layout over E_pad = B_pad (1 + R) entries of width L -- ids as int32,
// an entry's bias gradient, the two tables' gradient rows; the B_pad self entries first, then a rating's R neighbour
// entries, ids -1 = padding), ONE all_gather moves the blocks, r4r_narre_rows_apply_blocks tags the rows and runs the
// entry waves + the sweep straight over the gathered blocks, entries in (rank, in-block) order on every rank.
namespace r4r {
__global__ __launch_bounds__(256) void narre_dp_block_kernel(const int64_t *gid0, const int64_t *gid1, const float *grow0,
                                                             const float *grow1, const float *g, int *o_gid0, int *o_gid1,
                                                             float *o_g, float *o_grow0, float *o_grow1, int64_t B,
                                                             int64_t B_pad, int R, int L) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // element of the [E_pad, L] row arrays
    const int64_t E_pad = B_pad * (1 + R);
    if (i >= E_pad * L) return;
    const int64_t e = i / L;
    const int col = (int)(i - e * L);
    // block entry e -> this rank's entry: the self entries [0, B) stay, rating b's neighbour j sits at B + b R + j
    int64_t src = -1;
    if (e < B_pad) { if (e < B) src = e; }
    else { const int64_t b = (e - B_pad) / R; if (b < B) src = B + (e - B_pad); }
    o_grow0[i] = src >= 0 ? grow0[src * L + col] : 0.f;
    o_grow1[i] = src >= 0 ? grow1[src * L + col] : 0.f;
    if (col == 0) {
        o_gid0[e] = src >= 0 ? (int)gid0[src] : -1;
        o_gid1[e] = src >= 0 ? (int)gid1[src] : -1;
        o_g[e] = (src >= 0 && e < B) ? g[e] : 0.f;          // only a rating's own user / item entry carries a bias gradient
    }
}
__global__ void narre_tag_rows_blocks_kernel(RowSweep w, int *tag0, int *tag1) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= w.entries) return;
    const int64_t o = nrow_eoff<true>(w, e, 1);
    const int a = w.gid32_0[o], b = w.gid32_1[o];
    if (a >= 0) tag0[a] = w.now;
    if (b >= 0) tag1[b] = w.now;
}
}  // namespace r4r

extern "C" size_t r4r_narre_dp_block_bytes(int64_t B_pad, int R, int L) {
    return (B_pad < 0 || R < 0 || L < 1) ? 0 : mf_block(B_pad * (1 + R), L).bytes;
}

extern "C" int r4r_narre_dp_block(void *ws, size_t ws_bytes, int64_t B, int R, int T, int E, int L, int64_t V,
                                  int64_t n_users, int64_t n_items, void *block, int64_t B_pad, void *stream) {
    R4R_REQUIRE(block && B >= 0 && B_pad >= B && R >= 1 && L >= 1, "narre_dp_block: null block, B_pad < B, or bad R / L");
    R4R_REQUIRE(B == 0 || ws, "narre_dp_block: null workspace");
    if (B_pad == 0) return R4R_OK;
    NarreWs w{};
    if (B > 0) {
        if (ws_bytes < r4r_narre_ws_bytes(B, R, T, E, L, V, n_users, n_items)) {
            set_error("narre_dp_block: workspace %zu < %zu bytes", ws_bytes, r4r_narre_ws_bytes(B, R, T, E, L, V, n_users, n_items));
            return R4R_ERR_WORKSPACE;
        }
        w = narre_carve(ws, B, R, T, E, L, V, n_users, n_items);
    }
    const int64_t E_pad = B_pad * (1 + R);
    const MfBlock k = mf_block(E_pad, L);
    char *b = static_cast<char *>(block);
    narre_dp_block_kernel<<<(unsigned)cdiv(E_pad * L, 256), 256, 0, as_stream(stream)>>>(
        w.gid[0], w.gid[1], w.grow[0], w.grow[1], w.g, reinterpret_cast<int *>(b + k.uid), reinterpret_cast<int *>(b + k.iid),
        reinterpret_cast<float *>(b + k.g), reinterpret_cast<float *>(b + k.gu), reinterpret_cast<float *>(b + k.gi), B, B_pad, R, L);
    return check_launch("narre_dp_block");
}

extern "C" int r4r_narre_rows_apply_blocks(const void *blocks, int world, int64_t B_pad,
                                           const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                                           int64_t n_users, int64_t n_items, void *ws, size_t ws_bytes,
                                           int64_t B, int R, int T, int E, int L, int64_t V,
                                           float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                                           void *stream) {
    R4R_REQUIRE(blocks && rows_p && rows_m && rows_v && ws, "narre_rows_apply_blocks: null pointer");
    R4R_REQUIRE(world >= 1 && B_pad >= 0 && R >= 1, "narre_rows_apply_blocks: bad sizes");
    const int64_t E_pad = B_pad * (1 + R), entries = (int64_t)world * E_pad;
    R4R_REQUIRE(entries <= NROW_DP_MAX_ENTRIES, "narre_rows_apply_blocks: %lld entries outside 0..%d", (long long)entries,
                NROW_DP_MAX_ENTRIES);
    R4R_REQUIRE(L > 0 && L <= NR_MAX_L, "narre_rows_apply_blocks: latent_size %d outside 1..%d", L, NR_MAX_L);
    R4R_REQUIRE(adam_step >= 1 && adam_step < (1ll << 31), "narre_rows_apply_blocks: bad adam_step");
    if (ws_bytes < r4r_narre_ws_bytes(B, R, T, E, L, V, n_users, n_items)) {
        set_error("narre_rows_apply_blocks: workspace %zu < %zu bytes", ws_bytes, r4r_narre_ws_bytes(B, R, T, E, L, V, n_users, n_items));
        return R4R_ERR_WORKSPACE;
    }
    if (entries == 0) return R4R_OK;
    hipStream_t st = as_stream(stream);
    const NarreWs w = narre_carve(ws, B, R, T, E, L, V, n_users, n_items);
    RowSweep rs;
    float *rp[4], *rm[4], *rv[4];
    for (int k = 0; k < 4; ++k) {
        rp[k] = reinterpret_cast<float *>(rows_p[k]); rm[k] = reinterpret_cast<float *>(rows_m[k]);
        rv[k] = reinterpret_cast<float *>(rows_v[k]);
        R4R_REQUIRE(rp[k] && rm[k] && rv[k], "narre_rows_apply_blocks: row tensor %d: null parameter / moment pointer", k);
    }
    rs.p0 = rp[0]; rs.p1 = rp[1]; rs.p2 = rp[2]; rs.p3 = rp[3];
    rs.m0 = rm[0]; rs.m1 = rm[1]; rs.m2 = rm[2]; rs.m3 = rm[3];
    rs.v0 = rv[0]; rs.v1 = rv[1]; rs.v2 = rv[2]; rs.v3 = rv[3];
    const int64_t numel[4] = {n_users * L, n_items * L, n_users, n_items};
    int64_t begin[5], chunks = 0;
    for (int k = 0; k < 4; ++k) { begin[k] = chunks; chunks += cdiv(numel[k], NROW_CHUNK); }
    rs.n0 = numel[0]; rs.n1 = numel[1]; rs.n2 = numel[2]; rs.n3 = numel[3];
    rs.cb1 = (int)begin[1]; rs.cb2 = (int)begin[2]; rs.cb3 = (int)begin[3]; rs.cb_entries = (int)chunks;
    chunks += 2 * cdiv(entries, 4 * NROW_EPW);
    R4R_REQUIRE(chunks < (1ll << 31), "narre_rows_apply_blocks: too many chunks");
    rs.sweep_elsewhere = 0;
    const MfBlock k = mf_block(E_pad, L);
    const char *b0 = static_cast<const char *>(blocks);
    rs.gid0 = nullptr; rs.gid1 = nullptr;
    rs.gid32_0 = reinterpret_cast<const int *>(b0 + k.uid); rs.gid32_1 = reinterpret_cast<const int *>(b0 + k.iid);
    rs.g = reinterpret_cast<const float *>(b0 + k.g);
    rs.grow0 = reinterpret_cast<const float *>(b0 + k.gu); rs.grow1 = reinterpret_cast<const float *>(b0 + k.gi);
    rs.E_pad = E_pad; rs.blk_units = (int64_t)(k.bytes / 4);
    rs.tag0 = w.tag[0]; rs.tag1 = w.tag[1]; rs.entries = entries; rs.B = entries; rs.L = L; rs.now = (int)adam_step;
    rs.s = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    narre_tag_rows_blocks_kernel<<<(unsigned)cdiv(entries, 256), 256, 0, st>>>(rs, w.tag[0], w.tag[1]);
    const size_t lds = (size_t)entries * sizeof(int);
    static size_t attr16 = 0, attr32 = 0;
    if (L <= 16) {
        if (attr16 < lds) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(narre_rows_kernel<16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr16 = lds; }
        narre_rows_kernel<16, true><<<(unsigned)chunks, NROW_THREADS, lds, st>>>(rs);
    } else {
        if (attr32 < lds) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(narre_rows_kernel<32, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr32 = lds; }
        narre_rows_kernel<32, true><<<(unsigned)chunks, NROW_THREADS, lds, st>>>(rs);
    }
    return check_launch("narre_rows_apply_blocks");
}

// Any number of entries: the named rows by the bucketed entry waves of rows_large.hip, the others by the tagged sweep.
extern "C" int r4r_narre_rows_apply_large(const int64_t *gid0, const int64_t *gid1, const float *grow0, const float *grow1,
                                          const float *g_entry, int64_t entries,
                                          const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                                          int64_t n_users, int64_t n_items, void *ws, size_t ws_bytes,
                                          int64_t B, int R, int T, int E, int L, int64_t V,
                                          float lr, double beta1, double beta2, float eps, float weight_decay,
                                          int64_t adam_step, void *scratch, size_t scratch_bytes, void *stream) {
    R4R_REQUIRE(gid0 && gid1 && grow0 && grow1 && g_entry && rows_p && rows_m && rows_v && ws && scratch,
                "narre_rows_apply_large: null pointer");
    R4R_REQUIRE(entries >= 0, "narre_rows_apply_large: negative entry count");
    R4R_REQUIRE(L > 0 && L <= NARRE_MAX_L, "narre_rows_apply_large: latent_size %d outside 1..%d", L, NARRE_MAX_L);
    R4R_REQUIRE(adam_step >= 1 && adam_step < (1ll << 31), "narre_rows_apply_large: bad adam_step");
    if (ws_bytes < r4r_narre_ws_bytes(B, R, T, E, L, V, n_users, n_items)) {
        set_error("narre_rows_apply_large: workspace %zu < %zu bytes", ws_bytes, r4r_narre_ws_bytes(B, R, T, E, L, V, n_users, n_items));
        return R4R_ERR_WORKSPACE;
    }
    if (entries == 0) return R4R_OK;
    hipStream_t st = as_stream(stream);
    const NarreWs w = narre_carve(ws, B, R, T, E, L, V, n_users, n_items);
    float *rp[4], *rm[4], *rv[4];
    for (int k = 0; k < 4; ++k) {
        rp[k] = reinterpret_cast<float *>(rows_p[k]); rm[k] = reinterpret_cast<float *>(rows_m[k]);
        rv[k] = reinterpret_cast<float *>(rows_v[k]);
        R4R_REQUIRE(rp[k] && rm[k] && rv[k], "narre_rows_apply_large: row tensor %d: null parameter / moment pointer", k);
    }
    narre_tag_rows_kernel<<<(unsigned)cdiv(entries, 256), 256, 0, st>>>(gid0, gid1, entries, w.tag[0], w.tag[1], (int)adam_step);
    const int64_t nrow[2] = {n_users, n_items};
    const int64_t *gid[2] = {gid0, gid1};
    const float *grow[2] = {grow0, grow1};
    for (int t = 0; t < 2; ++t)                               // (one scratch buffer: the two chains run back to back)
        if (int rc = r4r_rows_apply_large(gid[t], grow[t], g_entry, entries, L, rp[t], rm[t], rv[t], rp[2 + t], rm[2 + t],
                                          rv[2 + t], nrow[t], scratch, scratch_bytes, lr, beta1, beta2, eps, weight_decay,
                                          adam_step, stream))
            return rc;
    // the rows no entry names: the sweep workgroups alone
    RowSweep rs{};
    rs.p0 = rp[0]; rs.p1 = rp[1]; rs.p2 = rp[2]; rs.p3 = rp[3];
    rs.m0 = rm[0]; rs.m1 = rm[1]; rs.m2 = rm[2]; rs.m3 = rm[3];
    rs.v0 = rv[0]; rs.v1 = rv[1]; rs.v2 = rv[2]; rs.v3 = rv[3];
    const int64_t numel[4] = {n_users * L, n_items * L, n_users, n_items};
    int64_t begin[5], chunks = 0;
    for (int k = 0; k < 4; ++k) { begin[k] = chunks; chunks += cdiv(numel[k], NROW_CHUNK); }
    R4R_REQUIRE(chunks < (1ll << 31), "narre_rows_apply_large: too many chunks");
    rs.n0 = numel[0]; rs.n1 = numel[1]; rs.n2 = numel[2]; rs.n3 = numel[3];
    rs.cb1 = (int)begin[1]; rs.cb2 = (int)begin[2]; rs.cb3 = (int)begin[3]; rs.cb_entries = (int)chunks;
    rs.sweep_elsewhere = 0;
    rs.tag0 = w.tag[0]; rs.tag1 = w.tag[1]; rs.entries = 0; rs.B = 0; rs.L = L; rs.now = (int)adam_step;
    rs.s = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    if (L <= 16) narre_rows_kernel<16><<<(unsigned)chunks, NROW_THREADS, 0, st>>>(rs);
    else narre_rows_kernel<32><<<(unsigned)chunks, NROW_THREADS, 0, st>>>(rs);
    return check_launch("narre_rows_apply_large");
}
