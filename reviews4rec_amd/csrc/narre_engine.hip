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
#include <stdlib.h>

#include "adam_device.h"
#include "textcnn.h"
#include "tokens_device.h"
#include "wgrad_device.h"
#include "rows_device.h"

namespace r4r {

constexpr int NF = 100;                // conv filters (common_pytorch_models.py:11)
#ifndef R4R_NHEAD_THREADS
#define R4R_NHEAD_THREADS 512
#endif
constexpr int NHEAD_THREADS = R4R_NHEAD_THREADS;   // threads of the per-rating head workgroup
constexpr int NR_MAX_L = 32, NR_MAX_R = 32;        // hard limits; the kernels are instantiated for <= 16 and <= 32

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
        o += (sz[i] + 3) & ~(int64_t)3;          // 16-byte aligned slots; pad floats stay 0 forever
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
template <int MR, int ML, int NT>
__global__ __launch_bounds__(NT) void narre_head_kernel(NarreHead a) {
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
    constexpr int PREG = (2 * MR * NF + NT - 1) / NT, WREG = (2 * ML * NF + NT - 1) / NT,
                  AREG = (2 * ML * 2 * ML + NT - 1) / NT;
    float pv[PREG], wv[WREG], av[AREG];
    int pa[PREG];
    const bool one_tile = a.tiles == 1;
#pragma unroll
    for (int u = 0; u < PREG; ++u) {
        pv[u] = 0.f; pa[u] = 0;
        if (one_tile && NT * u < 2 * R * NF) {             // uniform: rounds past the end cost nothing
            const int i = min(tid + NT * u, 2 * R * NF - 1);
            const int s = i >= R * NF, rem = i - s * R * NF, rr = rem / NF, f = rem - rr * NF;
            const size_t q = ((size_t)(b * R + rr) * a.tiles) * NP + f;
            pv[u] = a.pmax[s][q];
            pa[u] = a.parg[s][q];
        }
    }
#pragma unroll
    for (int u = 0; u < WREG; ++u) {
        wv[u] = 0.f;
        if (NT * u < 2 * L * NF) {
            const int i = min(tid + NT * u, 2 * L * NF - 1);
            const int s = i >= L * NF;
            wv[u] = fp[a.off[s ? NP_IFW : NP_UFW] + i - s * L * NF];
        }
    }
#pragma unroll
    for (int u = 0; u < AREG; ++u) {
        av[u] = 0.f;
        if (NT * u < 2 * L * L2) {
            const int i = min(tid + NT * u, 2 * L * L2 - 1);
            const int s = i >= L * L2;
            av[u] = fp[a.off[s ? NP_AIW0 : NP_AUW0] + i - s * L * L2];
        }
    }
    // the small reads ride in the same round trip; the ID vectors need their ids first (round 2)
    constexpr int FREG = (ML * ML + NT - 1) / NT, OREG = (2 * MR * ML + NT - 1) / NT;
    float f1v[FREG], ov[OREG];
    int64_t oid[OREG];
#pragma unroll
    for (int u = 0; u < FREG; ++u) f1v[u] = (NT * u < L * L) ? fp[a.off[NP_F1W] + min(tid + NT * u, L * L - 1)] : 0.f;
#pragma unroll
    for (int u = 0; u < OREG; ++u) {
        oid[u] = 0;
        if (NT * u < 2 * RL) {
            const int i = min(tid + NT * u, 2 * RL - 1);
            const int s = i >= RL, rem = i - s * RL, r = qd(rem, invL);
            oid[u] = a.other_id[s][b * R + r];
        }
    }
    const int ti = min(tid, L2 - 1), ts = ti >= L, tl = ti - ts * L;          // this thread's (side, l) for the [2][L] vectors
    const float fcb_r = fp[a.off[ts ? NP_IFB : NP_UFB] + tl], b0_r = fp[a.off[ts ? NP_AIB0 : NP_AUB0] + tl],
                w3_r = fp[a.off[ts ? NP_AIW3 : NP_AUW3] + tl];
    const int64_t sid_r = a.self_id[ts][b];
    const float f1b_r = fp[a.off[NP_F1B] + min(tid, L - 1)], f3w_r = fp[a.off[NP_F3W] + min(tid, L - 1)];
    const float m0 = fp[a.off[NP_AUB3]], m1 = fp[a.off[NP_AIB3]], m2 = fp[a.off[NP_F3B]], m3 = fp[a.off[NP_GB]];
    const int64_t sid0 = a.self_id[0][b], sid1 = a.self_id[1][b];
    // round 2: the reads that depend on ids
#pragma unroll
    for (int u = 0; u < OREG; ++u) {
        ov[u] = 0.f;
        if (NT * u < 2 * RL) {
            const int i = min(tid + NT * u, 2 * RL - 1);
            const int s = i >= RL, rem = i - s * RL, r = qd(rem, invL), l = rem - r * L;
            ov[u] = a.emb[1 - s][oid[u] * L + l];
        }
    }
    const float ev_r = a.emb[ts][sid_r * L + tl];
    const float ub_r = a.bias[0][sid0], ib_r = a.bias[1][sid1];
#pragma unroll
    for (int u = 0; u < WREG; ++u) {
        const int i = tid + NT * u;
        if (i < 2 * L * NF) {
            const int s = i >= L * NF, r = i - s * L * NF, l = r / NF;
            fcw[(s * L + l) * (NF + 1) + r - l * NF] = wv[u];
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
        fcb[tid] = fcb_r; b0[tid] = b0_r; w3[tid] = w3_r; ev[tid] = ev_r;
        evm[tid] = draw(4 * RL + ts * L + tl);
    }
    if (tid < L) { fv[tid] = f1b_r; fv[L + tid] = f3w_r; }
    if (tid == 0) { misc[0] = m0; misc[1] = m1; misc[2] = m2; misc[3] = m3; misc[4] = ub_r; misc[5] = ib_r; }
#pragma unroll
    for (int u = 0; u < OREG; ++u) {                        // other side's ID vectors: side s reads table 1-s
        const int i = tid + NT * u;
        if (i < 2 * RL) o[i] = ov[u];
    }
    if (one_tile) {                                         // pool finish of a one-tile document: relu + argmax
#pragma unroll
        for (int u = 0; u < PREG; ++u) {
            const int i = tid + NT * u;
            if (i < 2 * R * NF) {
                const int s = i >= R * NF, rem = i - s * R * NF, rr = rem / NF, f = rem - rr * NF;
                const int64_t n = b * R + rr;
                float best = pv[u];
                int bp = pa[u];
                if (!(best > 0.f)) { best = 0.f; bp = -1; }
                P[i] = best;
                a.pooled[s][n * NF + f] = best;
                a.argmax[s][n * NF + f] = bp;
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
                const float val = a.pmax[s][q];
                if (val > best) { best = val; bp = a.parg[s][q]; }
            }
            if (!(best > 0.f)) { best = 0.f; bp = -1; }
            P[i] = best;
            a.pooled[s][n * NF + f] = best;
            a.argmax[s][n * NF + f] = bp;
        }
    }
    __syncthreads();
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
    __syncthreads();
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
    // ---- S3: scores
    for (int i = tid; i < 2 * R; i += NT) {
        const int s = i >= R;
        float acc = 0.f;
        for (int k = 0; k < L; ++k) acc = fmaf(h[i * L + k] * hm[i * L + k], w3[s * L + k], acc);
        sc[i] = acc + misc[s];
    }
    __syncthreads();
    // ---- S4: softmax over the R reviews of a side (pads are not masked, like the reference)
    if (tid < 2) {
        float mx = -INFINITY, den = 0.f;
        for (int r = 0; r < R; ++r) mx = fmaxf(mx, sc[tid * R + r]);
        for (int r = 0; r < R; ++r) { const float e = expf(sc[tid * R + r] - mx); sc[tid * R + r] = e; den += e; }
        for (int r = 0; r < R; ++r) sc[tid * R + r] /= den;
    }
    __syncthreads();
    // ---- S5: attended review vector + the ID vector (NARRE.py:110-111)
    for (int i = tid; i < L2; i += NT) {
        const int s = i >= L, l = i - s * L;
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc = fmaf(sc[s * R + r], x[(s * R + r) * L + l], acc);
        v[i] = acc + ev[i] * evm[i];
    }
    __syncthreads();
    // ---- S6: interaction + dropout (final.0)
    for (int i = tid; i < L; i += NT) {
        const float m = draw(4 * RL + 2 * L + i);
        cdv[L + i] = m;
        cdv[i] = v[i] * v[L + i] * m;
    }
    __syncthreads();
    // ---- S7: final.1 + relu
    for (int k = tid; k < L; k += NT) {
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc = fmaf(cdv[l], F1[k * (L + 1) + l], acc);
        acc += fv[k];
        fh[k] = acc > 0.f ? acc : 0.f;
    }
    __syncthreads();
    // ---- S8: final.3, bias head, SE
    if (tid == 0) {
        float acc = 0.f;
        for (int k = 0; k < L; ++k) acc = fmaf(fh[k], fv[L + k], acc);
        const float rating = acc + misc[2];
        const float pred = ((rating + misc[4]) + misc[5]) + misc[3];
        a.pred[b] = pred;
        float g = 0.f;
        if (a.y) {
            const float d = pred - a.y[b];
            a.se[b] = d * d;
            g = 2.f * d * a.inv_denom;
        }
        misc[6] = g;
        if (a.want_grad) a.g[b] = g;
    }
    if (!a.want_grad) return;                               // uniform
    __syncthreads();
    const float g = misc[6];
    float *prow = a.part + (size_t)b * a.nhp;
    const int64_t nself = a.B;                              // entries [0, B): self rows, then B*R others
    // ---- B1: final.3 / final.1 bias, d fpre
    for (int k = tid; k < L; k += NT) {
        prow[head_col(a, a.off[NP_F3W] + k)] = g * fh[k];
        const float d = fh[k] > 0.f ? g * fv[L + k] : 0.f;
        fh[L + k] = d;
        prow[head_col(a, a.off[NP_F1B] + k)] = d;
    }
    if (tid == 0) {
        prow[head_col(a, a.off[NP_F3B])] = g;
        prow[head_col(a, a.off[NP_GB])] = g;
        // ids + tags of the self rows: user table entry b <- uid, item table entry b <- iid
        for (int s = 0; s < 2; ++s) {
            const int64_t id = a.self_id[s][b];
            a.gid[s][b] = id;
            a.tag[s][id] = a.now;
        }
    }
    for (int i = tid; i < 2 * R; i += NT) {                // others: side s's ids index table 1-s
        const int s = i >= R, r = i - s * R;
        const int64_t id = a.other_id[s][b * R + r];
        a.gid[1 - s][nself + b * R + r] = id;
        a.tag[1 - s][id] = a.now;
    }
    __syncthreads();
    // ---- B2: final.1 weight, d interaction -> d v
    for (int i = tid; i < L * L; i += NT) { const int k = qd(i, invL); prow[head_col(a, a.off[NP_F1W] + i)] = fh[L + k] * cdv[i - k * L]; }
    for (int l = tid; l < L; l += NT) {
        float acc = 0.f;
        for (int k = 0; k < L; ++k) acc = fmaf(fh[L + k], F1[k * (L + 1) + l], acc);
        const float dcat = acc * cdv[L + l];
        dv[l] = dcat * v[L + l];
        dv[L + l] = dcat * v[l];
    }
    __syncthreads();
    // ---- B3: self ID rows (compact), d attention weights
    for (int i = tid; i < L2; i += NT) {
        const int s = i >= L, l = i - s * L;
        a.grow[s][(size_t)b * L + l] = dv[i] * evm[i];
    }
    for (int i = tid; i < 2 * R; i += NT) {
        const int s = i >= R;
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc = fmaf(dv[s * L + l], x[i * L + l], acc);
        da[i] = acc;
    }
    __syncthreads();
    // ---- B4: softmax backward
    if (tid < 2) {
        float dot = 0.f;
        for (int r = 0; r < R; ++r) dot = fmaf(sc[tid * R + r], da[tid * R + r], dot);
        misc[7 + tid] = dot;
    }
    __syncthreads();
    for (int i = tid; i < 2 * R; i += NT) da[i] = sc[i] * (da[i] - misc[7 + (i >= R)]);   // d score
    __syncthreads();
    // ---- B5: scorer output layer, d hidden
    if (tid < 2) {
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc += da[tid * R + r];
        prow[head_col(a, a.off[tid ? NP_AIB3 : NP_AUB3])] = acc;
    }
    for (int i = tid; i < L2; i += NT) {
        const int s = i >= L, k = i - s * L;
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc = fmaf(da[s * R + r], h[(s * R + r) * L + k] * hm[(s * R + r) * L + k], acc);
        prow[head_col(a, a.off[s ? NP_AIW3 : NP_AUW3] + k)] = acc;
    }
    for (int i = tid; i < 2 * RL; i += NT) {
        const int s = i >= RL, rem = i - s * RL, r = qd(rem, invL), k = rem - r * L;
        dz[i] = h[i] > 0.f ? da[s * R + r] * w3[s * L + k] * hm[i] : 0.f;     // d hpre
    }
    __syncthreads();
    // ---- B6: scorer hidden layer gradients, d x (-> d z), d other (compact rows)
    for (int i = tid; i < L2; i += NT) {
        const int s = i >= L, k = i - s * L;
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc += dz[(s * R + r) * L + k];
        prow[head_col(a, a.off[s ? NP_AIB0 : NP_AUB0] + k)] = acc;
    }
    for (int i = tid; i < 2 * L * L2; i += NT) {
        const int s = i >= L * L2, rem = i - s * L * L2, k = qd(rem, invL2), j = rem - k * L2;
        float acc = 0.f;
        for (int r = 0; r < R; ++r) {
            const float c = j < L ? x[(s * R + r) * L + j] : o[(s * R + r) * L + j - L];
            acc = fmaf(dz[(s * R + r) * L + k], c, acc);
        }
        prow[head_col(a, a.off[s ? NP_AIW0 : NP_AUW0] + k * L2 + j)] = acc;
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
            a.grow[1 - s][(size_t)(nself + b * R + r) * L + j] = ao;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < (2 * MR * ML + NT - 1) / NT; ++it) {
        const int i = tid + NT * it;
        if (i < 2 * RL) dz[i] = dzv[it];
    }
    __syncthreads();
    // ---- B7: TextCNN FC gradients, d pooled
    for (int i = tid; i < L2; i += NT) {
        const int s = i >= L, l = i - s * L;
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc += dz[(s * R + r) * L + l];
        prow[head_col(a, a.off[s ? NP_IFB : NP_UFB] + l)] = acc;
    }
    for (int i = tid; i < 2 * L * NF; i += NT) {
        const int s = i >= L * NF, rem = i - s * L * NF, l = rem / NF, f = rem - l * NF;
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc = fmaf(dz[(s * R + r) * L + l], P[(s * R + r) * NF + f], acc);
        prow[head_col(a, a.off[s ? NP_IFW : NP_UFW] + l * NF + f)] = acc;
    }
    for (int i = tid; i < 2 * R * NF; i += NT) {
        const int s = i >= R * NF, rem = i - s * R * NF, r = rem / NF, f = rem - r * NF;
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc = fmaf(dz[(s * R + r) * L + l], fcw[(s * L + l) * (NF + 1) + f], acc);
        a.g_pooled[s][(b * R + r) * NF + f] = acc;
    }
}

static size_t narre_head_lds_bytes(int R, int L) {
    const size_t fl = (size_t)2 * R * NF + 2 * L * (NF + 1) + 2 * L * (2 * L + 1) + L * (L + 1) + 6 * 2 * R * L + 2 * 2 * R +
                      11 * 2 * L + 16;
    return fl * 4;
}

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
template <int ML>
__device__ __forceinline__ void narre_rows_block(const RowSweep &w, int bx) {
    if (bx >= w.cb_entries) {
        // ---- entry waves: 4 per workgroup, all of one table (user table's groups first)
        __shared__ int sid[NROW_MAX_ENTRIES];
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
        unsigned long long mine = 0;                        // bit c: entry c*64 + lane is a hit (nch <= 64)
        for (int c = (int)(k / 64); c < nch; ++c) {         // (no hit before k's chunk: k is the first)
            const int64_t j = (int64_t)c * 64 + lane;
            if (j < w.entries && sid[j] == row) mine |= 1ull << c;
        }
        while (__ballot(mine != 0)) {                       // four of a lane's hits per round, their loads together
            int cs[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                cs[u] = mine ? __ffsll((long long)mine) - 1 : -1;
                if (mine) mine &= mine - 1;
            }
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
            // entry workgroups first (the owner of a popular row is the longest), then the sweep
            for (int blk = blk0; blk < row_blocks; blk += nblk) {
                const int ne = row_blocks - rows.cb_entries;
                narre_rows_block<ML>(rows, blk < ne ? rows.cb_entries + blk : blk - ne);
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

// ---- 5: wgrad partial reduce + Adam on the dense parameters + next batch's compaction
constexpr int NRED_THREADS = 256;
struct DenseAdam {
    float *p, *m, *v;
    const float *g;
    int64_t lo0, hi0, lo1, hi1;
    AdamScalars s;
    int on;
};
__global__ __launch_bounds__(NRED_THREADS) void narre_reduce_kernel(WgradArgs w, int red_blocks, int comp_blocks,
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
        token_compact_block<NRED_THREADS / 64>(nx.t[blockIdx.y], nx.V, bx - red_blocks);
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

// =====================================================================================
// DeepCoNN++ (DeepCoNN.py:37-72 with model_type 'deepconn++'): the two TextCNN towers of DeepCoNN,
// then `final` = Linear(2L, L) -> ReLU -> Dropout -> Linear(L, 1) plus user / item / global bias
// instead of the FM.  Same launch structure as the NARRE step above and the same backward /
// reduce kernels; the ID bias vectors are updated by the D = 0 form of the MF sweep
// (mf_engine.hip).  Flat layout (13 slots): user_conv.convs.0.weight, .bias, user_conv.fc.weight,
// .bias, item_conv.(same four), final.0.weight, final.0.bias, final.3.weight, final.3.bias,
// global_bias.
enum { DP_UCW = 0, DP_UCB, DP_UFW, DP_UFB, DP_ICW, DP_ICB, DP_IFW, DP_IFB, DP_F0W, DP_F0B, DP_F3W, DP_F3B, DP_GB,
       DP_COUNT };
struct DLayout { int64_t off[DP_COUNT], size[DP_COUNT], total; };
static DLayout dcpp_layout(int E, int L) {
    DLayout lay;
    const int64_t sz[DP_COUNT] = {(int64_t)NF * 3 * E, NF, (int64_t)L * NF, L, (int64_t)NF * 3 * E, NF, (int64_t)L * NF, L,
                                  (int64_t)L * 2 * L, L, L, 1, 1};
    int64_t o = 0;
    for (int i = 0; i < DP_COUNT; ++i) {
        lay.off[i] = o;
        lay.size[i] = sz[i];
        o += (sz[i] + 3) & ~(int64_t)3;
    }
    lay.total = o;
    return lay;
}

struct DcppHead {
    const float *pmax[2]; const int *parg[2];       // conv partials [B, tiles, NP]
    const float *flat_p;
    int off[DP_COUNT];
    int col0_lo, col0_n, col1_lo;
    const float *bias[2];                           // user_bias, item_bias
    const int64_t *id[2];                           // uid, iid [B]
    const float *y;
    float *pooled[2]; int *argmax[2]; float *g_pooled[2];   // [B, 100]
    float *part;                                    // [B, NHP]
    float *g;                                       // [B]
    int *tag[2];
    float *mult;                                    // [B, 3L]: user_conv.dropout [L], item_conv.dropout [L], final.2 [L]
    float *pred, *se;
    int64_t B;
    int L, tiles, nhp, training, want_grad, now;
    float p_drop, inv_denom;
    uint64_t seed, offset;
};

// One workgroup of 256 threads per rating (the work is ~2000-element loops: FC gradients, d pooled).
template <int ML>
__global__ __launch_bounds__(256) void dcpp_head_kernel(DcppHead a) {
    __shared__ float P[2][NF];
    __shared__ float fcw[2][ML][NF + 1];
    __shared__ float W1[ML][2 * ML + 1];
    __shared__ float x[2 * ML], xm[2 * ML], dzs[2 * ML], hs[ML], hms[ML], dhs[ML], fcbs[2 * ML], b1s[ML], w3s[ML], misc[8];
    const int L = a.L, L2 = 2 * L, tid = threadIdx.x;
    const int64_t b = blockIdx.x;
    const float *fp = a.flat_p;
    const float keep = 1.f / (1.f - a.p_drop);
    const bool drop = a.training && a.p_drop > 0.f;
    const int ND = 3 * L;
    auto draw = [&](int k) -> float {
        float m = 1.f;
        if (drop) {
            const uint32_t r = philox_first_word(a.offset + (uint64_t)(b * ND + k), a.seed);
            m = ((float)(r >> 8) * (1.0f / 16777216.0f) >= a.p_drop) ? keep : 0.f;
        }
        if (a.mult) a.mult[b * ND + k] = m;
        return m;
    };
    const float invL2 = 1.f / (float)L2;
    auto qd = [](int v, float inv) { return (int)(((float)v + 0.5f) * inv); };
    // ---- S0: weights, pool finish (max over tiles, relu, first argmax), biases
    for (int i = tid; i < 2 * L * NF; i += 256) {
        const int s = i >= L * NF, r = i - s * L * NF, l = r / NF;
        fcw[s][l][r - l * NF] = fp[a.off[s ? DP_IFW : DP_UFW] + r];
    }
    for (int i = tid; i < L * L2; i += 256) { const int k = qd(i, invL2); W1[k][i - k * L2] = fp[a.off[DP_F0W] + i]; }
    if (tid < L2) fcbs[tid] = fp[a.off[tid >= L ? DP_IFB : DP_UFB] + (tid >= L ? tid - L : tid)];
    if (tid < L) { b1s[tid] = fp[a.off[DP_F0B] + tid]; w3s[tid] = fp[a.off[DP_F3W] + tid]; }
    if (tid == 0) {
        misc[0] = fp[a.off[DP_F3B]]; misc[1] = fp[a.off[DP_GB]];
        misc[2] = a.bias[0][a.id[0][b]]; misc[3] = a.bias[1][a.id[1][b]];
    }
    if (tid < 2 * NF) {
        const int s = tid >= NF, f = tid - s * NF;
        float best = -INFINITY;
        int bp = -1;
        for (int k = 0; k < a.tiles; ++k) {
            const size_t q = ((size_t)b * a.tiles + k) * NP + f;
            const float val = a.pmax[s][q];
            if (val > best) { best = val; bp = a.parg[s][q]; }
        }
        if (!(best > 0.f)) { best = 0.f; bp = -1; }
        P[s][f] = best;
        a.pooled[s][b * NF + f] = best;
        a.argmax[s][b * NF + f] = bp;
    }
    __syncthreads();
    // ---- S1: TextCNN FC + dropout (common_pytorch_models.py:35-37)
    if (tid < L2) {
        const int s = tid >= L, l = tid - s * L;
        float acc = 0.f;
        for (int f = 0; f < NF; ++f) acc = fmaf(P[s][f], fcw[s][l][f], acc);
        const float m = draw(tid);
        xm[tid] = m;
        x[tid] = (acc + fcbs[tid]) * m;
    }
    __syncthreads();
    // ---- S2: final.0 + relu + dropout (DeepCoNN.py:21-26)
    if (tid < L) {
        float acc = 0.f;
        for (int j = 0; j < L2; ++j) acc = fmaf(x[j], W1[tid][j], acc);
        acc += b1s[tid];
        hs[tid] = acc > 0.f ? acc : 0.f;
        hms[tid] = draw(L2 + tid);
    }
    __syncthreads();
    // ---- S3: final.3, bias head (DeepCoNN.py:68-72), SE
    if (tid == 0) {
        float acc = 0.f;
        for (int k = 0; k < L; ++k) acc = fmaf(hs[k] * hms[k], w3s[k], acc);
        const float rating = acc + misc[0];
        const float pred = ((rating + misc[2]) + misc[3]) + misc[1];
        a.pred[b] = pred;
        float g = 0.f;
        if (a.y) {
            const float d = pred - a.y[b];
            a.se[b] = d * d;
            g = 2.f * d * a.inv_denom;
        }
        misc[4] = g;
        if (a.want_grad) {
            a.g[b] = g;
            a.tag[0][a.id[0][b]] = a.now;
            a.tag[1][a.id[1][b]] = a.now;
        }
    }
    if (!a.want_grad) return;                               // uniform
    __syncthreads();
    const float g = misc[4];
    float *prow = a.part + (size_t)b * a.nhp;
    auto col = [&](int flat_off) { return flat_off < a.col0_lo + a.col0_n ? flat_off - a.col0_lo : a.col0_n + flat_off - a.col1_lo; };
    // ---- B1: final.3, d hidden
    if (tid < L) {
        prow[col(a.off[DP_F3W] + tid)] = g * hs[tid] * hms[tid];
        const float d = hs[tid] > 0.f ? g * w3s[tid] * hms[tid] : 0.f;
        dhs[tid] = d;
        prow[col(a.off[DP_F0B] + tid)] = d;
    }
    if (tid == 0) { prow[col(a.off[DP_F3B])] = g; prow[col(a.off[DP_GB])] = g; }
    __syncthreads();
    // ---- B2: final.0 weight, d x -> d z
    for (int i = tid; i < L * L2; i += 256) { const int k = qd(i, invL2); prow[col(a.off[DP_F0W] + i)] = dhs[k] * x[i - k * L2]; }
    if (tid < L2) {
        float acc = 0.f;
        for (int k = 0; k < L; ++k) acc = fmaf(dhs[k], W1[k][tid], acc);
        dzs[tid] = acc * xm[tid];
    }
    __syncthreads();
    // ---- B3: TextCNN FC gradients, d pooled
    if (tid < L2) prow[col(a.off[tid >= L ? DP_IFB : DP_UFB] + (tid >= L ? tid - L : tid))] = dzs[tid];
    for (int i = tid; i < 2 * L * NF; i += 256) {
        const int s = i >= L * NF, r = i - s * L * NF, l = r / NF, f = r - l * NF;
        prow[col(a.off[s ? DP_IFW : DP_UFW] + r)] = dzs[s * L + l] * P[s][f];
    }
    if (tid < 2 * NF) {
        const int s = tid >= NF, f = tid - s * NF;
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc = fmaf(dzs[s * L + l], fcw[s][l][f], acc);
        a.g_pooled[s][b * NF + f] = acc;
    }
}

struct DcppWs {
    float *wp[2], *pmax[2]; int *parg[2];
    int *flags[2][2], *slot[2][2], *list[2][2], *count[2][2]; float *ptab[2];
    float *pooled[2]; int *argmax[2]; float *g_pooled[2];
    float *part_w[2], *part_b[2];
    int *tag[2];
    float *part, *g, *mult;
    size_t bytes;
};
static DcppWs dcpp_carve(void *ws, int64_t B, int T, int E, int L, int64_t V, int64_t n_users, int64_t n_items) {
    DcppWs w;
    char *p = static_cast<char *>(ws);
    size_t o = 0;
    auto take = [&](size_t nbytes) { char *r = p ? p + o : nullptr; o += align256(nbytes); return r; };
    const size_t tiles128 = (size_t)(T + 2 + 127) / 128;
    const int ns = textcnn_wgrad_splits(B);
    w.tag[0] = reinterpret_cast<int *>(take((size_t)n_users * 4));          // persistent state first
    w.tag[1] = reinterpret_cast<int *>(take((size_t)n_items * 4));
    for (int t = 0; t < 2; ++t)
        for (int bf = 0; bf < 2; ++bf) {
            w.flags[bf][t] = reinterpret_cast<int *>(take((size_t)(V + 4) * 4));
            w.count[bf][t] = reinterpret_cast<int *>(take(256));
        }
    for (int t = 0; t < 2; ++t) {
        for (int bf = 0; bf < 2; ++bf) {
            w.slot[bf][t] = reinterpret_cast<int *>(take((size_t)(V + 4) * 4));
            w.list[bf][t] = reinterpret_cast<int *>(take((size_t)proj_row_capacity(B, T, V) * 4));
        }
        w.wp[t] = reinterpret_cast<float *>(take(textcnn_wp_floats(E) * 4));
        w.pmax[t] = reinterpret_cast<float *>(take((size_t)B * tiles128 * NP * 4));
        w.parg[t] = reinterpret_cast<int *>(take((size_t)B * tiles128 * NP * 4));
        w.pooled[t] = reinterpret_cast<float *>(take((size_t)B * NF * 4));
        w.argmax[t] = reinterpret_cast<int *>(take((size_t)B * NF * 4));
        w.g_pooled[t] = reinterpret_cast<float *>(take((size_t)B * NF * 4));
        w.part_w[t] = reinterpret_cast<float *>(take((size_t)ns * NF * 3 * E * 4));
        w.part_b[t] = reinterpret_cast<float *>(take((size_t)ns * NF * 4));
        w.ptab[t] = reinterpret_cast<float *>(take(proj_ptab_floats(B, T, V) * 4));
    }
    const DLayout lay = dcpp_layout(E, L);
    const int nhp = (int)((lay.off[DP_ICW] - lay.off[DP_UFW]) + (lay.total - lay.off[DP_IFW]));
    w.part = reinterpret_cast<float *>(take((size_t)B * nhp * 4));
    w.g = reinterpret_cast<float *>(take((size_t)B * 4));
    w.mult = reinterpret_cast<float *>(take((size_t)B * 3 * L * 4));
    w.bytes = o;
    return w;
}


// =====================================================================================
// TransNet / TransNet++ (TransNet.py:9-122, trained by main.py:26-53 with utils.init_transnet_optim,
// utils.py:70-92).  Three TextCNN towers -- source.user_conv, source.item_conv on the user's and the
// item's documents, target.conv on the review being rated -- and three losses from ONE forward:
//   target     mean (FM_t(t_ir) - y)^2               -> optimizer_target   (target.conv, target.fm)
//   transform  mean ||s_ir - t_ir||^2                -> optimizer_source   (source.*)
//   source     mean (FM_s([ue, ie,] s_ir) - y)^2     -> optimizer_source_fm (source_fm, ID vectors)
// with s_ir = dropout(project(cat(xu, xi))), t_ir = dropout(xt).  The reference runs three backward
// passes over the retained graph with an optimiser step after each, gradients accumulating; the
// gradient each optimiser CONSUMES is that of its own loss with respect to its own (disjoint)
// parameter group, evaluated at the pre-step weights of everything it flows through (the target
// pass reaches nothing else; the transform pass reaches source.* before optimizer_source has
// stepped, its contribution to target.* is zeroed unused at the next batch; the source pass
// stops at source_fm and the ID vectors as far as consumed gradients go).  So the step is one
// backward with three disjoint groups, and -- the three Adams sharing lr, weight decay and step
// count -- one flat Adam.  Pinned by the reference-generated 3-step trajectories (tests).
// Flat layout (22 slots): the three conv weight / bias pairs first, then every head parameter in
// one contiguous range.
enum { TN_UCW = 0, TN_UCB, TN_ICW, TN_ICB, TN_TCW, TN_TCB, TN_UFW, TN_UFB, TN_IFW, TN_IFB, TN_TFW, TN_TFB,
       TN_P0W, TN_P0B, TN_P2W, TN_P2B, TN_SV, TN_SLW, TN_SLB, TN_TV, TN_TLW, TN_TLB, TN_COUNT };
constexpr int TN_ID = 5;               // width of the ID vectors (TransNet.py:75-76)
constexpr int TN_FM_K = 8;             // factors of both FMs (TransNet.py:50,77,79)
struct TLayout { int64_t off[TN_COUNT], size[TN_COUNT], total; };
static TLayout tn_layout(int E, int L, int plus) {
    TLayout lay;
    const int64_t ns = L + (plus ? 2 * TN_ID : 0);
    const int64_t sz[TN_COUNT] = {(int64_t)NF * 3 * E, NF, (int64_t)NF * 3 * E, NF, (int64_t)NF * 3 * E, NF,
                                  (int64_t)L * NF, L, (int64_t)L * NF, L, (int64_t)L * NF, L,
                                  (int64_t)L * 2 * L, L, (int64_t)L * L, L,
                                  ns * TN_FM_K, ns, 1, (int64_t)L * TN_FM_K, L, 1};
    int64_t o = 0;
    for (int i = 0; i < TN_COUNT; ++i) {
        lay.off[i] = o;
        lay.size[i] = sz[i];
        o += (sz[i] + 3) & ~(int64_t)3;
    }
    lay.total = o;
    return lay;
}

struct TnHead {
    const float *pmax[3]; const int *parg[3];       // conv partials [B, tiles, NP]: user, item, this
    const float *flat_p;
    int off[TN_COUNT];
    int lo;                                         // first head parameter (column 0 of `part`)
    const float *emb[2];                            // user / item ID vectors [*, 5] (TransNet++) or NULL
    const int64_t *id[2];                           // uid, iid [B]
    const float *y;
    float *pooled[3]; int *argmax[3]; float *g_pooled[3];   // [B, 100]
    float *part;                                    // [B, NHP]
    float *grow[2];                                 // [B, 5] compact gradient rows of the ID vectors
    int *tag[2], *ctag[2];                          // row tags; sweep-chunk tags (rows_device.h)
    float *mult;                                    // [B, 5L + 10] dropout multipliers (see r4r.h)
    float *pred, *se;                               // source prediction and its SE
    float *aux;                                     // [B, 3]: target prediction, its SE, ||s_ir - t_ir||^2
    int64_t B;
    int L, tiles, nhp, training, want_grad, now, plus;
    float p_drop, inv_denom;
    uint64_t seed, offset;
};

// One workgroup of 256 threads per rating.
template <int ML>
__global__ __launch_bounds__(256) void tn_head_kernel(TnHead a) {
    constexpr int MS = ML + 2 * TN_ID;                      // widest source_fm input
    __shared__ float P[3][NF];
    __shared__ float fcw[3][ML][NF + 1];
    __shared__ float W0[ML][2 * ML + 1], W2[ML][ML + 1];
    __shared__ float Vs[MS][TN_FM_K], Vt[ML][TN_FM_K], lws[MS], lwt[ML];
    __shared__ float x[3 * ML], xm[3 * ML], fcb[3 * ML], dz[3 * ML];
    __shared__ float b0s[ML], b2s[ML], hs[ML], dhs[ML], sir[ML], sirm[ML], tir[ML], tirm[ML], dtmp[ML];
    __shared__ float fin[MS], finm[2 * TN_ID], dfs[MS], dft[ML], sks[TN_FM_K], skt[TN_FM_K], misc[8];
    const int L = a.L, L2 = 2 * L, L3 = 3 * L, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ns = L + (a.plus ? 2 * TN_ID : 0), eo = a.plus ? 2 * TN_ID : 0;   // final = [ue, ie, s_ir]
    const int64_t b = blockIdx.x;
    const float *fp = a.flat_p;
    const float keep = 1.f / (1.f - a.p_drop);
    const bool drop = a.training && a.p_drop > 0.f;
    const int ND = 5 * L + 2 * TN_ID;
    auto draw = [&](int k) -> float {
        float m = 1.f;
        if (drop) {
            const uint32_t r = philox_first_word(a.offset + (uint64_t)(b * ND + k), a.seed);
            m = ((float)(r >> 8) * (1.0f / 16777216.0f) >= a.p_drop) ? keep : 0.f;
        }
        if (a.mult) a.mult[b * ND + k] = m;
        return m;
    };
    const float invL2 = 1.f / (float)L2, invL = 1.f / (float)L;
    auto qd = [](int v, float inv) { return (int)(((float)v + 0.5f) * inv); };
    // ---- S0: weights, pool finish (max over tiles, relu, first argmax), ID vectors
    for (int i = tid; i < 3 * L * NF; i += 256) {
        const int s = i / (L * NF), r = i - s * L * NF, l = r / NF;
        fcw[s][l][r - l * NF] = fp[a.off[s == 0 ? TN_UFW : (s == 1 ? TN_IFW : TN_TFW)] + r];
    }
    for (int i = tid; i < L * L2; i += 256) { const int k = qd(i, invL2); W0[k][i - k * L2] = fp[a.off[TN_P0W] + i]; }
    for (int i = tid; i < L * L; i += 256) { const int k = qd(i, invL); W2[k][i - k * L] = fp[a.off[TN_P2W] + i]; }
    for (int i = tid; i < ns * TN_FM_K; i += 256) Vs[i / TN_FM_K][i % TN_FM_K] = fp[a.off[TN_SV] + i];
    for (int i = tid; i < L * TN_FM_K; i += 256) Vt[i / TN_FM_K][i % TN_FM_K] = fp[a.off[TN_TV] + i];
    if (tid < ns) lws[tid] = fp[a.off[TN_SLW] + tid];
    if (tid < L) { lwt[tid] = fp[a.off[TN_TLW] + tid]; b0s[tid] = fp[a.off[TN_P0B] + tid]; b2s[tid] = fp[a.off[TN_P2B] + tid]; }
    if (tid < L3) {
        const int s = tid / L;
        fcb[tid] = fp[a.off[s == 0 ? TN_UFB : (s == 1 ? TN_IFB : TN_TFB)] + (tid - s * L)];
    }
    if (tid == 0) { misc[0] = fp[a.off[TN_SLB]]; misc[1] = fp[a.off[TN_TLB]]; }
    if (a.plus && tid >= 64 && tid < 64 + 2 * TN_ID) {      // dropout.user / dropout.item on the ID vectors
        const int k = tid - 64, s = k >= TN_ID, c = k - s * TN_ID;
        const float m = draw(5 * L + k);
        finm[k] = m;
        fin[k] = a.emb[s][a.id[s][b] * TN_ID + c] * m;
    }
    for (int i = tid; i < 3 * NF; i += 256) {
        const int s = i / NF, f = i - s * NF;
        float best = -INFINITY;
        int bp = -1;
        for (int k = 0; k < a.tiles; ++k) {
            const size_t q = ((size_t)b * a.tiles + k) * NP + f;
            const float val = a.pmax[s][q];
            if (val > best) { best = val; bp = a.parg[s][q]; }
        }
        if (!(best > 0.f)) { best = 0.f; bp = -1; }
        P[s][f] = best;
        a.pooled[s][b * NF + f] = best;
        a.argmax[s][b * NF + f] = bp;
    }
    __syncthreads();
    // ---- S1: the towers' FC + dropout (common_pytorch_models.py:35-37): xu, xi, xt
    if (tid < L3) {
        const int s = tid / L, l = tid - s * L;
        float acc = 0.f;
        for (int f = 0; f < NF; ++f) acc = fmaf(P[s][f], fcw[s][l][f], acc);
        const float m = draw(tid);
        xm[tid] = m;
        x[tid] = (acc + fcb[tid]) * m;
    }
    __syncthreads();
    // ---- S2: source.project.0 + relu (TransNet.py:19-22); target: t_ir = dropout(xt) (TransNet.py:58)
    if (tid < L) {
        float acc = 0.f;
        for (int j = 0; j < L2; ++j) acc = fmaf(x[j], W0[tid][j], acc);
        acc += b0s[tid];
        hs[tid] = acc > 0.f ? acc : 0.f;
    } else if (tid >= 64 && tid < 64 + L) {
        const int l = tid - 64;
        const float m = draw(4 * L + l);
        tirm[l] = m;
        tir[l] = x[L2 + l] * m;
    }
    __syncthreads();
    // ---- S3: source.project.2 + dropout = s_ir (TransNet.py:33-36)
    if (tid < L) {
        float acc = 0.f;
        for (int j = 0; j < L; ++j) acc = fmaf(hs[j], W2[tid][j], acc);
        const float m = draw(3 * L + tid);
        sirm[tid] = m;
        const float v = (acc + b2s[tid]) * m;
        sir[tid] = v;
        fin[eo + tid] = v;
    }
    __syncthreads();
    // ---- S4: the two factorisation machines (common_pytorch_models.py:49-57): wave 0 source, wave 1 target
    if (wv < 2) {
        const int n = wv ? L : ns;
        const float xi = lane < n ? (wv ? tir[lane] : fin[lane]) : 0.f;
        float inter = 0.f, gacc = 0.f;
#pragma unroll
        for (int k = 0; k < TN_FM_K; ++k) {
            const float v = lane < n ? (wv ? Vt[lane][k] : Vs[lane][k]) : 0.f;
            const float s = wave_sum(xi * v);
            const float s2 = wave_sum(xi * xi * v * v);
            inter += s * s - s2;
            gacc += s * v - xi * v * v;
            if (lane == 0) (wv ? skt : sks)[k] = s;
        }
        const float lw = lane < n ? (wv ? lwt[lane] : lws[lane]) : 0.f;
        const float lin = wave_sum(xi * lw);
        const float out = 0.5f * inter + (lin + misc[wv]);
        if (lane < n) (wv ? dft : dfs)[lane] = gacc + lw;      // d FM / d x_i
        if (lane == 0) misc[2 + wv] = out;
    } else if (wv == 2) {
        const float d = lane < L ? sir[lane] - tir[lane] : 0.f;
        const float tr = wave_sum(d * d);
        if (lane == 0) misc[4] = tr;
    }
    __syncthreads();
    const float out_s = misc[2], out_t = misc[3];
    float g_s = 0.f, g_t = 0.f;
    if (a.y) {
        const float yb = a.y[b];
        g_s = 2.f * (out_s - yb) * a.inv_denom;
        g_t = 2.f * (out_t - yb) * a.inv_denom;
        if (tid == 0) {
            a.se[b] = (out_s - yb) * (out_s - yb);
            if (a.aux) { a.aux[b * 3] = out_t; a.aux[b * 3 + 1] = (out_t - yb) * (out_t - yb); a.aux[b * 3 + 2] = misc[4]; }
        }
    } else if (tid == 0 && a.aux) {
        a.aux[b * 3] = out_t; a.aux[b * 3 + 1] = 0.f; a.aux[b * 3 + 2] = misc[4];
    }
    if (tid == 0) {
        a.pred[b] = out_s;
        if (a.want_grad && a.plus) {
            for (int s = 0; s < 2; ++s) {
                const int64_t r = a.id[s][b];
                a.tag[s][r] = a.now;
                a.ctag[s][r * TN_ID / MF_CHUNK] = a.now;              // (a row can straddle two chunks)
                a.ctag[s][(r * TN_ID + TN_ID - 1) / MF_CHUNK] = a.now;
            }
        }
    }
    if (!a.want_grad) return;                               // uniform
    float *prow = a.part + (size_t)b * a.nhp;
    auto col = [&](int flat_off) { return flat_off - a.lo; };
    // ---- B1: FM parameter gradients (d V_ik = g (x_i s_k - x_i^2 V_ik), d lin.w_i = g x_i, d lin.b = g);
    // the ID vectors' compact gradient rows; d s_ir from the transform loss; d t_ir from the target loss
    for (int i = tid; i < ns * TN_FM_K; i += 256) {
        const int r = i / TN_FM_K, k = i - r * TN_FM_K;
        prow[col(a.off[TN_SV] + i)] = g_s * (fin[r] * sks[k] - fin[r] * fin[r] * Vs[r][k]);
    }
    for (int i = tid; i < L * TN_FM_K; i += 256) {
        const int r = i / TN_FM_K, k = i - r * TN_FM_K;
        prow[col(a.off[TN_TV] + i)] = g_t * (tir[r] * skt[k] - tir[r] * tir[r] * Vt[r][k]);
    }
    if (tid < ns) prow[col(a.off[TN_SLW] + tid)] = g_s * fin[tid];
    if (tid < L) prow[col(a.off[TN_TLW] + tid)] = g_t * tir[tid];
    if (tid == 0) { prow[col(a.off[TN_SLB])] = g_s; prow[col(a.off[TN_TLB])] = g_t; }
    if (a.plus && tid >= 64 && tid < 64 + 2 * TN_ID) {
        const int k = tid - 64, s = k >= TN_ID, c = k - s * TN_ID;
        a.grow[s][b * TN_ID + c] = g_s * dfs[k] * finm[k];
    }
    if (tid >= 128 && tid < 128 + L) {
        const int l = tid - 128;
        dtmp[l] = 2.f * (sir[l] - tir[l]) * a.inv_denom * sirm[l];       // d (project.2 output)
        dz[L2 + l] = g_t * dft[l] * tirm[l] * xm[L2 + l];                // d (target FC output)
    }
    __syncthreads();
    // ---- B2: source.project.2 gradients, d hidden
    for (int i = tid; i < L * L; i += 256) { const int k = qd(i, invL); prow[col(a.off[TN_P2W] + i)] = dtmp[k] * hs[i - k * L]; }
    if (tid < L) {
        prow[col(a.off[TN_P2B] + tid)] = dtmp[tid];
        float acc = 0.f;
        for (int k = 0; k < L; ++k) acc = fmaf(dtmp[k], W2[k][tid], acc);
        const float d = hs[tid] > 0.f ? acc : 0.f;
        dhs[tid] = d;
        prow[col(a.off[TN_P0B] + tid)] = d;
    }
    __syncthreads();
    // ---- B3: source.project.0 weight, d cat -> d (source FC outputs)
    for (int i = tid; i < L * L2; i += 256) { const int k = qd(i, invL2); prow[col(a.off[TN_P0W] + i)] = dhs[k] * x[i - k * L2]; }
    if (tid < L2) {
        float acc = 0.f;
        for (int k = 0; k < L; ++k) acc = fmaf(dhs[k], W0[k][tid], acc);
        dz[tid] = acc * xm[tid];
    }
    __syncthreads();
    // ---- B4: the towers' FC gradients, d pooled
    if (tid < L3) {
        const int s = tid / L;
        prow[col(a.off[s == 0 ? TN_UFB : (s == 1 ? TN_IFB : TN_TFB)] + (tid - s * L))] = dz[tid];
    }
    for (int i = tid; i < 3 * L * NF; i += 256) {
        const int s = i / (L * NF), r = i - s * L * NF, l = r / NF, f = r - l * NF;
        prow[col(a.off[s == 0 ? TN_UFW : (s == 1 ? TN_IFW : TN_TFW)] + r)] = dz[s * L + l] * P[s][f];
    }
    for (int i = tid; i < 3 * NF; i += 256) {
        const int s = i / NF, f = i - s * NF;
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc = fmaf(dz[s * L + l], fcw[s][l][f], acc);
        a.g_pooled[s][b * NF + f] = acc;
    }
}

struct TnWs {
    float *wp[3], *pmax[3]; int *parg[3];
    int *flags[2][3], *slot[2][3], *list[2][3], *count[2][3]; float *ptab[3];
    float *pooled[3]; int *argmax[3]; float *g_pooled[3];
    float *part_w[3], *part_b[3];
    int *tag[2], *ctag[2];
    float *part, *grow[2], *mult, *aux;
    size_t bytes, persist;
};
static TnWs tn_carve(void *ws, int64_t B, int T, int E, int L, int plus, int64_t V, int64_t n_users, int64_t n_items) {
    TnWs w;
    char *p = static_cast<char *>(ws);
    size_t o = 0;
    auto take = [&](size_t nbytes) { char *r = p ? p + o : nullptr; o += align256(nbytes); return r; };
    const size_t tiles128 = (size_t)(T + 2 + 127) / 128;
    const int ns = textcnn_wgrad_splits(B);
    w.tag[0] = reinterpret_cast<int *>(take((size_t)n_users * 4));          // persistent state first
    w.tag[1] = reinterpret_cast<int *>(take((size_t)n_items * 4));
    w.ctag[0] = reinterpret_cast<int *>(take((size_t)cdiv(n_users * TN_ID, MF_CHUNK) * 4));
    w.ctag[1] = reinterpret_cast<int *>(take((size_t)cdiv(n_items * TN_ID, MF_CHUNK) * 4));
    w.persist = o;
    for (int t = 0; t < 3; ++t)
        for (int bf = 0; bf < 2; ++bf) {
            w.flags[bf][t] = reinterpret_cast<int *>(take((size_t)(V + 4) * 4));
            w.count[bf][t] = reinterpret_cast<int *>(take(256));
        }
    for (int t = 0; t < 3; ++t) {
        for (int bf = 0; bf < 2; ++bf) {
            w.slot[bf][t] = reinterpret_cast<int *>(take((size_t)(V + 4) * 4));
            w.list[bf][t] = reinterpret_cast<int *>(take((size_t)proj_row_capacity(B, T, V) * 4));
        }
        w.wp[t] = reinterpret_cast<float *>(take(textcnn_wp_floats(E) * 4));
        w.pmax[t] = reinterpret_cast<float *>(take((size_t)B * tiles128 * NP * 4));
        w.parg[t] = reinterpret_cast<int *>(take((size_t)B * tiles128 * NP * 4));
        w.pooled[t] = reinterpret_cast<float *>(take((size_t)B * NF * 4));
        w.argmax[t] = reinterpret_cast<int *>(take((size_t)B * NF * 4));
        w.g_pooled[t] = reinterpret_cast<float *>(take((size_t)B * NF * 4));
        w.part_w[t] = reinterpret_cast<float *>(take((size_t)ns * NF * 3 * E * 4));
        w.part_b[t] = reinterpret_cast<float *>(take((size_t)ns * NF * 4));
        w.ptab[t] = reinterpret_cast<float *>(take(proj_ptab_floats(B, T, V) * 4));
    }
    const TLayout lay = tn_layout(E, L, plus);
    w.part = reinterpret_cast<float *>(take((size_t)B * (lay.total - lay.off[TN_UFW]) * 4));
    w.grow[0] = reinterpret_cast<float *>(take((size_t)B * TN_ID * 4));
    w.grow[1] = reinterpret_cast<float *>(take((size_t)B * TN_ID * 4));
    w.mult = reinterpret_cast<float *>(take((size_t)B * (5 * L + 2 * TN_ID) * 4));
    w.aux = reinterpret_cast<float *>(take((size_t)B * 3 * 4));
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
    R4R_REQUIRE(R > 0 && R <= NR_MAX_R, "narre_step: narre_num_reviews %d outside 1..%d", R, NR_MAX_R);
    R4R_REQUIRE(L > 0 && L <= NR_MAX_L, "narre_step: latent_size %d outside 1..%d", L, NR_MAX_L);
    R4R_REQUIRE(E > 0 && E % 4 == 0, "narre_step: word_embed_size %d must be a positive multiple of 4", E);
    const bool train_step = flat_g != nullptr;
    R4R_REQUIRE(!train_step || (y && se && flat_m && flat_v && rows_m && rows_v && adam_step >= 1),
                "narre_step: a training step needs ratings, se, gradient / moment buffers and adam_step >= 1");
    R4R_REQUIRE(!y || se, "narre_step: se buffer required when y is given");
    R4R_REQUIRE(!next_user_reviews == !next_item_reviews, "narre_step: next_user_reviews and next_item_reviews go together");
    R4R_REQUIRE(!next_user_reviews || train_step, "narre_step: the next batch's tokens ride on the backward launches");
    R4R_REQUIRE(token_buffer == 0 || token_buffer == 1, "narre_step: token_buffer must be 0 or 1");
    R4R_REQUIRE(adam_step < (1ll << 31), "narre_step: step tag overflow");
    R4R_REQUIRE(!train_step || B * (1 + R) <= NROW_MAX_ENTRIES, "narre_step: %lld ID entries per table > %d (use the "
                "module path for larger batches)", (long long)(B * (1 + R)), NROW_MAX_ENTRIES);
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
            pt[t].ptab = w.ptab[t]; pt[t].pmax = w.pmax[t]; pt[t].parg = w.parg[t];
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
    const bool small = R <= 16 && L <= 16;
    static size_t lds_set[2] = {0, 0};
    if (lds > lds_set[small]) {
        (void)hipFuncSetAttribute(small ? reinterpret_cast<const void *>(narre_head_kernel<16, 16, NHEAD_THREADS>)
                                        : reinterpret_cast<const void *>(narre_head_kernel<32, 32, NHEAD_THREADS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lds_set[small] = lds;
    }
    if (small) narre_head_kernel<16, 16, NHEAD_THREADS><<<(unsigned)B, NHEAD_THREADS, lds, st>>>(h);
    else narre_head_kernel<32, 32, NHEAD_THREADS><<<(unsigned)B, NHEAD_THREADS, lds, st>>>(h);
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
        rp[k] = reinterpret_cast<float *>(rows_p[k]); rm[k] = reinterpret_cast<float *>(rows_m[k]);
        rv[k] = reinterpret_cast<float *>(rows_v[k]);
        R4R_REQUIRE(rp[k] && rm[k] && rv[k], "narre_step: row tensor %d: null parameter / moment pointer", k);
    }
    rs.p0 = rp[0]; rs.p1 = rp[1]; rs.p2 = rp[2]; rs.p3 = rp[3];
    rs.m0 = rm[0]; rs.m1 = rm[1]; rs.m2 = rm[2]; rs.m3 = rm[3];
    rs.v0 = rv[0]; rs.v1 = rv[1]; rs.v2 = rv[2]; rs.v3 = rv[3];
    const int64_t numel[4] = {n_users * L, n_items * L, n_users, n_items};
    int64_t begin[5], chunks = 0;
    for (int k = 0; k < 4; ++k) { begin[k] = chunks; chunks += cdiv(numel[k], NROW_CHUNK); }
    rs.n0 = numel[0]; rs.n1 = numel[1]; rs.n2 = numel[2]; rs.n3 = numel[3];
    rs.cb1 = (int)begin[1]; rs.cb2 = (int)begin[2]; rs.cb3 = (int)begin[3]; rs.cb_entries = (int)chunks;
    chunks += 2 * cdiv(B * (1 + R), 4);                     // the entry waves, 4 per workgroup, per table
    R4R_REQUIRE(chunks < (1ll << 31), "narre_step: too many chunks");
    rs.gid0 = w.gid[0]; rs.gid1 = w.gid[1]; rs.grow0 = w.grow[0]; rs.grow1 = w.grow[1]; rs.g = w.g;
    rs.tag0 = w.tag[0]; rs.tag1 = w.tag[1]; rs.entries = B * (1 + R); rs.B = B; rs.L = L; rs.now = (int)adam_step;
    rs.s = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);

    const int packed = 3 * E / 4 <= 64;                     // narrow windows: one wave per filter
    const dim3 bgrid(packed ? (NF + 3) / 4 : NF, wa.nsplit, prefetch ? 5 : 4);
    if (L <= 16) narre_backward_kernel<16><<<bgrid, WG_THREADS, 0, st>>>(wa, cs, cs_blocks, nx, packed, rs, (int)chunks, 2);
    else narre_backward_kernel<32><<<bgrid, WG_THREADS, 0, st>>>(wa, cs, cs_blocks, nx, packed, rs, (int)chunks, 2);

    // 5: wgrad reduce + Adam on the dense parameters (+ next batch's compaction)
    const int red_blocks = (NF * 3 * E + NF + NRED_THREADS - 1) / NRED_THREADS;
    const int comp_blocks = prefetch ? (int)cdiv((V + 3) / 4, NRED_THREADS) : 0;
    DenseAdam opt;
    opt.on = 1; opt.p = flat_p; opt.m = flat_m; opt.v = flat_v; opt.g = flat_g;
    opt.lo0 = hc.lo0; opt.hi0 = hc.hi0; opt.lo1 = hc.lo1; opt.hi1 = hc.hi1;
    opt.s = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    const int64_t longest = hc.hi0 - hc.lo0 > hc.hi1 - hc.lo1 ? hc.hi0 - hc.lo0 : hc.hi1 - hc.lo1;
    const int opt_blocks = (int)cdiv(longest, NRED_THREADS);
    narre_reduce_kernel<<<dim3(red_blocks + comp_blocks + opt_blocks, 2), NRED_THREADS, 0, st>>>(wa, red_blocks, comp_blocks,
                                                                                                nx, opt);

    return check_launch("narre_step");
}

// ------------------------------------------------------------------------------ DeepCoNN++
extern "C" int r4r_deepconnpp_nparam(void) { return DP_COUNT; }

extern "C" int r4r_deepconnpp_layout(int E, int L, int64_t *offsets, int64_t *sizes, int64_t *total) {
    R4R_REQUIRE(offsets && sizes && total, "deepconnpp_layout: null pointer");
    R4R_REQUIRE(E > 0 && L > 0 && L <= NR_MAX_L, "deepconnpp_layout: bad sizes");
    const DLayout lay = dcpp_layout(E, L);
    for (int i = 0; i < DP_COUNT; ++i) { offsets[i] = lay.off[i]; sizes[i] = lay.size[i]; }
    *total = lay.total;
    return R4R_OK;
}

extern "C" size_t r4r_deepconnpp_ws_bytes(int64_t B, int T, int E, int L, int64_t V, int64_t n_users, int64_t n_items) {
    if (B < 0 || T <= 0 || E <= 0 || L <= 0 || V <= 0 || n_users <= 0 || n_items <= 0) return 0;
    return dcpp_carve(nullptr, B, T, E, L, V, n_users, n_items).bytes;
}

// which: 0 dropout multipliers [B, 3L]; 5 d loss / d pred [B]; 6 + 2 * tower + buffer: a token buffer's counter
extern "C" size_t r4r_deepconnpp_ws_offset(int64_t B, int T, int E, int L, int64_t V, int64_t n_users, int64_t n_items,
                                           int which) {
    const DcppWs w = dcpp_carve(reinterpret_cast<void *>(256), B, T, E, L, V, n_users, n_items);
    if (which >= 6 && which < 10)
        return (size_t)(reinterpret_cast<char *>(w.count[(which - 6) & 1][(which - 6) >> 1]) - reinterpret_cast<char *>(256));
    const char *q = which == 0 ? reinterpret_cast<char *>(w.mult) : reinterpret_cast<char *>(w.g);
    return (size_t)(q - reinterpret_cast<char *>(256));
}

extern "C" int r4r_deepconnpp_step(const float *table, int64_t V, const int64_t *user_idx, const int64_t *item_idx,
                                   const int64_t *uid, const int64_t *iid, const float *y,
                                   float *flat_p, float *flat_g, float *flat_m, float *flat_v,
                                   const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                                   int64_t n_users, int64_t n_items,
                                   float *pred, float *se, float *sse_accum, void *ws, size_t ws_bytes,
                                   int64_t B, int T, int E, int L,
                                   float dropout_p, int training, uint64_t seed, uint64_t offset, float inv_denom,
                                   int conv_algo, int token_buffer, int tokens_ready,
                                   const int64_t *next_user_idx, const int64_t *next_item_idx,
                                   float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                                   void *stream) {
    R4R_REQUIRE(table && user_idx && item_idx && uid && iid && flat_p && rows_p && pred && ws, "deepconnpp_step: null pointer");
    R4R_REQUIRE(V > 0 && B >= 0 && T > 0 && n_users > 0 && n_items > 0, "deepconnpp_step: bad sizes");
    R4R_REQUIRE(L > 0 && L <= NR_MAX_L, "deepconnpp_step: latent_size %d outside 1..%d", L, NR_MAX_L);
    R4R_REQUIRE(E > 0 && E % 4 == 0, "deepconnpp_step: word_embed_size %d must be a positive multiple of 4", E);
    const bool train_step = flat_g != nullptr;
    R4R_REQUIRE(!train_step || (y && se && flat_m && flat_v && rows_m && rows_v && adam_step >= 1),
                "deepconnpp_step: a training step needs ratings, se, gradient / moment buffers and adam_step >= 1");
    R4R_REQUIRE(!y || se, "deepconnpp_step: se buffer required when y is given");
    R4R_REQUIRE(!next_user_idx == !next_item_idx, "deepconnpp_step: next_user_idx and next_item_idx go together");
    R4R_REQUIRE(!next_user_idx || train_step, "deepconnpp_step: the next batch's tokens ride on the backward launches");
    R4R_REQUIRE(token_buffer == 0 || token_buffer == 1, "deepconnpp_step: token_buffer must be 0 or 1");
    R4R_REQUIRE(adam_step < (1ll << 31), "deepconnpp_step: step tag overflow");
    R4R_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "deepconnpp_step: dropout %f outside [0,1)", (double)dropout_p);
    R4R_REQUIRE(B * (int64_t)((T + 2 + 127) / 128) < (1ll << 31), "deepconnpp_step: grid too large");
    if (ws_bytes < r4r_deepconnpp_ws_bytes(B, T, E, L, V, n_users, n_items)) {
        set_error("deepconnpp_step: workspace %zu < %zu bytes", ws_bytes, r4r_deepconnpp_ws_bytes(B, T, E, L, V, n_users, n_items));
        return R4R_ERR_WORKSPACE;
    }
    if (B == 0) return R4R_OK;
    hipStream_t st = as_stream(stream);
    const DLayout lay = dcpp_layout(E, L);
    R4R_REQUIRE(lay.total < (1ll << 31), "deepconnpp_step: dense parameter buffer too large");
    const int64_t lo0 = lay.off[DP_UFW], hi0 = lay.off[DP_ICW], lo1 = lay.off[DP_IFW], hi1 = lay.total;
    const int nhp = (int)((hi0 - lo0) + (hi1 - lo1));
    const DcppWs w = dcpp_carve(ws, B, T, E, L, V, n_users, n_items);
    const float *P[DP_COUNT];
    float *G[DP_COUNT];
    for (int i = 0; i < DP_COUNT; ++i) { P[i] = flat_p + lay.off[i]; G[i] = flat_g ? flat_g + lay.off[i] : nullptr; }

    const int64_t *idx[2] = {user_idx, item_idx};
    const int algo = textcnn_pick_algo(conv_algo, B, T, E, NF);
    int tiles;
    if (algo == R4R_CONV_PROJECT) {
        ProjTower pt[2];
        for (int t = 0; t < 2; ++t) {
            pt[t].idx = idx[t];
            pt[t].conv_w = P[t ? DP_ICW : DP_UCW]; pt[t].conv_b = P[t ? DP_ICB : DP_UCB];
            pt[t].flags = w.flags[token_buffer][t]; pt[t].slot = w.slot[token_buffer][t];
            pt[t].list = w.list[token_buffer][t]; pt[t].count = w.count[token_buffer][t];
            pt[t].ptab = w.ptab[t]; pt[t].pmax = w.pmax[t]; pt[t].parg = w.parg[t];
        }
        if (!tokens_ready)
            if (int rc = textcnn_proj_tokens_launch(V, pt, 2, B, T, /*zero_state=*/false, st)) return rc;
        if (int rc = textcnn_proj_compute_launch(table, V, pt, 2, B, T, E, NF, st)) return rc;
        tiles = proj_tiles(T);
    } else {
        FwdTower ft[2];
        for (int t = 0; t < 2; ++t) {
            ft[t].idx = idx[t];
            ft[t].conv_w = P[t ? DP_ICW : DP_UCW]; ft[t].conv_b = P[t ? DP_ICB : DP_UCB];
            ft[t].wp = w.wp[t]; ft[t].pmax = w.pmax[t]; ft[t].parg = w.parg[t];
        }
        if (int rc = textcnn_fwd_launch(table, ft, 2, B, T, E, NF, st)) return rc;
        tiles = textcnn_tiles(T);
    }

    DcppHead h;
    for (int t = 0; t < 2; ++t) {
        h.pmax[t] = w.pmax[t]; h.parg[t] = w.parg[t];
        h.pooled[t] = w.pooled[t]; h.argmax[t] = w.argmax[t]; h.g_pooled[t] = w.g_pooled[t];
        h.bias[t] = reinterpret_cast<const float *>(rows_p[t]);
        h.tag[t] = w.tag[t];
        R4R_REQUIRE(h.bias[t], "deepconnpp_step: null bias pointer");
    }
    h.id[0] = uid; h.id[1] = iid; h.flat_p = flat_p;
    for (int i = 0; i < DP_COUNT; ++i) h.off[i] = (int)lay.off[i];
    h.col0_lo = (int)lo0; h.col0_n = (int)(hi0 - lo0); h.col1_lo = (int)lo1;
    h.y = y; h.part = w.part; h.g = w.g; h.mult = w.mult; h.pred = pred; h.se = se;
    h.B = B; h.L = L; h.tiles = tiles; h.nhp = nhp; h.training = training; h.want_grad = train_step;
    h.now = (int)adam_step; h.p_drop = dropout_p; h.inv_denom = inv_denom; h.seed = seed; h.offset = offset;
    if (L <= 16) dcpp_head_kernel<16><<<(unsigned)B, 256, 0, st>>>(h);
    else dcpp_head_kernel<32><<<(unsigned)B, 256, 0, st>>>(h);
    if (!train_step) return check_launch("deepconnpp_step(forward)");

    WgradTower wt[2];
    WgradArgs wa;
    for (int t = 0; t < 2; ++t) {
        wt[t].idx = idx[t]; wt[t].g_pooled = w.g_pooled[t]; wt[t].argmax = w.argmax[t];
        wt[t].part_w = w.part_w[t]; wt[t].part_b = w.part_b[t];
        wt[t].d_w = G[t ? DP_ICW : DP_UCW]; wt[t].d_b = G[t ? DP_ICB : DP_UCB];
    }
    for (int k = 0; k < MAX_TOWERS; ++k) wa.t[k] = wt[k < 2 ? k : 0];
    wa.table = table; wa.N = B; wa.T = T; wa.E = E; wa.F = NF;
    wa.nsplit = textcnn_wgrad_splits(B);
    wa.per_split = (int)cdiv(B, wa.nsplit);
    ColSum cs;
    cs.part = w.part; cs.se = se; cs.flat_g = flat_g; cs.sse_accum = sse_accum; cs.B = B; cs.nhp = nhp;
    cs.col0_lo = (int)lo0; cs.col0_n = (int)(hi0 - lo0); cs.col1_lo = (int)lo1;
    const int cs_blocks = (nhp + 1 + CS_COLS - 1) / CS_COLS;
    const bool prefetch = next_user_idx && algo == R4R_CONV_PROJECT;
    TokenArgs nx{};
    if (prefetch) {
        ProjTower nt[2];
        const int64_t *nidx[2] = {next_user_idx, next_item_idx};
        const int ob = token_buffer ^ 1;
        for (int t = 0; t < 2; ++t) {
            nt[t] = ProjTower{};
            nt[t].idx = nidx[t];
            nt[t].flags = w.flags[ob][t]; nt[t].slot = w.slot[ob][t]; nt[t].list = w.list[ob][t]; nt[t].count = w.count[ob][t];
        }
        nx = make_token_args(V, nt, 2, B, T);
    }
    const int packed = 3 * E / 4 <= 64;
    narre_backward_kernel<0><<<dim3(packed ? (NF + 3) / 4 : NF, wa.nsplit, prefetch ? 4 : 3), WG_THREADS, 0, st>>>(
        wa, cs, cs_blocks, nx, packed, RowSweep{}, 0, 2);

    const int red_blocks = (NF * 3 * E + NF + NRED_THREADS - 1) / NRED_THREADS;
    const int comp_blocks = prefetch ? (int)cdiv((V + 3) / 4, NRED_THREADS) : 0;
    DenseAdam opt;
    opt.on = 1; opt.p = flat_p; opt.m = flat_m; opt.v = flat_v; opt.g = flat_g;
    opt.lo0 = lo0; opt.hi0 = hi0; opt.lo1 = lo1; opt.hi1 = hi1;
    opt.s = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    const int64_t longest = hi0 - lo0 > hi1 - lo1 ? hi0 - lo0 : hi1 - lo1;
    narre_reduce_kernel<<<dim3(red_blocks + comp_blocks + (int)cdiv(longest, NRED_THREADS), 2), NRED_THREADS, 0, st>>>(
        wa, red_blocks, comp_blocks, nx, opt);

    float *rp[2], *rm[2], *rv[2];
    for (int k = 0; k < 2; ++k) {
        rp[k] = reinterpret_cast<float *>(rows_p[k]); rm[k] = reinterpret_cast<float *>(rows_m[k]);
        rv[k] = reinterpret_cast<float *>(rows_v[k]);
        R4R_REQUIRE(rp[k] && rm[k] && rv[k], "deepconnpp_step: bias vector %d: null parameter / moment pointer", k);
    }
    return mf_bias_rows_launch(rp[0], rm[0], rv[0], rp[1], rm[1], rv[1], n_users, n_items, uid, iid, w.g, w.tag[0], w.tag[1], B,
                               (int)adam_step, opt.s, st);
}

// ------------------------------------------------------------------------------ TransNet(++)
extern "C" int r4r_transnet_nparam(void) { return TN_COUNT; }

extern "C" int r4r_transnet_layout(int E, int L, int plus, int64_t *offsets, int64_t *sizes, int64_t *total) {
    R4R_REQUIRE(offsets && sizes && total, "transnet_layout: null pointer");
    R4R_REQUIRE(E > 0 && L > 0 && L <= NR_MAX_L, "transnet_layout: bad sizes");
    const TLayout lay = tn_layout(E, L, plus);
    for (int i = 0; i < TN_COUNT; ++i) { offsets[i] = lay.off[i]; sizes[i] = lay.size[i]; }
    *total = lay.total;
    return R4R_OK;
}

extern "C" size_t r4r_transnet_ws_bytes(int64_t B, int T, int E, int L, int plus, int64_t V, int64_t n_users, int64_t n_items) {
    if (B < 0 || T <= 0 || E <= 0 || L <= 0 || V <= 0 || n_users <= 0 || n_items <= 0) return 0;
    return tn_carve(nullptr, B, T, E, L, plus, V, n_users, n_items).bytes;
}

// which: 0 dropout multipliers [B, 5L + 10]; 1 / 2 compact gradient rows of the user / item ID vectors
// [B, 5]; 3 the per-rating auxiliary outputs [B, 3]; 4 the SIZE of the persistent head of the workspace
// (row and chunk tags: zero once, carry over when switching buffers); 6 + 2 * tower + buffer: a token
// buffer's counter (towers 0 user, 1 item, 2 this review)
extern "C" size_t r4r_transnet_ws_offset(int64_t B, int T, int E, int L, int plus, int64_t V, int64_t n_users,
                                         int64_t n_items, int which) {
    const TnWs w = tn_carve(reinterpret_cast<void *>(256), B, T, E, L, plus, V, n_users, n_items);
    if (which == 4) return w.persist;
    if (which >= 6 && which < 12)
        return (size_t)(reinterpret_cast<char *>(w.count[(which - 6) & 1][(which - 6) >> 1]) - reinterpret_cast<char *>(256));
    const char *q = which == 0 ? reinterpret_cast<char *>(w.mult) : which == 1 ? reinterpret_cast<char *>(w.grow[0])
                  : which == 2 ? reinterpret_cast<char *>(w.grow[1]) : reinterpret_cast<char *>(w.aux);
    return (size_t)(q - reinterpret_cast<char *>(256));
}

extern "C" int r4r_transnet_step(const float *table, int64_t V,
                                 const int64_t *user_idx, const int64_t *item_idx, const int64_t *this_idx,
                                 const int64_t *uid, const int64_t *iid, const float *y,
                                 float *flat_p, float *flat_g, float *flat_m, float *flat_v,
                                 const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                                 int64_t n_users, int64_t n_items,
                                 float *pred, float *se, float *sse_accum, void *ws, size_t ws_bytes,
                                 int64_t B, int T, int E, int L, int plus,
                                 float dropout_p, int training, uint64_t seed, uint64_t offset, float inv_denom,
                                 int conv_algo, int token_buffer, int tokens_ready,
                                 const int64_t *next_user_idx, const int64_t *next_item_idx, const int64_t *next_this_idx,
                                 float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                                 void *stream) {
    R4R_REQUIRE(table && user_idx && item_idx && this_idx && flat_p && pred && ws, "transnet_step: null pointer");
    R4R_REQUIRE(!plus || (uid && iid && rows_p), "transnet_step: TransNet++ needs the ids and the ID-vector tables");
    R4R_REQUIRE(V > 0 && B >= 0 && T > 0 && n_users > 0 && n_items > 0, "transnet_step: bad sizes");
    R4R_REQUIRE(L > 0 && L <= NR_MAX_L, "transnet_step: latent_size %d outside 1..%d", L, NR_MAX_L);
    R4R_REQUIRE(E > 0 && E % 4 == 0, "transnet_step: word_embed_size %d must be a positive multiple of 4", E);
    const bool train_step = flat_g != nullptr;
    R4R_REQUIRE(!train_step || (y && se && flat_m && flat_v && adam_step >= 1 && (!plus || (rows_m && rows_v))),
                "transnet_step: a training step needs ratings, se, gradient / moment buffers and adam_step >= 1");
    R4R_REQUIRE(!y || se, "transnet_step: se buffer required when y is given");
    R4R_REQUIRE(!next_user_idx == !next_item_idx && !next_user_idx == !next_this_idx,
                "transnet_step: the three next-batch index arrays go together");
    R4R_REQUIRE(!next_user_idx || train_step, "transnet_step: the next batch's tokens ride on the backward launches");
    R4R_REQUIRE(token_buffer == 0 || token_buffer == 1, "transnet_step: token_buffer must be 0 or 1");
    R4R_REQUIRE(adam_step < (1ll << 31), "transnet_step: step tag overflow");
    R4R_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "transnet_step: dropout %f outside [0,1)", (double)dropout_p);
    R4R_REQUIRE(B * (int64_t)((T + 2 + 127) / 128) < (1ll << 31), "transnet_step: grid too large");
    if (ws_bytes < r4r_transnet_ws_bytes(B, T, E, L, plus, V, n_users, n_items)) {
        set_error("transnet_step: workspace %zu < %zu bytes", ws_bytes,
                  r4r_transnet_ws_bytes(B, T, E, L, plus, V, n_users, n_items));
        return R4R_ERR_WORKSPACE;
    }
    if (B == 0) return R4R_OK;
    hipStream_t st = as_stream(stream);
    const TLayout lay = tn_layout(E, L, plus);
    R4R_REQUIRE(lay.total < (1ll << 31), "transnet_step: dense parameter buffer too large");
    const int64_t lo = lay.off[TN_UFW], hi = lay.total;
    const int nhp = (int)(hi - lo);
    const TnWs w = tn_carve(ws, B, T, E, L, plus, V, n_users, n_items);
    const float *P[TN_COUNT];
    float *G[TN_COUNT];
    for (int i = 0; i < TN_COUNT; ++i) { P[i] = flat_p + lay.off[i]; G[i] = flat_g ? flat_g + lay.off[i] : nullptr; }
    const int cw[3] = {TN_UCW, TN_ICW, TN_TCW}, cb[3] = {TN_UCB, TN_ICB, TN_TCB};

    const int64_t *idx[3] = {user_idx, item_idx, this_idx};
    const int algo = textcnn_pick_algo(conv_algo, B, T, E, NF);
    int tiles;
    if (algo == R4R_CONV_PROJECT) {
        ProjTower pt[3];
        for (int t = 0; t < 3; ++t) {
            pt[t].idx = idx[t];
            pt[t].conv_w = P[cw[t]]; pt[t].conv_b = P[cb[t]];
            pt[t].flags = w.flags[token_buffer][t]; pt[t].slot = w.slot[token_buffer][t];
            pt[t].list = w.list[token_buffer][t]; pt[t].count = w.count[token_buffer][t];
            pt[t].ptab = w.ptab[t]; pt[t].pmax = w.pmax[t]; pt[t].parg = w.parg[t];
        }
        if (!tokens_ready)
            if (int rc = textcnn_proj_tokens_launch(V, pt, 3, B, T, /*zero_state=*/false, st)) return rc;
        if (int rc = textcnn_proj_compute_launch(table, V, pt, 3, B, T, E, NF, st)) return rc;
        tiles = proj_tiles(T);
    } else {
        FwdTower ft[3];
        for (int t = 0; t < 3; ++t) {
            ft[t].idx = idx[t];
            ft[t].conv_w = P[cw[t]]; ft[t].conv_b = P[cb[t]];
            ft[t].wp = w.wp[t]; ft[t].pmax = w.pmax[t]; ft[t].parg = w.parg[t];
        }
        if (int rc = textcnn_fwd_launch(table, ft, 3, B, T, E, NF, st)) return rc;
        tiles = textcnn_tiles(T);
    }

    TnHead h;
    for (int t = 0; t < 3; ++t) {
        h.pmax[t] = w.pmax[t]; h.parg[t] = w.parg[t];
        h.pooled[t] = w.pooled[t]; h.argmax[t] = w.argmax[t]; h.g_pooled[t] = w.g_pooled[t];
    }
    float *rp[2] = {nullptr, nullptr}, *rm[2] = {nullptr, nullptr}, *rv[2] = {nullptr, nullptr};
    for (int t = 0; t < 2; ++t) {
        if (plus) {
            rp[t] = reinterpret_cast<float *>(rows_p[t]);
            R4R_REQUIRE(rp[t], "transnet_step: null ID-vector table");
            if (train_step) {
                rm[t] = reinterpret_cast<float *>(rows_m[t]); rv[t] = reinterpret_cast<float *>(rows_v[t]);
                R4R_REQUIRE(rm[t] && rv[t], "transnet_step: ID-vector table %d: null moment pointer", t);
            }
        }
        h.emb[t] = rp[t]; h.tag[t] = w.tag[t]; h.ctag[t] = w.ctag[t]; h.grow[t] = w.grow[t];
    }
    h.id[0] = uid; h.id[1] = iid; h.flat_p = flat_p;
    for (int i = 0; i < TN_COUNT; ++i) h.off[i] = (int)lay.off[i];
    h.lo = (int)lo;
    h.y = y; h.part = w.part; h.mult = w.mult; h.pred = pred; h.se = se; h.aux = w.aux;
    h.B = B; h.L = L; h.tiles = tiles; h.nhp = nhp; h.training = training; h.want_grad = train_step;
    h.now = (int)adam_step; h.plus = plus; h.p_drop = dropout_p; h.inv_denom = inv_denom; h.seed = seed; h.offset = offset;
    if (L <= 16) tn_head_kernel<16><<<(unsigned)B, 256, 0, st>>>(h);
    else tn_head_kernel<32><<<(unsigned)B, 256, 0, st>>>(h);
    if (!train_step) return check_launch("transnet_step(forward)");

    WgradTower wt[3];
    WgradArgs wa;
    for (int t = 0; t < 3; ++t) {
        wt[t].idx = idx[t]; wt[t].g_pooled = w.g_pooled[t]; wt[t].argmax = w.argmax[t];
        wt[t].part_w = w.part_w[t]; wt[t].part_b = w.part_b[t];
        wt[t].d_w = G[cw[t]]; wt[t].d_b = G[cb[t]];
    }
    for (int k = 0; k < MAX_TOWERS; ++k) wa.t[k] = wt[k < 3 ? k : 0];
    wa.table = table; wa.N = B; wa.T = T; wa.E = E; wa.F = NF;
    wa.nsplit = textcnn_wgrad_splits(B);
    wa.per_split = (int)cdiv(B, wa.nsplit);
    ColSum cs;
    cs.part = w.part; cs.se = se; cs.flat_g = flat_g; cs.sse_accum = sse_accum; cs.B = B; cs.nhp = nhp;
    cs.col0_lo = (int)lo; cs.col0_n = nhp; cs.col1_lo = (int)hi;
    cs.aux = w.aux; cs.inv_denom = inv_denom;                // sse_accum: [sum source SE, sum of batch-mean target SE, sum of batch-mean transform loss]
    const int cs_blocks = (nhp + 3 + CS_COLS - 1) / CS_COLS;
    const bool prefetch = next_user_idx && algo == R4R_CONV_PROJECT;
    TokenArgs nx{};
    if (prefetch) {
        ProjTower nt[3];
        const int64_t *nidx[3] = {next_user_idx, next_item_idx, next_this_idx};
        const int ob = token_buffer ^ 1;
        for (int t = 0; t < 3; ++t) {
            nt[t] = ProjTower{};
            nt[t].idx = nidx[t];
            nt[t].flags = w.flags[ob][t]; nt[t].slot = w.slot[ob][t]; nt[t].list = w.list[ob][t]; nt[t].count = w.count[ob][t];
        }
        nx = make_token_args(V, nt, 3, B, T);
    }
    const int packed = 3 * E / 4 <= 64;
    narre_backward_kernel<0><<<dim3(packed ? (NF + 3) / 4 : NF, wa.nsplit, prefetch ? 5 : 4), WG_THREADS, 0, st>>>(
        wa, cs, cs_blocks, nx, packed, RowSweep{}, 0, 3);

    const int red_blocks = (NF * 3 * E + NF + NRED_THREADS - 1) / NRED_THREADS;
    const int comp_blocks = prefetch ? (int)cdiv((V + 3) / 4, NRED_THREADS) : 0;
    DenseAdam opt;
    opt.on = 1; opt.p = flat_p; opt.m = flat_m; opt.v = flat_v; opt.g = flat_g;
    opt.lo0 = lo; opt.hi0 = hi; opt.lo1 = hi; opt.hi1 = hi;       // every head parameter in one range (tower 0's slice)
    opt.s = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    narre_reduce_kernel<<<dim3(red_blocks + comp_blocks + (int)cdiv(hi - lo, NRED_THREADS), 3), NRED_THREADS, 0, st>>>(
        wa, red_blocks, comp_blocks, nx, opt);
    if (!plus) return check_launch("transnet_step");
    return mf_table_rows_launch(rp[0], rm[0], rv[0], rp[1], rm[1], rv[1], n_users, n_items, TN_ID, uid, iid, w.grow[0], w.grow[1],
                                w.tag[0], w.tag[1], w.ctag[0], w.ctag[1], B, (int)adam_step, opt.s, st);
}
