// Fused native training step for TransNet / TransNet++ (launch roles shared with the other review
// models: step_device.h).
#include "step_device.h"
#include "trace_device.h"

namespace r4r {

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
        o += (sz[i] + 31) & ~(int64_t)31;
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
    const float *embm[2], *embv[2];                 // their Adam moments (scheduled sweep: pending updates are applied on the way)
    MfTimeBlock tb; AdamScalars sc0;                // the schedule (rows_device.h; tb.rlast_u == NULL: nothing can be pending)
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

HEAD_TRACE_DEFINE(r4r_debug_tn_head_trace)
BWD_TRACE_DEFINE(r4r_debug_tn_bwd_trace)
// One workgroup of 256 threads per rating.
// (a pointer array of the kernel arguments indexed by a run-time tower / side makes every lane FETCH the pointer from
// the argument segment, a round trip in front of the access it serves: select between the constant-index elements)
#define SEL2(arr, s) ((s) ? (arr)[1] : (arr)[0])
#define SEL3(arr, s) ((s) == 0 ? (arr)[0] : ((s) == 1 ? (arr)[1] : (arr)[2]))
template <int ML>
__global__ __launch_bounds__(256) void tn_head_kernel(TnHead a) {
    HEAD_STAMP(0)
    constexpr int MS = ML + 2 * TN_ID;                      // widest source_fm input
    __shared__ __attribute__((aligned(16))) float P[3][NF];
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
    // ---- S0: weights, pool finish (max over tiles, relu, first argmax), ID vectors.  Every global read of
    // this prologue is issued into registers before anything waits (a load -> LDS-store loop is one
    // memory round trip per iteration, and the tile loop of the pool finish one per tile: 14.5 us
    // of a 26 us kernel at cfg5 -- tools/head_trace.py)
    // (the kernel is bound by vector-ALU issue: the towers' FC weights move as 16-byte units -- NF / 4 = 25 filter quads
    // per row -- a quarter of the loads and index computations)
    constexpr int NQ4 = NF / 4;
    typedef float hq4 __attribute__((ext_vector_type(4)));
    constexpr int WREG = (ML * NQ4 + 255) / 256, AREG = (ML * 2 * ML + 255) / 256, FREG = (ML * ML + 255) / 256,
                  SREG = (MS * TN_FM_K + 255) / 256, TREG = (ML * TN_FM_K + 255) / 256, PT = 8;
    hq4 wreg[3][WREG];
    float av[AREG], f2v[FREG], vsv[SREG], vtv[TREG];
    const int wtot = L * NQ4;
    // What later reads depend on goes out FIRST (loads return in order: waiting for the oldest requests does not wait for
    // the weights behind them): the rating's ids -- the ID vectors need them, a second round trip that now runs under
    // the first -- and the pool partials of this thread's (tower, filter) items, first PT tiles each (as a loop behind
    // the weights' LDS writes they were a third round trip).
    const bool idt = a.plus && tid >= 64 && tid < 64 + 2 * TN_ID;
    int64_t id_r = 0;
    if (idt) id_r = SEL2(a.id, (tid - 64) >= TN_ID)[b];
    constexpr int PITEMS = (3 * NF + 255) / 256;            // (tower, filter) items per thread: 2
    float pv0[PITEMS][PT];
    int pp0[PITEMS][PT];
#pragma unroll
    for (int it = 0; it < PITEMS; ++it) {
        const int i = min(tid + 256 * it, 3 * NF - 1), s = i / NF, f = i - s * NF;
#pragma unroll
        for (int k = 0; k < PT; ++k) {
            const size_t q = ((size_t)b * a.tiles + min(k, a.tiles - 1)) * NP + f;   // (a clamped duplicate never wins: strict >)
            pv0[it][k] = SEL3(a.pmax, s)[q];
            pp0[it][k] = SEL3(a.parg, s)[q];
        }
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const float *src = fp + (s == 0 ? a.off[TN_UFW] : (s == 1 ? a.off[TN_IFW] : a.off[TN_TFW]));
#pragma unroll
        for (int u = 0; u < WREG; ++u)
            wreg[s][u] = *reinterpret_cast<const hq4 *>(src + 4 * min(tid + 256 * u, wtot - 1));   // (clamped: rounds past the end re-read the last quad, unused)
    }
#pragma unroll
    for (int u = 0; u < AREG; ++u) av[u] = 256 * u < L * L2 ? fp[a.off[TN_P0W] + min(tid + 256 * u, L * L2 - 1)] : 0.f;
#pragma unroll
    for (int u = 0; u < FREG; ++u) f2v[u] = 256 * u < L * L ? fp[a.off[TN_P2W] + min(tid + 256 * u, L * L - 1)] : 0.f;
#pragma unroll
    for (int u = 0; u < SREG; ++u) vsv[u] = 256 * u < ns * TN_FM_K ? fp[a.off[TN_SV] + min(tid + 256 * u, ns * TN_FM_K - 1)] : 0.f;
#pragma unroll
    for (int u = 0; u < TREG; ++u) vtv[u] = 256 * u < L * TN_FM_K ? fp[a.off[TN_TV] + min(tid + 256 * u, L * TN_FM_K - 1)] : 0.f;
    const float lws_r = fp[a.off[TN_SLW] + min(tid, ns - 1)];
    const int tl = min(tid, L - 1), t3 = min(tid, L3 - 1), s3 = t3 / L;
    const float lwt_r = fp[a.off[TN_TLW] + tl], b0_r = fp[a.off[TN_P0B] + tl], b2_r = fp[a.off[TN_P2B] + tl];
    const float fcb_r = fp[(s3 == 0 ? a.off[TN_UFB] : (s3 == 1 ? a.off[TN_IFB] : a.off[TN_TFB])) + (t3 - s3 * L)];
    const float m0 = fp[a.off[TN_SLB]], m1 = fp[a.off[TN_TLB]];
    float idv = 0.f;
    if (idt) {
        const int k = tid - 64, s = k >= TN_ID, c = k - s * TN_ID;
        const int64_t r = id_r, e = r * TN_ID + c;
        idv = SEL2(a.emb, s)[e];
        if (a.tb.rlast_u) {                                 // the element's pending gradient-zero updates (not written back)
            float mq = SEL2(a.embm, s)[e], vq = SEL2(a.embv, s)[e];
            const int cur = tb_current(a.tb, s ? a.tb.rlast_i : a.tb.rlast_u, e, r, a.now);
#pragma unroll
            for (int j = 0; j < MF_TB_MAX - 1; ++j) {       // steps now - 7 .. now - 1
                AdamScalars sc = a.sc0;
                sc.lr_over_bc1 = a.tb.lr_bc1[j];
                sc.inv_sqrt_bc2 = a.tb.isb2[j];
                if (a.now - (MF_TB_MAX - 1 - j) > cur) adam_elem_fast(idv, 0.f, mq, vq, sc);
            }
        }
    }
    // pool finish: thread i < 3 NF owns (tower, filter) i; threads 0 .. 3 NF - 257 a second one
#pragma unroll
    for (int it = 0; it < PITEMS; ++it) {
        const int i = tid + 256 * it;
        if (i >= 3 * NF) break;
        const int s = i / NF, f = i - s * NF;
        float best = -INFINITY;
        int bp = -1;
#pragma unroll
        for (int k = 0; k < PT; ++k)
            if (pv0[it][k] > best) { best = pv0[it][k]; bp = pp0[it][k]; }
        for (int k0 = PT; k0 < a.tiles; k0 += PT) {          // documents of more than PT tiles: the rest, PT at a time
            float v[PT];
            int pp[PT];
#pragma unroll
            for (int k = 0; k < PT; ++k) {
                const bool in = k0 + k < a.tiles;
                const size_t q = ((size_t)b * a.tiles + (in ? k0 + k : 0)) * NP + f;
                v[k] = in ? SEL3(a.pmax, s)[q] : -INFINITY;
                pp[k] = SEL3(a.parg, s)[q];
            }
#pragma unroll
            for (int k = 0; k < PT; ++k)
                if (v[k] > best) { best = v[k]; bp = pp[k]; }
        }
        if (!(best > 0.f)) { best = 0.f; bp = -1; }
        P[s][f] = best;
        SEL3(a.pooled, s)[b * NF + f] = best;
        SEL3(a.argmax, s)[b * NF + f] = bp;
    }
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int u = 0; u < WREG; ++u) {
            const int r = tid + 256 * u;
            if (r < wtot) {
                const int l = r / NQ4, f0 = 4 * (r - l * NQ4);
#pragma unroll
                for (int c = 0; c < 4; ++c) fcw[s][l][f0 + c] = wreg[s][u][c];   // (rows of NF + 1 floats: four 4-byte writes)
            }
        }
#pragma unroll
    for (int u = 0; u < AREG; ++u) {
        const int i = tid + 256 * u;
        if (i < L * L2) { const int k = qd(i, invL2); W0[k][i - k * L2] = av[u]; }
    }
#pragma unroll
    for (int u = 0; u < FREG; ++u) {
        const int i = tid + 256 * u;
        if (i < L * L) { const int k = qd(i, invL); W2[k][i - k * L] = f2v[u]; }
    }
#pragma unroll
    for (int u = 0; u < SREG; ++u) {
        const int i = tid + 256 * u;
        if (i < ns * TN_FM_K) Vs[i / TN_FM_K][i % TN_FM_K] = vsv[u];
    }
#pragma unroll
    for (int u = 0; u < TREG; ++u) {
        const int i = tid + 256 * u;
        if (i < L * TN_FM_K) Vt[i / TN_FM_K][i % TN_FM_K] = vtv[u];
    }
    if (tid < ns) lws[tid] = lws_r;
    if (tid < L) { lwt[tid] = lwt_r; b0s[tid] = b0_r; b2s[tid] = b2_r; }
    if (tid < L3) fcb[tid] = fcb_r;
    if (tid == 0) { misc[0] = m0; misc[1] = m1; }
    if (idt) {                                              // dropout.user / dropout.item on the ID vectors
        const int k = tid - 64;
        const float m = draw(5 * L + k);
        finm[k] = m;
        fin[k] = idv * m;
    }
    __syncthreads();
    HEAD_STAMP(1)
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
    HEAD_STAMP(2)
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
    HEAD_STAMP(3)
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
    HEAD_STAMP(4)
    // ---- S4: the two factorisation machines (common_pytorch_models.py:49-57): wave 0 source, wave 1 target
    if (wv < 2) {
        // input i = lane + 64 u: one per lane, two in the 64-wide instantiation (TransNet++'s source FM reads
        // latent_size + 10 inputs, up to 74)
        constexpr int NI = (MS + 63) / 64;
        const int n = wv ? L : ns;
        float xi[NI], lw[NI], gacc[NI];
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = lane + 64 * u;
            xi[u] = i < n ? (wv ? tir[i] : fin[i]) : 0.f;
            lw[u] = i < n ? (wv ? lwt[i] : lws[i]) : 0.f;
            gacc[u] = 0.f;
        }
        float inter = 0.f;
#pragma unroll
        for (int k = 0; k < TN_FM_K; ++k) {
            float v[NI], pv = 0.f, pv2 = 0.f;
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                const int i = lane + 64 * u;
                v[u] = i < n ? (wv ? Vt[i][k] : Vs[i][k]) : 0.f;
                pv += xi[u] * v[u];
                pv2 += xi[u] * xi[u] * v[u] * v[u];
            }
            const float s = wave_sum(pv);
            const float s2 = wave_sum(pv2);
            inter += s * s - s2;
#pragma unroll
            for (int u = 0; u < NI; ++u) gacc[u] += s * v[u] - xi[u] * v[u] * v[u];
            if (lane == 0) (wv ? skt : sks)[k] = s;
        }
        float pl = 0.f;
#pragma unroll
        for (int u = 0; u < NI; ++u) pl += xi[u] * lw[u];
        const float lin = wave_sum(pl);
        const float out = 0.5f * inter + (lin + misc[wv]);
#pragma unroll
        for (int u = 0; u < NI; ++u)
            if (lane + 64 * u < n) (wv ? dft : dfs)[lane + 64 * u] = gacc[u] + lw[u];      // d FM / d x_i
        if (lane == 0) misc[2 + wv] = out;
    } else if (wv == 2) {
        const float d = lane < L ? sir[lane] - tir[lane] : 0.f;
        const float tr = wave_sum(d * d);
        if (lane == 0) misc[4] = tr;
    }
    __syncthreads();
    HEAD_STAMP(5)
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
                const int64_t r = SEL2(a.id, s)[b];
                SEL2(a.tag, s)[r] = a.now;
                SEL2(a.ctag, s)[r * TN_ID / MF_CHUNK] = a.now;              // (a row can straddle two chunks)
                SEL2(a.ctag, s)[(r * TN_ID + TN_ID - 1) / MF_CHUNK] = a.now;
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
        SEL2(a.grow, s)[b * TN_ID + c] = g_s * dfs[k] * finm[k];
    }
    if (tid >= 128 && tid < 128 + L) {
        const int l = tid - 128;
        dtmp[l] = 2.f * (sir[l] - tir[l]) * a.inv_denom * sirm[l];       // d (project.2 output)
        dz[L2 + l] = g_t * dft[l] * tirm[l] * xm[L2 + l];                // d (target FC output)
    }
    __syncthreads();
    HEAD_STAMP(6)
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
    HEAD_STAMP(7)
    // ---- B3: source.project.0 weight, d cat -> d (source FC outputs)
    for (int i = tid; i < L * L2; i += 256) { const int k = qd(i, invL2); prow[col(a.off[TN_P0W] + i)] = dhs[k] * x[i - k * L2]; }
    if (tid < L2) {
        float acc = 0.f;
        for (int k = 0; k < L; ++k) acc = fmaf(dhs[k], W0[k][tid], acc);
        dz[tid] = acc * xm[tid];
    }
    __syncthreads();
    HEAD_STAMP(8)
    // ---- B4: the towers' FC gradients, d pooled
    if (tid < L3) {
        const int s = tid / L;
        prow[col((s == 0 ? a.off[TN_UFB] : (s == 1 ? a.off[TN_IFB] : a.off[TN_TFB])) + (tid - s * L))] = dz[tid];
    }
    for (int s = 0; s < 3; ++s) {                           // (filter quads: 16-byte reads of the pooled features, 16-byte stores)
        float *dst = prow + col((s == 0 ? a.off[TN_UFW] : (s == 1 ? a.off[TN_IFW] : a.off[TN_TFW])));
        for (int r = tid; r < L * NQ4; r += 256) {
            const int l = r / NQ4, q = r - l * NQ4;
            const hq4 p4 = *reinterpret_cast<const hq4 *>(&P[s][4 * q]);
            *reinterpret_cast<hq4 *>(dst + l * NF + 4 * q) = dz[s * L + l] * p4;
        }
    }
    for (int i = tid; i < 3 * NF; i += 256) {
        const int s = i / NF, f = i - s * NF;
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc = fmaf(dz[s * L + l], fcw[s][l][f], acc);
        SEL3(a.g_pooled, s)[b * NF + f] = acc;
    }
    HEAD_STAMP(9)
}

struct TnWs {
    float *wp[3], *pmax[3]; int *parg[3];
    int *flags[2][3], *slot[2][3], *list[2][3], *count[2][3]; float *ptab[3];
    float *pooled[3]; int *argmax[3]; float *g_pooled[3];
    float *part_w[3], *part_b[3];
    int *tag[2], *ctag[2], *rlast[2], *tb_err;
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
    w.rlast[0] = reinterpret_cast<int *>(take((size_t)n_users * 4));         // the scheduled sweep's state (rows_device.h)
    w.rlast[1] = reinterpret_cast<int *>(take((size_t)n_items * 4));
    w.tb_err = reinterpret_cast<int *>(take(4));
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

// ------------------------------------------------------------------------------ TransNet(++)
extern "C" int r4r_transnet_nparam(void) { return TN_COUNT; }

extern "C" int r4r_transnet_layout(int E, int L, int plus, int64_t *offsets, int64_t *sizes, int64_t *total) {
    R4R_REQUIRE(offsets && sizes && total, "transnet_layout: null pointer");
    R4R_REQUIRE(E > 0 && L > 0 && L <= HEAD_MAX_L, "transnet_layout: bad sizes");
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
// (row and chunk tags, the temporally blocked sweep's pending counts: zero once, carry over when switching
// buffers); 5 the int the sweep sets if more updates were ever pending than a visit applies; 6 + 2 * tower + buffer: a token
// buffer's counter (towers 0 user, 1 item, 2 this review)
extern "C" size_t r4r_transnet_ws_offset(int64_t B, int T, int E, int L, int plus, int64_t V, int64_t n_users,
                                         int64_t n_items, int which) {
    const TnWs w = tn_carve(reinterpret_cast<void *>(256), B, T, E, L, plus, V, n_users, n_items);
    if (which == 4) return w.persist;
    if (which == 5) return (size_t)(reinterpret_cast<char *>(w.tb_err) - reinterpret_cast<char *>(256));
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
                                 int sweep_period, int64_t sweep_base, int sweep_all,
                                 float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                                 void *stream) {
    R4R_REQUIRE(table && user_idx && item_idx && this_idx && flat_p && pred && ws, "transnet_step: null pointer");
    R4R_REQUIRE(sweep_period >= 1 && sweep_period <= MF_TB_MAX, "transnet_step: sweep_period %d outside 1..%d", sweep_period, MF_TB_MAX);
    R4R_REQUIRE(sweep_base >= 0 && (!flat_g || sweep_base < adam_step), "transnet_step: sweep_base outside 0..adam_step - 1");
    R4R_REQUIRE(!plus || (uid && iid && rows_p), "transnet_step: TransNet++ needs the ids and the ID-vector tables");
    R4R_REQUIRE(V > 0 && B >= 0 && T > 0 && n_users > 0 && n_items > 0, "transnet_step: bad sizes");
    R4R_REQUIRE(L > 0 && L <= HEAD_MAX_L, "transnet_step: latent_size %d outside 1..%d", L, HEAD_MAX_L);
    R4R_REQUIRE(E > 0 && E % 4 == 0, "transnet_step: word_embed_size %d must be a positive multiple of 4", E);
    const bool train_step = flat_g != nullptr;
    // flat_m == NULL on a training step: gradients only (flat_g, the compact ID-vector rows) -- the data-parallel
    // form, where the exchange sits between gradient and update (r4r_adam_multi + r4r_transnet_rows_apply)
    const bool apply = flat_m != nullptr;
    R4R_REQUIRE(!train_step || (y && se && adam_step >= 1 && (!apply || (flat_v && (!plus || (rows_m && rows_v))))),
                "transnet_step: a training step needs ratings, se, gradient buffers and adam_step >= 1 (+ moments to update)");
    R4R_REQUIRE(!y || se, "transnet_step: se buffer required when y is given");
    R4R_REQUIRE(!next_user_idx == !next_item_idx && !next_user_idx == !next_this_idx,
                "transnet_step: the three next-batch index arrays go together");
    R4R_REQUIRE(!next_user_idx || train_step, "transnet_step: the next batch's tokens ride on the backward launches");
    R4R_REQUIRE(token_buffer == 0 || token_buffer == 1, "transnet_step: token_buffer must be 0 or 1");
    R4R_REQUIRE(adam_step < (1ll << 31), "transnet_step: step tag overflow");
    R4R_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "transnet_step: dropout %f outside [0,1)", (double)dropout_p);
    R4R_REQUIRE(B * (int64_t)((T + 2 + 127) / 128) < (1ll << 31), "transnet_step: grid too large");
    R4R_REQUIRE(!(plus && train_step) || B <= 32768, "transnet_step: batch %lld > 32768 (the ID-vector sweep keeps a side's ids "
                "in LDS; use the module path for larger batches)", (long long)B);
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
            pt[t].ptab = w.ptab[t]; pt[t].pmax = w.pmax[t]; pt[t].parg = w.parg[t]; pt[t].wimg = w.wp[t];
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
            if (train_step && (apply || (rows_m && rows_v))) {   // (gradients only: the moments still serve the catch-up)
                rm[t] = reinterpret_cast<float *>(rows_m[t]); rv[t] = reinterpret_cast<float *>(rows_v[t]);
                R4R_REQUIRE(rm[t] && rv[t], "transnet_step: ID-vector table %d: null moment pointer", t);
            }
        }
        h.emb[t] = rp[t]; h.tag[t] = w.tag[t]; h.ctag[t] = w.ctag[t]; h.grow[t] = w.grow[t];
    }
    h.id[0] = uid; h.id[1] = iid; h.flat_p = flat_p;
    // the scheduled sweep over the ID-vector tables (rows_device.h): the rows a rating reads catch up in registers
    MfTimeBlock tb{};
    h.tb = tb; h.sc0 = AdamScalars{};
    const bool sched = plus && train_step && rm[0] && rm[1] &&
                       ((reinterpret_cast<uintptr_t>(rp[0]) | reinterpret_cast<uintptr_t>(rp[1]) | reinterpret_cast<uintptr_t>(rm[0]) |
                         reinterpret_cast<uintptr_t>(rm[1]) | reinterpret_cast<uintptr_t>(rv[0]) | reinterpret_cast<uintptr_t>(rv[1])) & 15) == 0;
    if (sched) {
        tb.rlast_u = w.rlast[0]; tb.rlast_i = w.rlast[1]; tb.err = w.tb_err; tb.base = (int)sweep_base;
        tb.period = sweep_period; tb.flush = (sweep_all || sweep_period == 1) ? 1 : 0; tb.inc = 1;
        mf_time_block_scalars(tb, lr, beta1, beta2, eps, weight_decay, adam_step);
        if (sweep_period > 1) {
            h.tb = tb; h.sc0 = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
            for (int t = 0; t < 2; ++t) { h.embm[t] = rm[t]; h.embv[t] = rv[t]; }
        }
    }
    for (int i = 0; i < TN_COUNT; ++i) h.off[i] = (int)lay.off[i];
    h.lo = (int)lo;
    h.y = y; h.part = w.part; h.mult = w.mult; h.pred = pred; h.se = se; h.aux = w.aux;
    h.B = B; h.L = L; h.tiles = tiles; h.nhp = nhp; h.training = training; h.want_grad = train_step;
    h.now = (int)adam_step; h.plus = plus; h.p_drop = dropout_p; h.inv_denom = inv_denom; h.seed = seed; h.offset = offset;
    if (L <= 16) tn_head_kernel<16><<<(unsigned)B, 256, 0, st>>>(h);
    else if (L <= 32) tn_head_kernel<32><<<(unsigned)B, 256, 0, st>>>(h);
    else tn_head_kernel<64><<<(unsigned)B, 256, 0, st>>>(h);          // latent_size 33 .. 64 (hyper_params.py:63 has no bound)
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
    wa.table_bytes = (int64_t)V * E * 4;                   // (the wide wgrad reads the rows through a buffer resource)
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
    const int gx = packed ? (NF + 3) / 4 : NF;
    const dim3 bgrid(gx, wa.nsplit, 3 + backward_cs_slices(cs_blocks, gx * wa.nsplit) + (prefetch ? 1 : 0));
    if (packed) narre_backward_kernel<0, false><<<bgrid, WG_THREADS, 0, st>>>(BackwardArgs{wa, cs, cs_blocks, nx, 1, RowSweep{}, 0, 3});
    else narre_backward_kernel<0, true><<<bgrid, WG_THREADS, 0, st>>>(BackwardArgs{wa, cs, cs_blocks, nx, 0, RowSweep{}, 0, 3});

    const int red_blocks = (NF * 3 * E + NF + NRED_THREADS - 1) / NRED_THREADS;
    const int comp_blocks = prefetch ? (int)cdiv((V + 3) / 4, NRED_THREADS * compact_groups(V)) : 0;
    DenseAdam opt;
    opt.on = apply ? 1 : 0; opt.p = flat_p; opt.m = flat_m; opt.v = flat_v; opt.g = flat_g;
    opt.lo0 = lo; opt.hi0 = hi; opt.lo1 = hi; opt.hi1 = hi;       // every head parameter in one range (tower 0's slice)
    opt.s = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    const int opt_blocks = apply ? (int)cdiv(hi - lo, NRED_THREADS) : 0;
    narre_reduce_kernel<<<dim3(red_blocks + comp_blocks + opt_blocks, 3), NRED_THREADS, 0, st>>>(
        wa, red_blocks, comp_blocks, nx, opt);
    if (!plus || !apply) return check_launch("transnet_step");
    // the ID-vector tables: the scheduled sweep (unaligned tables: the plain tagged sweep)
    return mf_table_rows_launch(rp[0], rm[0], rv[0], rp[1], rm[1], rv[1], n_users, n_items, TN_ID, uid, iid, w.grow[0], w.grow[1],
                                w.tag[0], w.tag[1], w.ctag[0], w.ctag[1], B, (int)adam_step, opt.s, st, sched ? &tb : nullptr);
}

// What the temporally blocked sweep left pending (r4r_transnet_step with next_uid and sweep_period > 1): every chunk
// of the two ID-vector tables takes its pending updates now.  adam_step = the LAST COMPLETED step; same optimiser
// scalars as the steps that deferred; `ws` and the shape arguments are the step's.
extern "C" int r4r_transnet_rows_flush(const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                                       int64_t n_users, int64_t n_items, void *ws, size_t ws_bytes,
                                       int64_t B, int T, int E, int L, int64_t V,
                                       int sweep_period, int64_t sweep_base,
                                       float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                                       void *stream) {
    R4R_REQUIRE(rows_p && rows_m && rows_v && ws, "transnet_rows_flush: null pointer");
    R4R_REQUIRE(adam_step >= 0 && adam_step < (1ll << 31), "transnet_rows_flush: bad adam_step");
    R4R_REQUIRE(sweep_period >= 1 && sweep_period <= MF_TB_MAX && sweep_base >= 0 && sweep_base <= adam_step,
                "transnet_rows_flush: sweep_period outside 1..%d or sweep_base outside 0..adam_step", MF_TB_MAX);
    if (ws_bytes < r4r_transnet_ws_bytes(B, T, E, L, 1, V, n_users, n_items)) {
        set_error("transnet_rows_flush: workspace %zu < %zu bytes", ws_bytes, r4r_transnet_ws_bytes(B, T, E, L, 1, V, n_users, n_items));
        return R4R_ERR_WORKSPACE;
    }
    if (adam_step == 0 || sweep_base == adam_step) return R4R_OK;   // no step yet / nothing can be pending
    const TnWs w = tn_carve(ws, B, T, E, L, 1, V, n_users, n_items);
    float *rp[2], *rm[2], *rv[2];
    for (int t = 0; t < 2; ++t) {
        rp[t] = reinterpret_cast<float *>(rows_p[t]); rm[t] = reinterpret_cast<float *>(rows_m[t]);
        rv[t] = reinterpret_cast<float *>(rows_v[t]);
        R4R_REQUIRE(rp[t] && rm[t] && rv[t], "transnet_rows_flush: ID-vector table %d: null pointer", t);
    }
    if ((rows_p[0] | rows_p[1] | rows_m[0] | rows_m[1] | rows_v[0] | rows_v[1]) & 15) return R4R_OK;   // (unaligned tables never defer)
    MfTimeBlock tb{};
    tb.rlast_u = w.rlast[0]; tb.rlast_i = w.rlast[1]; tb.err = w.tb_err; tb.base = (int)sweep_base;
    tb.period = sweep_period; tb.flush = 1; tb.inc = 0;
    mf_time_block_scalars(tb, lr, beta1, beta2, eps, weight_decay, adam_step);
    const AdamScalars sc = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    return mf_table_rows_launch(rp[0], rm[0], rv[0], rp[1], rm[1], rv[1], n_users, n_items, TN_ID, nullptr, nullptr, nullptr, nullptr,
                                w.tag[0], w.tag[1], w.ctag[0], w.ctag[1], 0, (int)adam_step, sc, as_stream(stream), &tb);
}

// Data parallel, TransNet++: the ID-vector update from ALL ranks' compact rows (gathered by the
// caller in rank order; ids -1 pad ragged shards), after a gradients-only r4r_transnet_step
// (flat_m == NULL).  `ws` and the shape arguments are the step's: the row / chunk tags live there.
namespace r4r {
__global__ void tn_tag_rows_kernel(const int64_t *uid, const int64_t *iid, int64_t n, int *tag_u, int *tag_i, int *ctag_u,
                                   int *ctag_i, int now) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int64_t u = uid[e], i = iid[e];
    if (u < 0) return;
    tag_u[u] = now; tag_i[i] = now;
    ctag_u[u * TN_ID / MF_CHUNK] = now; ctag_u[(u * TN_ID + TN_ID - 1) / MF_CHUNK] = now;
    ctag_i[i * TN_ID / MF_CHUNK] = now; ctag_i[(i * TN_ID + TN_ID - 1) / MF_CHUNK] = now;
}
}  // namespace r4r

extern "C" int r4r_transnet_rows_apply(const int64_t *uid_all, const int64_t *iid_all, const float *gu_all,
                                       const float *gi_all, int sweep_period, int64_t sweep_base, int sweep_all, int64_t B_all,
                                       const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                                       int64_t n_users, int64_t n_items, void *ws, size_t ws_bytes,
                                       int64_t B, int T, int E, int L, int64_t V,
                                       float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                                       void *stream) {
    R4R_REQUIRE(uid_all && iid_all && gu_all && gi_all && rows_p && rows_m && rows_v && ws, "transnet_rows_apply: null pointer");
    R4R_REQUIRE(B_all >= 0 && B_all <= 32768, "transnet_rows_apply: %lld gathered ratings outside 0..32768", (long long)B_all);
    R4R_REQUIRE(sweep_period >= 1 && sweep_period <= MF_TB_MAX, "transnet_rows_apply: sweep_period outside 1..%d", MF_TB_MAX);
    R4R_REQUIRE(adam_step >= 1 && adam_step < (1ll << 31) && sweep_base >= 0 && sweep_base < adam_step,
                "transnet_rows_apply: bad adam_step, or sweep_base outside 0..adam_step - 1");
    if (ws_bytes < r4r_transnet_ws_bytes(B, T, E, L, 1, V, n_users, n_items)) {
        set_error("transnet_rows_apply: workspace %zu < %zu bytes", ws_bytes, r4r_transnet_ws_bytes(B, T, E, L, 1, V, n_users, n_items));
        return R4R_ERR_WORKSPACE;
    }
    if (B_all == 0) return R4R_OK;
    hipStream_t st = as_stream(stream);
    const TnWs w = tn_carve(ws, B, T, E, L, 1, V, n_users, n_items);
    float *rp[2], *rm[2], *rv[2];
    for (int t = 0; t < 2; ++t) {
        rp[t] = reinterpret_cast<float *>(rows_p[t]); rm[t] = reinterpret_cast<float *>(rows_m[t]);
        rv[t] = reinterpret_cast<float *>(rows_v[t]);
        R4R_REQUIRE(rp[t] && rm[t] && rv[t], "transnet_rows_apply: ID-vector table %d: null pointer", t);
    }
    // the gathered entries' sweep on the schedule (the same period / base / all on every rank)
    tn_tag_rows_kernel<<<(unsigned)cdiv(B_all, 256), 256, 0, st>>>(uid_all, iid_all, B_all, w.tag[0], w.tag[1], w.ctag[0],
                                                                 w.ctag[1], (int)adam_step);
    const AdamScalars sc = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    const bool aligned = ((rows_p[0] | rows_p[1] | rows_m[0] | rows_m[1] | rows_v[0] | rows_v[1]) & 15) == 0;
    MfTimeBlock tb{};
    tb.rlast_u = w.rlast[0]; tb.rlast_i = w.rlast[1]; tb.err = w.tb_err; tb.base = (int)sweep_base;
    tb.period = sweep_period; tb.flush = (sweep_all || sweep_period == 1) ? 1 : 0; tb.inc = 1;
    mf_time_block_scalars(tb, lr, beta1, beta2, eps, weight_decay, adam_step);
    return mf_table_rows_launch(rp[0], rm[0], rv[0], rp[1], rm[1], rv[1], n_users, n_items, TN_ID, uid_all, iid_all, gu_all,
                                gi_all, w.tag[0], w.tag[1], w.ctag[0], w.ctag[1], B_all, (int)adam_step, sc, st,
                                aligned ? &tb : nullptr);
}

// ---- the same update with ONE exchange and no glue launches: every rank packs its compact entries into one block
// (r4r_transnet_dp_block: ids as int32 + the two gradient rows, rows_device.h mf_block's layout at D = 5), ONE
// all_gather moves the blocks, and r4r_transnet_rows_apply_blocks runs the scheduled sweep straight over them -- its
// workgroups find the rows the step names by scanning the ids (mf_engine.hip's SCAN form), nobody tags them first.
namespace r4r {
__global__ __launch_bounds__(256) void tn_dp_block_kernel(const int64_t *uid, const int64_t *iid, const float *grow_u,
                                                          const float *grow_i, int *uid32, int *iid32, float *g, float *gu,
                                                          float *gi, int64_t B, int64_t B_pad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // element of the [B_pad, TN_ID] row arrays
    if (i < B_pad) {
        const bool real = i < B;
        uid32[i] = real ? (int)uid[i] : -1;
        iid32[i] = real ? (int)iid[i] : -1;
        g[i] = 0.f;
    }
    if (i < B_pad * TN_ID) {
        const bool real = i < B * TN_ID;
        gu[i] = real ? grow_u[i] : 0.f;
        gi[i] = real ? grow_i[i] : 0.f;
    }
}
}  // namespace r4r

extern "C" size_t r4r_transnet_dp_block_bytes(int64_t B_pad) { return B_pad < 0 ? 0 : mf_block(B_pad, TN_ID).bytes; }

extern "C" int r4r_transnet_dp_block(const int64_t *uid, const int64_t *iid, void *ws, size_t ws_bytes, int64_t B, int T,
                                     int E, int L, int64_t V, int64_t n_users, int64_t n_items, void *block, int64_t B_pad,
                                     void *stream) {
    R4R_REQUIRE(block && B >= 0 && B_pad >= B, "transnet_dp_block: null block, or B_pad < B");
    R4R_REQUIRE(B == 0 || (uid && iid && ws), "transnet_dp_block: null ids / workspace");
    if (B_pad == 0) return R4R_OK;
    const float *gu = nullptr, *gi = nullptr;
    if (B > 0) {
        if (ws_bytes < r4r_transnet_ws_bytes(B, T, E, L, 1, V, n_users, n_items)) {
            set_error("transnet_dp_block: workspace %zu < %zu bytes", ws_bytes, r4r_transnet_ws_bytes(B, T, E, L, 1, V, n_users, n_items));
            return R4R_ERR_WORKSPACE;
        }
        const TnWs w = tn_carve(ws, B, T, E, L, 1, V, n_users, n_items);
        gu = w.grow[0]; gi = w.grow[1];
    }
    const MfBlock k = mf_block(B_pad, TN_ID);
    char *b = static_cast<char *>(block);
    tn_dp_block_kernel<<<(unsigned)cdiv(B_pad * TN_ID, 256), 256, 0, as_stream(stream)>>>(
        uid, iid, gu, gi, reinterpret_cast<int *>(b + k.uid), reinterpret_cast<int *>(b + k.iid), reinterpret_cast<float *>(b + k.g),
        reinterpret_cast<float *>(b + k.gu), reinterpret_cast<float *>(b + k.gi), B, B_pad);
    return check_launch("transnet_dp_block");
}

extern "C" int r4r_transnet_rows_apply_blocks(const void *blocks, int world, int64_t B_pad, int sweep_period, int64_t sweep_base,
                                              int sweep_all, const uint64_t *rows_p, const uint64_t *rows_m,
                                              const uint64_t *rows_v, int64_t n_users, int64_t n_items, void *ws, size_t ws_bytes,
                                              int64_t B, int T, int E, int L, int64_t V,
                                              float lr, double beta1, double beta2, float eps, float weight_decay,
                                              int64_t adam_step, void *stream) {
    R4R_REQUIRE(blocks && rows_p && rows_m && rows_v && ws, "transnet_rows_apply_blocks: null pointer");
    R4R_REQUIRE(world >= 1 && B_pad >= 0 && (int64_t)world * B_pad <= 2048,
                "transnet_rows_apply_blocks: %lld gathered ratings outside 0..2048", (long long)world * B_pad);
    R4R_REQUIRE(sweep_period >= 1 && sweep_period <= MF_TB_MAX, "transnet_rows_apply_blocks: sweep_period outside 1..%d", MF_TB_MAX);
    R4R_REQUIRE(adam_step >= 1 && adam_step < (1ll << 31) && sweep_base >= 0 && sweep_base < adam_step,
                "transnet_rows_apply_blocks: bad adam_step, or sweep_base outside 0..adam_step - 1");
    if (ws_bytes < r4r_transnet_ws_bytes(B, T, E, L, 1, V, n_users, n_items)) {
        set_error("transnet_rows_apply_blocks: workspace %zu < %zu bytes", ws_bytes, r4r_transnet_ws_bytes(B, T, E, L, 1, V, n_users, n_items));
        return R4R_ERR_WORKSPACE;
    }
    if ((int64_t)world * B_pad == 0) return R4R_OK;
    const TnWs w = tn_carve(ws, B, T, E, L, 1, V, n_users, n_items);
    float *rp[2], *rm[2], *rv[2];
    for (int t = 0; t < 2; ++t) {
        rp[t] = reinterpret_cast<float *>(rows_p[t]); rm[t] = reinterpret_cast<float *>(rows_m[t]);
        rv[t] = reinterpret_cast<float *>(rows_v[t]);
        R4R_REQUIRE(rp[t] && rm[t] && rv[t], "transnet_rows_apply_blocks: ID-vector table %d: null pointer", t);
    }
    const AdamScalars sc = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    const bool aligned = ((rows_p[0] | rows_p[1] | rows_m[0] | rows_m[1] | rows_v[0] | rows_v[1]) & 15) == 0;
    MfTimeBlock tb{};
    tb.rlast_u = w.rlast[0]; tb.rlast_i = w.rlast[1]; tb.err = w.tb_err; tb.base = (int)sweep_base;
    tb.period = sweep_period; tb.flush = (sweep_all || sweep_period == 1) ? 1 : 0; tb.inc = 1;
    mf_time_block_scalars(tb, lr, beta1, beta2, eps, weight_decay, adam_step);
    return mf_table_rows_blocks_launch(rp[0], rm[0], rv[0], rp[1], rm[1], rv[1], n_users, n_items, TN_ID, blocks, world, B_pad,
                                       w.ctag[0], w.ctag[1], (int)adam_step, sc, as_stream(stream), aligned ? &tb : nullptr);
}
