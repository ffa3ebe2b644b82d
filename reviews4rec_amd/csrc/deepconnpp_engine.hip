// Fused native training step for DeepCoNN++ (launch roles shared with the other review models:
// step_device.h).
#include "step_device.h"

namespace r4r {

// DeepCoNN++ (DeepCoNN.py:37-72 with model_type 'deepconn++'): the two TextCNN towers of DeepCoNN,
// then `final` = Linear(2L, L) -> ReLU -> Dropout -> Linear(L, 1) plus user / item / global bias
// instead of the FM.  Same launch structure as the NARRE step (narre_engine.hip) and the same backward /
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
        o += (sz[i] + 31) & ~(int64_t)31;
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

BWD_TRACE_DEFINE(r4r_debug_dcpp_bwd_trace)
// One workgroup of 256 threads per rating (the work is ~2000-element loops: FC gradients, d pooled).
#define SEL2(arr, s) ((s) ? (arr)[1] : (arr)[0])
template <int ML>
__global__ __launch_bounds__(256) void dcpp_head_kernel(DcppHead a) {
    __shared__ __attribute__((aligned(16))) float P[2][NF];
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
    // ---- S0: weights, pool finish (max over tiles, relu, first argmax), biases.  All global reads are
    // issued into registers before anything waits (a load -> LDS-store loop is one memory round
    // trip per iteration, the tile loop of the pool finish one per tile)
    // (the FC matrices as 16-byte units, NF / 4 = 25 filter quads per row: a quarter of the loads and index computations --
    // the head kernels are bound by vector-ALU issue)
    constexpr int NQ4 = NF / 4;
    typedef float hq4 __attribute__((ext_vector_type(4)));
    constexpr int WREG = (2 * ML * NQ4 + 255) / 256, AREG = (ML * 2 * ML + 255) / 256, PT = 8;
    hq4 wv[WREG];
    float av[AREG];
    const int wtot = 2 * L * NQ4;
    // What later reads depend on goes out FIRST (loads return in order: waiting for the oldest requests does not wait
    // for the weights behind them): the rating's ids -- their bias elements are a second round trip, now under the
    // first -- and this thread's pool partials, first PT tiles (as a loop behind the weights they were a third one).
    // Every load unconditional at a clamped address (behind a uniform `if` each got a branch and a full wait), the
    // towers' pointer arrays selected, not indexed (transnet_engine.hip).
    const int64_t uid_r = a.id[0][b], iid_r = a.id[1][b];
    const int ps = tid >= NF, pf = min(tid - ps * NF, NF - 1);
    float pv0[PT];
    int pp0[PT];
#pragma unroll
    for (int k = 0; k < PT; ++k) {
        const size_t q = ((size_t)b * a.tiles + min(k, a.tiles - 1)) * NP + pf;   // (a clamped duplicate never wins: strict >)
        pv0[k] = SEL2(a.pmax, ps)[q];
        pp0[k] = SEL2(a.parg, ps)[q];
    }
#pragma unroll
    for (int u = 0; u < WREG; ++u) {
        const int i = min(tid + 256 * u, wtot - 1), s = i >= L * NQ4;
        wv[u] = *reinterpret_cast<const hq4 *>(fp + (s ? a.off[DP_IFW] : a.off[DP_UFW]) + 4 * (i - s * L * NQ4));
    }
#pragma unroll
    for (int u = 0; u < AREG; ++u) av[u] = fp[a.off[DP_F0W] + min(tid + 256 * u, L * L2 - 1)];
    const int t2 = min(tid, L2 - 1), tl = min(tid, L - 1);
    const float fcb_r = fp[(t2 >= L ? a.off[DP_IFB] : a.off[DP_UFB]) + (t2 >= L ? t2 - L : t2)];
    const float b1_r = fp[a.off[DP_F0B] + tl], w3_r = fp[a.off[DP_F3W] + tl];
    const float m0 = fp[a.off[DP_F3B]], m1 = fp[a.off[DP_GB]];
    const float m2 = a.bias[0][uid_r], m3 = a.bias[1][iid_r];
    if (tid < 2 * NF) {
        const int s = ps, f = pf;
        float best = -INFINITY;
        int bp = -1;
#pragma unroll
        for (int k = 0; k < PT; ++k)
            if (pv0[k] > best) { best = pv0[k]; bp = pp0[k]; }
        for (int k0 = PT; k0 < a.tiles; k0 += PT) {          // documents of more than PT tiles: the rest, PT at a time
            float v[PT];
            int pp[PT];
#pragma unroll
            for (int k = 0; k < PT; ++k) {
                const bool in = k0 + k < a.tiles;
                const size_t q = ((size_t)b * a.tiles + (in ? k0 + k : 0)) * NP + f;
                v[k] = in ? SEL2(a.pmax, s)[q] : -INFINITY;
                pp[k] = SEL2(a.parg, s)[q];
            }
#pragma unroll
            for (int k = 0; k < PT; ++k)
                if (v[k] > best) { best = v[k]; bp = pp[k]; }
        }
        if (!(best > 0.f)) { best = 0.f; bp = -1; }
        P[s][f] = best;
        SEL2(a.pooled, s)[b * NF + f] = best;
        SEL2(a.argmax, s)[b * NF + f] = bp;
    }
#pragma unroll
    for (int u = 0; u < WREG; ++u) {
        const int i = tid + 256 * u;
        if (i < wtot) {
            const int s = i >= L * NQ4, r = i - s * L * NQ4, l = r / NQ4, f0 = 4 * (r - l * NQ4);
#pragma unroll
            for (int c = 0; c < 4; ++c) fcw[s][l][f0 + c] = wv[u][c];           // (rows of NF + 1 floats: four 4-byte writes)
        }
    }
#pragma unroll
    for (int u = 0; u < AREG; ++u) {
        const int i = tid + 256 * u;
        if (i < L * L2) { const int k = qd(i, invL2); W1[k][i - k * L2] = av[u]; }
    }
    if (tid < L2) fcbs[tid] = fcb_r;
    if (tid < L) { b1s[tid] = b1_r; w3s[tid] = w3_r; }
    if (tid == 0) { misc[0] = m0; misc[1] = m1; misc[2] = m2; misc[3] = m3; }
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
    for (int i = tid; i < 2 * L * NQ4; i += 256) {          // (filter quads: 16-byte reads of the pooled features, 16-byte stores)
        const int s = i >= L * NQ4, r = i - s * L * NQ4, l = r / NQ4, q = r - l * NQ4;
        const hq4 p4 = *reinterpret_cast<const hq4 *>(&P[s][4 * q]);
        *reinterpret_cast<hq4 *>(prow + col(a.off[s ? DP_IFW : DP_UFW] + l * NF + 4 * q)) = dzs[s * L + l] * p4;
    }
    if (tid < 2 * NF) {
        const int s = tid >= NF, f = tid - s * NF;
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc = fmaf(dzs[s * L + l], fcw[s][l][f], acc);
        SEL2(a.g_pooled, s)[b * NF + f] = acc;
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

}  // namespace r4r

using namespace r4r;

// ------------------------------------------------------------------------------ DeepCoNN++
extern "C" int r4r_deepconnpp_nparam(void) { return DP_COUNT; }

extern "C" int r4r_deepconnpp_layout(int E, int L, int64_t *offsets, int64_t *sizes, int64_t *total) {
    R4R_REQUIRE(offsets && sizes && total, "deepconnpp_layout: null pointer");
    R4R_REQUIRE(E > 0 && L > 0 && L <= HEAD_MAX_L, "deepconnpp_layout: bad sizes");
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
    R4R_REQUIRE(L > 0 && L <= HEAD_MAX_L, "deepconnpp_step: latent_size %d outside 1..%d", L, HEAD_MAX_L);
    R4R_REQUIRE(E > 0 && E % 4 == 0, "deepconnpp_step: word_embed_size %d must be a positive multiple of 4", E);
    const bool train_step = flat_g != nullptr;
    // flat_m == NULL on a training step: gradients only (flat_g, d loss / d pred in the workspace) -- the
    // data-parallel form (r4r_adam_multi + r4r_deepconnpp_rows_apply after the exchange)
    const bool apply = flat_m != nullptr;
    R4R_REQUIRE(!train_step || (y && se && adam_step >= 1 && (!apply || (flat_v && rows_m && rows_v))),
                "deepconnpp_step: a training step needs ratings, se, gradient buffers and adam_step >= 1 (+ moments to update)");
    R4R_REQUIRE(!y || se, "deepconnpp_step: se buffer required when y is given");
    R4R_REQUIRE(!next_user_idx == !next_item_idx, "deepconnpp_step: next_user_idx and next_item_idx go together");
    R4R_REQUIRE(!next_user_idx || train_step, "deepconnpp_step: the next batch's tokens ride on the backward launches");
    R4R_REQUIRE(token_buffer == 0 || token_buffer == 1, "deepconnpp_step: token_buffer must be 0 or 1");
    R4R_REQUIRE(adam_step < (1ll << 31), "deepconnpp_step: step tag overflow");
    R4R_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "deepconnpp_step: dropout %f outside [0,1)", (double)dropout_p);
    R4R_REQUIRE(B * (int64_t)((T + 2 + 127) / 128) < (1ll << 31), "deepconnpp_step: grid too large");
    R4R_REQUIRE(!train_step || B <= 32768, "deepconnpp_step: batch %lld > 32768 (the bias sweep keeps a side's ids in LDS; "
                "use the module path for larger batches)", (long long)B);
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
            pt[t].ptab = w.ptab[t]; pt[t].pmax = w.pmax[t]; pt[t].parg = w.parg[t]; pt[t].wimg = w.wp[t];
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
    else if (L <= 32) dcpp_head_kernel<32><<<(unsigned)B, 256, 0, st>>>(h);
    else dcpp_head_kernel<64><<<(unsigned)B, 256, 0, st>>>(h);          // latent_size 33 .. 64 (hyper_params.py:63 has no bound)
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
    wa.table_bytes = (int64_t)V * E * 4;                   // (the wide wgrad reads the rows through a buffer resource)
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
    const int gx = packed ? (NF + 3) / 4 : NF;
    const dim3 bgrid(gx, wa.nsplit, 2 + backward_cs_slices(cs_blocks, gx * wa.nsplit) + (prefetch ? 1 : 0));
    if (packed) narre_backward_kernel<0, false><<<bgrid, WG_THREADS, 0, st>>>(BackwardArgs{wa, cs, cs_blocks, nx, 1, RowSweep{}, 0, 2});
    else narre_backward_kernel<0, true><<<bgrid, WG_THREADS, 0, st>>>(BackwardArgs{wa, cs, cs_blocks, nx, 0, RowSweep{}, 0, 2});

    const int red_blocks = (NF * 3 * E + NF + NRED_THREADS - 1) / NRED_THREADS;
    const int comp_blocks = prefetch ? (int)cdiv((V + 3) / 4, NRED_THREADS * compact_groups(V)) : 0;
    DenseAdam opt;
    opt.on = apply ? 1 : 0; opt.p = flat_p; opt.m = flat_m; opt.v = flat_v; opt.g = flat_g;
    opt.lo0 = lo0; opt.hi0 = hi0; opt.lo1 = lo1; opt.hi1 = hi1;
    opt.s = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    const int64_t longest = hi0 - lo0 > hi1 - lo1 ? hi0 - lo0 : hi1 - lo1;
    narre_reduce_kernel<<<dim3(red_blocks + comp_blocks + (apply ? (int)cdiv(longest, NRED_THREADS) : 0), 2), NRED_THREADS, 0,
                          st>>>(wa, red_blocks, comp_blocks, nx, opt);
    if (!apply) return check_launch("deepconnpp_step");

    float *rp[2], *rm[2], *rv[2];
    for (int k = 0; k < 2; ++k) {
        rp[k] = reinterpret_cast<float *>(rows_p[k]); rm[k] = reinterpret_cast<float *>(rows_m[k]);
        rv[k] = reinterpret_cast<float *>(rows_v[k]);
        R4R_REQUIRE(rp[k] && rm[k] && rv[k], "deepconnpp_step: bias vector %d: null parameter / moment pointer", k);
    }
    return mf_bias_rows_launch(rp[0], rm[0], rv[0], rp[1], rm[1], rv[1], n_users, n_items, uid, iid, w.g, w.tag[0], w.tag[1], B,
                               (int)adam_step, opt.s, st);
}

// Data parallel: the ID bias update from ALL ranks' (uid, iid, d loss / d pred), gathered by the
// caller in rank order (ids -1 pad ragged shards), after a gradients-only r4r_deepconnpp_step
// (flat_m == NULL).  `ws` and the shape arguments are the step's: the row tags live there.
namespace r4r {
__global__ void dcpp_tag_rows_kernel(const int64_t *uid, const int64_t *iid, int64_t n, int *tag_u, int *tag_i, int now) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n || uid[e] < 0) return;
    tag_u[uid[e]] = now;
    tag_i[iid[e]] = now;
}
}  // namespace r4r

extern "C" int r4r_deepconnpp_rows_apply(const int64_t *uid_all, const int64_t *iid_all, const float *g_all, int64_t B_all,
                                         const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                                         int64_t n_users, int64_t n_items, void *ws, size_t ws_bytes,
                                         int64_t B, int T, int E, int L, int64_t V,
                                         float lr, double beta1, double beta2, float eps, float weight_decay,
                                         int64_t adam_step, void *stream) {
    R4R_REQUIRE(uid_all && iid_all && g_all && rows_p && rows_m && rows_v && ws, "deepconnpp_rows_apply: null pointer");
    R4R_REQUIRE(B_all >= 0 && B_all <= 32768, "deepconnpp_rows_apply: %lld gathered ratings outside 0..32768", (long long)B_all);
    R4R_REQUIRE(adam_step >= 1 && adam_step < (1ll << 31), "deepconnpp_rows_apply: bad adam_step");
    if (ws_bytes < r4r_deepconnpp_ws_bytes(B, T, E, L, V, n_users, n_items)) {
        set_error("deepconnpp_rows_apply: workspace %zu < %zu bytes", ws_bytes, r4r_deepconnpp_ws_bytes(B, T, E, L, V, n_users, n_items));
        return R4R_ERR_WORKSPACE;
    }
    if (B_all == 0) return R4R_OK;
    hipStream_t st = as_stream(stream);
    const DcppWs w = dcpp_carve(ws, B, T, E, L, V, n_users, n_items);
    float *rp[2], *rm[2], *rv[2];
    for (int k = 0; k < 2; ++k) {
        rp[k] = reinterpret_cast<float *>(rows_p[k]); rm[k] = reinterpret_cast<float *>(rows_m[k]);
        rv[k] = reinterpret_cast<float *>(rows_v[k]);
        R4R_REQUIRE(rp[k] && rm[k] && rv[k], "deepconnpp_rows_apply: bias vector %d: null pointer", k);
    }
    dcpp_tag_rows_kernel<<<(unsigned)cdiv(B_all, 256), 256, 0, st>>>(uid_all, iid_all, B_all, w.tag[0], w.tag[1], (int)adam_step);
    const AdamScalars sc = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    return mf_bias_rows_launch(rp[0], rm[0], rv[0], rp[1], rm[1], rv[1], n_users, n_items, uid_all, iid_all, g_all, w.tag[0],
                               w.tag[1], B_all, (int)adam_step, sc, st);
}
