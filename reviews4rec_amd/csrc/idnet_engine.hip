// Fused native training step for the ID-only recommenders with dense layers: model_type 'MF'
// (MF.py:60-68: MLP (+) GMF -> factorisation machine) and the NeuMF family (NeuMF.py: GMF, MLP,
// NeuMF; trained in three stages by main.py:289-340).  'MF_dot' / 'bias_only' have their own, leaner
// step (mf_engine.hip); the pattern here is the review models' (narre_engine.hip):
//
//   1  idnet_head_kernel   one workgroup per rating: the ID-row gathers + Philox dropout, the elementwise
//                          (GMF) product, the projection MLP (Dropout -> Linear(2L, L) -> ReLU ->
//                          Linear(L, L)), the final layer (a Linear, or MF's TorchFM), the bias head,
//                          SE -- and the backward of all of it: the rating's contribution to every dense
//                          parameter gradient as ONE row of a [B, NP] matrix, its ID-table gradient rows
//                          compact ([B, L] per table), row tags for the sweeps
//   2  idnet_reduce_kernel column sums of that matrix in a fixed order = the dense gradient, Adam on
//                          every dense parameter the moment its sum is complete, running sum of SE
//   3+ the tagged Adam sweep of mf_engine.hip over the first user / item table pair AND the two bias vectors
//      in one launch (NeuMF's second pair in another): every row moves every step (weight decay, SURVEY
//      fact 4), the dense gradient of a table is never materialised
//
// 3 launches per step (4 for NeuMF) against ~40 dependent ones op by op.  Variants share one kernel:
//   variant      rows gathered            z (input of the final layer)            final
//   0 MF         A                         [mlp(A), uA * iA]        (MF.py:60-66)  TorchFM(2L, L)
//   1 GMF        A                         uA * iA                  (NeuMF.py:32)  Linear(L, 1)
//   2 MLP        A                         mlp(A)                   (NeuMF.py:67)  Linear(L, 1)
//   3 NeuMF      A (gmf_*), B (mlp_*)      [uA * iA, mlp(B)]        (NeuMF.py:134) Linear(2L, 1)
#include "adam_device.h"
#include "rows_device.h"
#include "textcnn.h"

namespace r4r {

enum { IDN_MF = 0, IDN_GMF, IDN_MLP, IDN_NEUMF };
// flat dense layout: projection / project .1 and .3, final (FM: V + lin, else Linear), global bias
enum { IP_P1W = 0, IP_P1B, IP_P3W, IP_P3B, IP_FV, IP_FW, IP_FB, IP_GB, IP_COUNT };
constexpr int IDN_MAX_L = 64;        // LDS arrays of the head kernel are sized by it (85 KB at 64: W1 [64][129], W3 [64][65], FV [128][65])

struct ILayout { int64_t off[IP_COUNT], size[IP_COUNT], total; };

__host__ __device__ inline bool idn_has_mlp(int v) { return v != IDN_GMF; }
__host__ __device__ inline bool idn_has_gmf(int v) { return v != IDN_MLP; }
__host__ __device__ inline int idn_zwidth(int v, int L) { return (v == IDN_MF || v == IDN_NEUMF) ? 2 * L : L; }
__host__ __device__ inline int idn_pairs(int v) { return v == IDN_NEUMF ? 2 : 1; }
// dropout draws per rating: the gathered rows (2L per pair), then the projection's input (2L)
__host__ __device__ inline int idn_draws(int v, int L) { return 2 * L * idn_pairs(v) + (idn_has_mlp(v) ? 2 * L : 0); }

static ILayout idn_layout(int variant, int L) {
    ILayout lay;
    const int nz = idn_zwidth(variant, L);
    const bool mlp = idn_has_mlp(variant);
    const int64_t sz[IP_COUNT] = {mlp ? (int64_t)L * 2 * L : 0, mlp ? L : 0, mlp ? (int64_t)L * L : 0, mlp ? L : 0,
                                  variant == IDN_MF ? (int64_t)nz * L : 0, nz, 1, 1};
    int64_t o = 0;
    for (int i = 0; i < IP_COUNT; ++i) {
        lay.off[i] = o;
        lay.size[i] = sz[i];
        o += (sz[i] + 31) & ~(int64_t)31;          // 128-byte aligned slots (whole cache lines for the GEMM's weight-row pieces); pad floats stay 0 forever
    }
    lay.total = o;
    return lay;
}

struct IdnHead {
    const float *flat_p;
    int off[IP_COUNT], size[IP_COUNT];
    const float *tab[2][2];            // [pair][side] ID tables [rows, L]
    const float *tabm[2][2], *tabv[2][2];   // their Adam moments (scheduled sweeps: pending updates are applied on the way)
    const int *rl[2][2];               // [pair][side] per row: the last step an entry wave updated it (NULL: nothing can be pending)
    MfTimeBlock tb; AdamScalars sc0;   // the schedule (rows_device.h): base, period, the steps' scalars
    const float *bias[2];
    const int64_t *id[2];              // uid, iid [B]
    const float *y;
    float *part;                       // [B, np]
    float *grow[2][2];                 // [pair][side] compact gradient rows [B, L]
    float *g;                          // [B] d mean(SE) / d pred
    int *tag[2];
    int *ctag[2];                      // per sweep chunk of a table: the last step that touched a row in it (NULL: not kept)
    float *mult;                       // [B, draws]
    float *pred, *se;
    int64_t B;
    int L, np, variant, training, want_grad, now;
    float p_drop, inv_denom;
    uint64_t seed, offset;
};

template <int ML>
__global__ __launch_bounds__(256) void idnet_head_kernel(IdnHead a) {
    __shared__ float W1[ML][2 * ML + 1], W3[ML][ML + 1], FV[2 * ML][ML + 1];
    __shared__ float row[2][2][ML], rm[2][2][ML];          // gathered rows after dropout, their multipliers
    __shared__ float cat[2 * ML], catm[2 * ML], hid[ML], mlp[ML], z[2 * ML], dz[2 * ML], dmlp[ML], dhid[ML], dcat[2 * ML];
    __shared__ float b1[ML], b3[ML], fw[2 * ML], sfm[ML], tfm[ML], misc[8];
    const int L = a.L, L2 = 2 * L, tid = threadIdx.x, v = a.variant;
    const int64_t b = blockIdx.x;
    const float *fp = a.flat_p;
    const bool has_mlp = idn_has_mlp(v), has_gmf = idn_has_gmf(v), fm = v == IDN_MF;
    const int npair = idn_pairs(v), nz = idn_zwidth(v, L), ND = idn_draws(v, L);
    const int mp = v == IDN_NEUMF ? 1 : 0;                  // the pair the projection reads
    const float keep = 1.f / (1.f - a.p_drop);
    const bool drop = a.training && a.p_drop > 0.f;
    auto draw = [&](int k) -> float {
        float m = 1.f;
        if (drop) {
            const uint32_t r = philox_first_word(a.offset + (uint64_t)(b * ND + k), a.seed);
            m = ((float)(r >> 8) * (1.0f / 16777216.0f) >= a.p_drop) ? keep : 0.f;
        }
        if (a.mult) a.mult[b * ND + k] = m;
        return m;
    };
    // ---- S0: weights -> LDS, the ID rows (+ dropout), biases
    const int64_t uid = a.id[0][b], iid = a.id[1][b];
    if (has_mlp) {
        for (int i = tid; i < L * L2; i += 256) { const int k = i / L2; W1[k][i - k * L2] = fp[a.off[IP_P1W] + i]; }
        for (int i = tid; i < L * L; i += 256) { const int k = i / L; W3[k][i - k * L] = fp[a.off[IP_P3W] + i]; }
        if (tid < L) { b1[tid] = fp[a.off[IP_P1B] + tid]; b3[tid] = fp[a.off[IP_P3B] + tid]; }
    }
    if (fm) for (int i = tid; i < nz * L; i += 256) { const int r = i / L; FV[r][i - r * L] = fp[a.off[IP_FV] + i]; }
    if (tid < nz) fw[tid] = fp[a.off[IP_FW] + tid];
    for (int i = tid; i < npair * L2; i += 256) {            // draw k = pair * 2L + side * L + l
        const int pr = i / L2, r = i - pr * L2, s = r >= L, l = r - s * L;
        const float m = draw(i);
        rm[pr][s][l] = m;
        const int64_t rid = s ? iid : uid, e = rid * L + l;
        float x = a.tab[pr][s][e];
        if (a.rl[pr][s]) {                                  // the element's pending gradient-zero updates (not written back)
            float mq = a.tabm[pr][s][e], vq = a.tabv[pr][s][e];
            const int cur = tb_current(a.tb, a.rl[pr][s], e, rid, a.now);
#pragma unroll
            for (int j = 0; j < MF_TB_MAX - 1; ++j) {       // steps now - 7 .. now - 1
                AdamScalars sc = a.sc0;
                sc.lr_over_bc1 = a.tb.lr_bc1[j];
                sc.inv_sqrt_bc2 = a.tb.isb2[j];
                if (a.now - (MF_TB_MAX - 1 - j) > cur) adam_elem_fast(x, 0.f, mq, vq, sc);
            }
        }
        row[pr][s][l] = x * m;
    }
    if (tid == 0) {
        misc[0] = fp[a.off[IP_FB]]; misc[1] = fp[a.off[IP_GB]];
        misc[2] = a.bias[0][uid]; misc[3] = a.bias[1][iid];
    }
    __syncthreads();
    // ---- S1: projection input = Dropout(cat[user, item]) (MF.py:26-27,61-62; NeuMF.py:51-52,64-66)
    if (has_mlp && tid < L2) {
        const int s = tid >= L, l = tid - s * L;
        const float m = draw(npair * L2 + tid);
        catm[tid] = m;
        cat[tid] = row[mp][s][l] * m;
    }
    __syncthreads();
    // ---- S2: Linear(2L, L) + ReLU
    if (has_mlp && tid < L) {
        float acc = 0.f;
        for (int j = 0; j < L2; ++j) acc = fmaf(cat[j], W1[tid][j], acc);
        acc += b1[tid];
        hid[tid] = acc > 0.f ? acc : 0.f;
    }
    __syncthreads();
    // ---- S3: Linear(L, L); z = what the final layer reads
    if (has_mlp && tid < L) {
        float acc = 0.f;
        for (int k = 0; k < L; ++k) acc = fmaf(hid[k], W3[tid][k], acc);
        mlp[tid] = acc + b3[tid];
    }
    __syncthreads();
    if (tid < nz) {
        float val;
        if (v == IDN_GMF) val = row[0][0][tid] * row[0][1][tid];
        else if (v == IDN_MLP) val = mlp[tid];
        else if (v == IDN_MF) val = tid < L ? mlp[tid] : row[0][0][tid - L] * row[0][1][tid - L];
        else val = tid < L ? row[0][0][tid] * row[0][1][tid] : mlp[tid - L];
        z[tid] = val;
    }
    __syncthreads();
    // ---- S4: final layer.  FM (common_pytorch_models.py:49-57): 0.5 (|zV|^2 - z^2 . V^2) + lin(z)
    if (fm && tid < L) {                                    // column k = tid of z V and of z^2 V^2
        float s = 0.f, t = 0.f;
        for (int i = 0; i < nz; ++i) s = fmaf(z[i], FV[i][tid], s);
        for (int i = 0; i < nz; ++i) t = fmaf(z[i] * z[i], FV[i][tid] * FV[i][tid], t);
        sfm[tid] = s;
        tfm[tid] = t;
    }
    __syncthreads();
    if (tid == 0) {
        float rating = 0.f;
        for (int i = 0; i < nz; ++i) rating = fmaf(z[i], fw[i], rating);
        rating += misc[0];
        if (fm) {
            float s1 = 0.f, s2 = 0.f;
            for (int k = 0; k < L; ++k) s1 = fmaf(sfm[k], sfm[k], s1);
            for (int k = 0; k < L; ++k) s2 += tfm[k];
            rating = 0.5f * (s1 - s2) + rating;
        }
        const float pred = ((misc[2] + misc[3]) + misc[1]) + rating;      // user_bias + item_bias + global_bias + rating
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
            a.tag[0][uid] = a.now;
            a.tag[1][iid] = a.now;
            if (a.ctag[0]) {                                // chunk tags of both table pairs (same ids, same width)
                const int64_t id2[2] = {uid, iid};
                for (int s = 0; s < 2; ++s) {
                    const int64_t e0 = id2[s] * a.L;
                    a.ctag[s][e0 / MF_CHUNK] = a.now; a.ctag[s][(e0 + a.L - 1) / MF_CHUNK] = a.now;
                }
            }
        }
    }
    if (!a.want_grad) return;                               // uniform
    __syncthreads();
    const float g = misc[4];
    float *prow = a.part + (size_t)b * a.np;
    // ---- B1: final layer -> d z
    if (tid < nz) {
        prow[a.off[IP_FW] + tid] = g * z[tid];
        float d = fw[tid];
        if (fm) {
            float acc = 0.f;
            for (int k = 0; k < L; ++k) acc += sfm[k] * FV[tid][k] - z[tid] * FV[tid][k] * FV[tid][k];
            d += acc;
        }
        dz[tid] = g * d;
    }
    if (fm) for (int i = tid; i < nz * L; i += 256) {
        const int r = i / L, k = i - r * L;
        prow[a.off[IP_FV] + i] = g * (z[r] * sfm[k] - z[r] * z[r] * FV[r][k]);
    }
    if (tid == 0) { prow[a.off[IP_FB]] = g; prow[a.off[IP_GB]] = g; }
    if (tid < IP_COUNT) {                                   // the alignment pads between slots: columns nobody owns
        const int end = tid + 1 < IP_COUNT ? a.off[tid + 1] : a.np;
        for (int c = a.off[tid] + a.size[tid]; c < end; ++c) prow[c] = 0.f;
    }
    __syncthreads();
    // ---- B2: split d z; Linear(L, L) backward
    if (has_mlp && tid < L) {
        const int at = v == IDN_NEUMF ? L + tid : tid;      // where mlp sits inside z
        const float d = dz[at];
        dmlp[tid] = d;
        prow[a.off[IP_P3B] + tid] = d;
    }
    __syncthreads();
    if (has_mlp) {
        for (int i = tid; i < L * L; i += 256) { const int k = i / L; prow[a.off[IP_P3W] + i] = dmlp[k] * hid[i - k * L]; }
        if (tid < L) {
            float acc = 0.f;
            for (int k = 0; k < L; ++k) acc = fmaf(dmlp[k], W3[k][tid], acc);
            const float d = hid[tid] > 0.f ? acc : 0.f;
            dhid[tid] = d;
            prow[a.off[IP_P1B] + tid] = d;
        }
    }
    __syncthreads();
    // ---- B3: Linear(2L, L) backward -> d cat -> (through the projection's dropout) d rows of pair mp
    if (has_mlp) {
        for (int i = tid; i < L * L2; i += 256) { const int k = i / L2; prow[a.off[IP_P1W] + i] = dhid[k] * cat[i - k * L2]; }
        if (tid < L2) {
            float acc = 0.f;
            for (int k = 0; k < L; ++k) acc = fmaf(dhid[k], W1[k][tid], acc);
            dcat[tid] = acc * catm[tid];
        }
    }
    __syncthreads();
    // ---- B4: compact ID-table gradient rows (through the row dropout): GMF product + projection paths
    for (int i = tid; i < npair * L2; i += 256) {
        const int pr = i / L2, r = i - pr * L2, s = r >= L, l = r - s * L;
        float d = 0.f;
        if (has_gmf && pr == 0) {
            const int at = v == IDN_MF ? L + l : l;         // where the product sits inside z
            d = dz[at] * row[0][1 - s][l];
        }
        if (has_mlp && pr == mp) d += dcat[r];
        a.grow[pr][s][(size_t)b * L + l] = d * rm[pr][s][l];
    }
}

// ---- 2: dense gradient = column sums of part [B, np] (fixed order), Adam, running SE
struct IdnReduce {
    const float *part, *se;
    float *flat_g, *flat_p, *flat_m, *flat_v, *sse_accum;
    int64_t B;
    int np, apply;
    AdamScalars s;
};
constexpr int IR_ROWS = 16, IR_COLS = 16;
__global__ __launch_bounds__(IR_ROWS * IR_COLS) void idnet_reduce_kernel(IdnReduce c) {
    __shared__ float red[IR_ROWS][IR_COLS];
    const int ox = threadIdx.x & (IR_COLS - 1), rg = threadIdx.x / IR_COLS;
    const int col = blockIdx.x * IR_COLS + ox;              // column np = the SE accumulator
    float s = 0.f;
    if (col < c.np) for (int64_t b = rg; b < c.B; b += IR_ROWS) s += c.part[(size_t)b * c.np + col];
    else if (col == c.np) for (int64_t b = rg; b < c.B; b += IR_ROWS) s += c.se[b];
    red[rg][ox] = s;
    __syncthreads();
    if (rg == 0 && col <= c.np) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < IR_ROWS; ++r) t += red[r][ox];
        if (col == c.np) { if (c.sse_accum) c.sse_accum[0] += t; return; }
        c.flat_g[col] = t;
        if (c.apply) {
            float P = c.flat_p[col], M = c.flat_m[col], V = c.flat_v[col];
            adam_elem(P, t, M, V, c.s);
            c.flat_p[col] = P; c.flat_m[col] = M; c.flat_v[col] = V;
        }
    }
}

struct IdnWs {
    int *tag[2];
    int *ctag[2], *rlast[2][2], *tb_err;               // the scheduled sweeps' state (rows_device.h); rlast per [pair][side]
    float *part, *g, *mult, *grow[2][2];
    size_t bytes, persist;
};
static IdnWs idn_carve(void *ws, int variant, int64_t B, int L, int64_t n_users, int64_t n_items) {
    IdnWs w;
    char *p = static_cast<char *>(ws);
    size_t o = 0;
    auto take = [&](size_t nbytes) { char *r = p ? p + o : nullptr; o += align256(nbytes); return r; };
    w.tag[0] = reinterpret_cast<int *>(take((size_t)n_users * 4));          // persistent state first (zeroed once)
    w.tag[1] = reinterpret_cast<int *>(take((size_t)n_items * 4));
    for (int s = 0; s < 2; ++s) {
        const size_t chunks = (size_t)cdiv((s ? n_items : n_users) * (int64_t)L, MF_CHUNK);
        w.ctag[s] = reinterpret_cast<int *>(take(chunks * 4));
        w.rlast[0][s] = reinterpret_cast<int *>(take((size_t)(s ? n_items : n_users) * 4));
        w.rlast[1][s] = reinterpret_cast<int *>(take((size_t)(s ? n_items : n_users) * 4));
    }
    w.tb_err = reinterpret_cast<int *>(take(4));
    w.persist = o;
    const ILayout lay = idn_layout(variant, L);
    w.part = reinterpret_cast<float *>(take((size_t)B * lay.total * 4));
    w.g = reinterpret_cast<float *>(take((size_t)B * 4));
    w.mult = reinterpret_cast<float *>(take((size_t)B * idn_draws(variant, L) * 4));
    for (int pr = 0; pr < 2; ++pr)
        for (int s = 0; s < 2; ++s) w.grow[pr][s] = reinterpret_cast<float *>(take((size_t)B * L * 4));
    w.bytes = o;
    return w;
}

}  // namespace r4r

using namespace r4r;

extern "C" int r4r_idnet_nparam(void) { return IP_COUNT; }

extern "C" int r4r_idnet_layout(int variant, int L, int64_t *offsets, int64_t *sizes, int64_t *total) {
    R4R_REQUIRE(offsets && sizes && total, "idnet_layout: null pointer");
    R4R_REQUIRE(variant >= 0 && variant <= IDN_NEUMF && L > 0 && L <= IDN_MAX_L, "idnet_layout: variant %d, latent_size %d "
                "(0..3, 1..%d)", variant, L, IDN_MAX_L);
    const ILayout lay = idn_layout(variant, L);
    for (int i = 0; i < IP_COUNT; ++i) { offsets[i] = lay.off[i]; sizes[i] = lay.size[i]; }
    *total = lay.total;
    return R4R_OK;
}

extern "C" size_t r4r_idnet_ws_bytes(int variant, int64_t B, int L, int64_t n_users, int64_t n_items) {
    if (variant < 0 || variant > IDN_NEUMF || B < 0 || L <= 0 || L > IDN_MAX_L || n_users <= 0 || n_items <= 0) return 0;
    return idn_carve(nullptr, variant, B, L, n_users, n_items).bytes;
}

// which: 0 dropout multipliers [B, draws]; 1 d loss / d pred [B]; 2 the persistent head's size; 3 the int the
// temporally blocked sweeps set if more updates were ever pending than a visit applies (a broken schedule);
// 4 + 2 * pair + side: compact gradient rows [B, L] of that ID table
extern "C" size_t r4r_idnet_ws_offset(int variant, int64_t B, int L, int64_t n_users, int64_t n_items, int which) {
    const IdnWs w = idn_carve(reinterpret_cast<void *>(256), variant, B, L, n_users, n_items);
    if (which == 2) return w.persist;
    if (which == 3) return (size_t)(reinterpret_cast<char *>(w.tb_err) - reinterpret_cast<char *>(256));
    const char *q = which == 0 ? reinterpret_cast<char *>(w.mult)
                               : which == 1 ? reinterpret_cast<char *>(w.g)
                                            : reinterpret_cast<char *>(w.grow[((which - 4) >> 1) & 1][(which - 4) & 1]);
    return (size_t)(q - reinterpret_cast<char *>(256));
}

extern "C" int r4r_idnet_step(int variant, const int64_t *uid, const int64_t *iid, const float *y,
                              float *flat_p, float *flat_g, float *flat_m, float *flat_v,
                              const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                              int64_t n_users, int64_t n_items,
                              float *pred, float *se, float *sse_accum, void *ws, size_t ws_bytes,
                              int64_t B, int L, float dropout_p, int training, uint64_t seed, uint64_t offset,
                              float inv_denom, int sweep_period, int64_t sweep_base, int sweep_all,
                              float lr, double beta1, double beta2, float eps, float weight_decay,
                              int64_t adam_step, void *stream) {
    R4R_REQUIRE(uid && iid && flat_p && rows_p && pred && ws, "idnet_step: null pointer");
    R4R_REQUIRE(sweep_period >= 1 && sweep_period <= MF_TB_MAX, "idnet_step: sweep_period %d outside 1..%d", sweep_period, MF_TB_MAX);
    R4R_REQUIRE(sweep_base >= 0 && (!flat_g || sweep_base < adam_step), "idnet_step: sweep_base outside 0..adam_step - 1");
    R4R_REQUIRE(variant >= 0 && variant <= IDN_NEUMF, "idnet_step: variant %d outside 0..3", variant);
    R4R_REQUIRE(L > 0 && L <= IDN_MAX_L, "idnet_step: latent_size %d outside 1..%d", L, IDN_MAX_L);
    R4R_REQUIRE(n_users > 0 && n_items > 0 && B >= 0, "idnet_step: bad sizes");
    const bool train_step = flat_g != nullptr;
    // flat_m == NULL on a training step: gradients only (flat_g, the compact rows and d loss / d pred in the
    // workspace) -- the data-parallel form (r4r_adam_multi + r4r_idnet_rows_apply after the exchange)
    const bool apply = flat_m != nullptr;
    R4R_REQUIRE(!train_step || (y && se && adam_step >= 1 && (!apply || (flat_v && rows_m && rows_v))),
                "idnet_step: a training step needs ratings, se, gradient buffers and adam_step >= 1 (+ moments to update)");
    R4R_REQUIRE(!y || se, "idnet_step: se buffer required when y is given");
    R4R_REQUIRE(adam_step < (1ll << 31), "idnet_step: step tag overflow");
    R4R_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "idnet_step: dropout %f outside [0,1)", (double)dropout_p);
    R4R_REQUIRE(!train_step || B <= 32768, "idnet_step: batch %lld > 32768 (the table sweeps keep a side's ids in LDS; use the "
                "module path for larger batches)", (long long)B);
    if (ws_bytes < r4r_idnet_ws_bytes(variant, B, L, n_users, n_items)) {
        set_error("idnet_step: workspace %zu < %zu bytes", ws_bytes, r4r_idnet_ws_bytes(variant, B, L, n_users, n_items));
        return R4R_ERR_WORKSPACE;
    }
    if (B == 0) return R4R_OK;
    hipStream_t st = as_stream(stream);
    const ILayout lay = idn_layout(variant, L);
    R4R_REQUIRE(lay.total < (1ll << 31) && B * lay.total < (1ll << 31), "idnet_step: head-gradient matrix too large");
    const IdnWs w = idn_carve(ws, variant, B, L, n_users, n_items);
    const int npair = idn_pairs(variant);

    // rows_*: [pair A user table, pair A item table, pair B user table, pair B item table, user_bias, item_bias]
    IdnHead h;
    h.flat_p = flat_p;
    for (int i = 0; i < IP_COUNT; ++i) { h.off[i] = (int)lay.off[i]; h.size[i] = (int)lay.size[i]; }
    for (int pr = 0; pr < 2; ++pr)
        for (int s = 0; s < 2; ++s) {
            h.tab[pr][s] = reinterpret_cast<const float *>(rows_p[(pr < npair ? pr : 0) * 2 + s]);
            h.grow[pr][s] = w.grow[pr][s];
            R4R_REQUIRE(h.tab[pr][s], "idnet_step: null ID table");
        }
    for (int s = 0; s < 2; ++s) {
        h.bias[s] = reinterpret_cast<const float *>(rows_p[4 + s]);
        h.tag[s] = w.tag[s];
        R4R_REQUIRE(h.bias[s], "idnet_step: null bias vector");
    }
    h.id[0] = uid; h.id[1] = iid; h.y = y; h.part = w.part; h.g = w.g; h.mult = w.mult; h.pred = pred; h.se = se;
    // the scheduled sweeps (rows_device.h): 16-byte aligned tables, a training step that has the moments
    bool tb_on = train_step && rows_m && rows_v;
    for (int k = 0; k < 2 * npair && tb_on; ++k) tb_on = rows_m[k] && rows_v[k] && ((rows_p[k] | rows_m[k] | rows_v[k]) & 15) == 0;
    MfTimeBlock tb0{};
    if (tb_on) {
        tb0.err = w.tb_err; tb0.base = (int)sweep_base; tb0.period = sweep_period;
        tb0.flush = (sweep_all || sweep_period == 1) ? 1 : 0; tb0.inc = 1;
        mf_time_block_scalars(tb0, lr, beta1, beta2, eps, weight_decay, adam_step);
    }
    h.tb = tb0; h.sc0 = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step > 0 ? adam_step : 1, nullptr);
    for (int pr = 0; pr < 2; ++pr)
        for (int s = 0; s < 2; ++s) {
            const int k = (pr < npair ? pr : 0) * 2 + s;
            const bool cu = tb_on && sweep_period > 1;      // (rows the head reads catch up in registers)
            h.rl[pr][s] = cu ? w.rlast[pr < npair ? pr : 0][s] : nullptr;
            h.tabm[pr][s] = cu ? reinterpret_cast<const float *>(rows_m[k]) : nullptr;
            h.tabv[pr][s] = cu ? reinterpret_cast<const float *>(rows_v[k]) : nullptr;
        }
    for (int s = 0; s < 2; ++s) h.ctag[s] = (tb_on && apply) ? w.ctag[s] : nullptr;
    h.B = B; h.L = L; h.np = (int)lay.total; h.variant = variant; h.training = training; h.want_grad = train_step;
    h.now = (int)adam_step; h.p_drop = dropout_p; h.inv_denom = inv_denom; h.seed = seed; h.offset = offset;
    if (L <= 16) idnet_head_kernel<16><<<(unsigned)B, 256, 0, st>>>(h);
    else if (L <= 32) idnet_head_kernel<32><<<(unsigned)B, 256, 0, st>>>(h);
    else idnet_head_kernel<IDN_MAX_L><<<(unsigned)B, 256, 0, st>>>(h);       // latent_size 33 .. 64 (hyper_params.py:63 has no bound)
    if (!train_step) {
        if (int rc = check_launch("idnet_step(forward)")) return rc;
        return R4R_OK;
    }
    const AdamScalars sc = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    IdnReduce r;
    r.part = w.part; r.se = se; r.flat_g = flat_g; r.flat_p = flat_p; r.flat_m = flat_m; r.flat_v = flat_v;
    r.sse_accum = sse_accum; r.B = B; r.np = (int)lay.total; r.apply = apply ? 1 : 0; r.s = sc;
    idnet_reduce_kernel<<<(unsigned)cdiv(lay.total + 1, IR_COLS), IR_ROWS * IR_COLS, 0, st>>>(r);
    if (!apply) return check_launch("idnet_step(gradients)");

    float *rp[6], *rm[6], *rv[6];
    for (int k = 0; k < 6; ++k) {
        const bool used = k >= 4 || k < 2 * npair;
        rp[k] = reinterpret_cast<float *>(rows_p[k]); rm[k] = reinterpret_cast<float *>(rows_m[k]);
        rv[k] = reinterpret_cast<float *>(rows_v[k]);
        R4R_REQUIRE(!used || (rp[k] && rm[k] && rv[k]), "idnet_step: table / bias %d: null parameter / moment pointer", k);
    }
    // the first table pair and the two bias vectors in one launch (an entry wave that owns a row updates the row
    // and its bias element); NeuMF's second pair in another
    MfTimeBlock tb[2];
    for (int pr = 0; pr < 2; ++pr) {
        tb[pr] = tb0;
        tb[pr].rlast_u = w.rlast[pr][0]; tb[pr].rlast_i = w.rlast[pr][1];
    }
    if (int rc = mf_table_bias_rows_launch(rp[0], rm[0], rv[0], rp[1], rm[1], rv[1], rp[4], rm[4], rv[4], rp[5], rm[5], rv[5],
                                           n_users, n_items, L, uid, iid, w.grow[0][0], w.grow[0][1], w.g, w.tag[0], w.tag[1],
                                           B, (int)adam_step, sc, st, tb_on ? w.ctag[0] : nullptr, tb_on ? w.ctag[1] : nullptr,
                                           tb_on ? &tb[0] : nullptr))
        return rc;
    if (npair == 2)
        return mf_table_rows_launch(rp[2], rm[2], rv[2], rp[3], rm[3], rv[3], n_users, n_items, L, uid, iid, w.grow[1][0],
                                    w.grow[1][1], w.tag[0], w.tag[1], tb_on ? w.ctag[0] : nullptr, tb_on ? w.ctag[1] : nullptr,
                                    B, (int)adam_step, sc, st, tb_on ? &tb[1] : nullptr);
    return R4R_OK;
}

// What the temporally blocked sweeps left pending (r4r_idnet_step with next_uid and sweep_period > 1): every chunk of
// the variant's table pair(s) takes its pending updates now.  adam_step = the LAST COMPLETED step.
extern "C" int r4r_idnet_rows_flush(int variant, const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                                    int64_t n_users, int64_t n_items, void *ws, size_t ws_bytes, int64_t B, int L,
                                    int sweep_period, int64_t sweep_base,
                                    float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                                    void *stream) {
    R4R_REQUIRE(rows_p && rows_m && rows_v && ws, "idnet_rows_flush: null pointer");
    R4R_REQUIRE(sweep_period >= 1 && sweep_period <= MF_TB_MAX && sweep_base >= 0 && sweep_base <= adam_step,
                "idnet_rows_flush: sweep_period outside 1..%d or sweep_base outside 0..adam_step", MF_TB_MAX);
    R4R_REQUIRE(variant >= 0 && variant <= IDN_NEUMF && L > 0 && L <= IDN_MAX_L && n_users > 0 && n_items > 0 && B >= 0,
                "idnet_rows_flush: bad arguments");
    R4R_REQUIRE(adam_step >= 0 && adam_step < (1ll << 31), "idnet_rows_flush: bad adam_step");
    if (ws_bytes < r4r_idnet_ws_bytes(variant, B, L, n_users, n_items)) {
        set_error("idnet_rows_flush: workspace %zu < %zu bytes", ws_bytes, r4r_idnet_ws_bytes(variant, B, L, n_users, n_items));
        return R4R_ERR_WORKSPACE;
    }
    if (adam_step == 0 || sweep_base == adam_step) return R4R_OK;
    const IdnWs w = idn_carve(ws, variant, B, L, n_users, n_items);
    const int npair = idn_pairs(variant);
    for (int k = 0; k < 2 * npair; ++k) {
        R4R_REQUIRE(rows_p[k] && rows_m[k] && rows_v[k], "idnet_rows_flush: table %d: null pointer", k);
        if ((rows_p[k] | rows_m[k] | rows_v[k]) & 15) return R4R_OK;       // (unaligned tables never defer)
    }
    const AdamScalars sc = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    auto f = [](uint64_t x) { return reinterpret_cast<float *>(x); };
    for (int pr = 0; pr < npair; ++pr) {
        MfTimeBlock tb{};
        tb.rlast_u = w.rlast[pr][0]; tb.rlast_i = w.rlast[pr][1]; tb.err = w.tb_err; tb.base = (int)sweep_base;
        tb.period = sweep_period; tb.flush = 1; tb.inc = 0;
        mf_time_block_scalars(tb, lr, beta1, beta2, eps, weight_decay, adam_step);
        if (int rc = mf_table_rows_launch(f(rows_p[2 * pr]), f(rows_m[2 * pr]), f(rows_v[2 * pr]), f(rows_p[2 * pr + 1]),
                                          f(rows_m[2 * pr + 1]), f(rows_v[2 * pr + 1]), n_users, n_items, L, nullptr, nullptr, nullptr,
                                          nullptr, w.tag[0], w.tag[1], w.ctag[0], w.ctag[1], 0, (int)adam_step, sc,
                                          as_stream(stream), &tb))
            return rc;
    }
    return R4R_OK;
}

// Data parallel: the ID-table / bias updates from ALL ranks' compact rows, gathered by the caller in rank order
// (ids -1 pad ragged shards), after a gradients-only r4r_idnet_step (flat_m == NULL) and the dense exchange
// (all-reduce of flat_g + r4r_adam_multi).  gu_all / gi_all: [pairs] device pointers to [B_all, L] rows (HOST
// arrays of 2 entries; the second pair only for NeuMF); g_all [B_all].  `ws` and B are the step's own: the row
// tags live in the workspace's persistent head.
namespace r4r {
__global__ void idn_tag_rows_kernel(const int64_t *uid, const int64_t *iid, int64_t n, int *tag_u, int *tag_i, int now,
                                    int L, int *ctag_u, int *ctag_i) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    if (uid[e] < 0) return;
    tag_u[uid[e]] = now;
    tag_i[iid[e]] = now;
    if (ctag_u) {
        const int64_t u0 = uid[e] * L, i0 = iid[e] * L;
        ctag_u[u0 / MF_CHUNK] = now; ctag_u[(u0 + L - 1) / MF_CHUNK] = now;
        ctag_i[i0 / MF_CHUNK] = now; ctag_i[(i0 + L - 1) / MF_CHUNK] = now;
    }
}
}  // namespace r4r

extern "C" int r4r_idnet_rows_apply(int variant, const int64_t *uid_all, const int64_t *iid_all, const float *g_all,
                                    const uint64_t *gu_all, const uint64_t *gi_all,
                                    int sweep_period, int64_t sweep_base, int sweep_all,
                                    int64_t B_all,
                                    const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                                    int64_t n_users, int64_t n_items, void *ws, size_t ws_bytes, int64_t B, int L,
                                    float lr, double beta1, double beta2, float eps, float weight_decay,
                                    int64_t adam_step, void *stream) {
    R4R_REQUIRE(uid_all && iid_all && g_all && gu_all && gi_all && rows_p && rows_m && rows_v && ws, "idnet_rows_apply: null pointer");
    R4R_REQUIRE(variant >= 0 && variant <= IDN_NEUMF && L > 0 && L <= IDN_MAX_L, "idnet_rows_apply: bad variant / latent_size");
    R4R_REQUIRE(B_all >= 0 && B_all <= 32768, "idnet_rows_apply: %lld gathered ratings outside 0..32768", (long long)B_all);
    R4R_REQUIRE(adam_step >= 1 && adam_step < (1ll << 31) && sweep_base >= 0 && sweep_base < adam_step,
                "idnet_rows_apply: bad adam_step, or sweep_base outside 0..adam_step - 1");
    R4R_REQUIRE(sweep_period >= 1 && sweep_period <= MF_TB_MAX, "idnet_rows_apply: sweep_period outside 1..%d", MF_TB_MAX);
    if (ws_bytes < r4r_idnet_ws_bytes(variant, B, L, n_users, n_items)) {
        set_error("idnet_rows_apply: workspace %zu < %zu bytes", ws_bytes, r4r_idnet_ws_bytes(variant, B, L, n_users, n_items));
        return R4R_ERR_WORKSPACE;
    }
    if (B_all == 0) return R4R_OK;
    hipStream_t st = as_stream(stream);
    const IdnWs w = idn_carve(ws, variant, B, L, n_users, n_items);
    const int npair = idn_pairs(variant);
    float *rp[6], *rm[6], *rv[6];
    for (int k = 0; k < 6; ++k) {
        const bool used = k >= 4 || k < 2 * npair;
        rp[k] = reinterpret_cast<float *>(rows_p[k]); rm[k] = reinterpret_cast<float *>(rows_m[k]);
        rv[k] = reinterpret_cast<float *>(rows_v[k]);
        R4R_REQUIRE(!used || (rp[k] && rm[k] && rv[k]), "idnet_rows_apply: table / bias %d: null pointer", k);
    }
    // the gathered entries' sweeps on the schedule (16-byte aligned tables; the same period / base / all on every rank)
    bool tb_on = true;
    for (int k = 0; k < 2 * npair; ++k) tb_on = tb_on && ((rows_p[k] | rows_m[k] | rows_v[k]) & 15) == 0;
    idn_tag_rows_kernel<<<(unsigned)cdiv(B_all, 256), 256, 0, st>>>(uid_all, iid_all, B_all, w.tag[0], w.tag[1], (int)adam_step, L,
                                                                  tb_on ? w.ctag[0] : nullptr, tb_on ? w.ctag[1] : nullptr);
    const AdamScalars sc = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    MfTimeBlock tb[2];
    for (int pr = 0; pr < 2; ++pr) {
        tb[pr] = MfTimeBlock{};
        tb[pr].rlast_u = w.rlast[pr][0]; tb[pr].rlast_i = w.rlast[pr][1]; tb[pr].err = w.tb_err; tb[pr].base = (int)sweep_base;
        tb[pr].period = sweep_period; tb[pr].flush = (sweep_all || sweep_period == 1) ? 1 : 0; tb[pr].inc = 1;
        mf_time_block_scalars(tb[pr], lr, beta1, beta2, eps, weight_decay, adam_step);
    }
    const float *gu[2], *gi[2];
    for (int pr = 0; pr < npair; ++pr) {
        gu[pr] = reinterpret_cast<const float *>(gu_all[pr]); gi[pr] = reinterpret_cast<const float *>(gi_all[pr]);
        R4R_REQUIRE(gu[pr] && gi[pr], "idnet_rows_apply: null gradient rows of pair %d", pr);
    }
    if (int rc = mf_table_bias_rows_launch(rp[0], rm[0], rv[0], rp[1], rm[1], rv[1], rp[4], rm[4], rv[4], rp[5], rm[5], rv[5],
                                           n_users, n_items, L, uid_all, iid_all, gu[0], gi[0], g_all, w.tag[0], w.tag[1],
                                           B_all, (int)adam_step, sc, st, tb_on ? w.ctag[0] : nullptr, tb_on ? w.ctag[1] : nullptr,
                                           tb_on ? &tb[0] : nullptr))
        return rc;
    if (npair == 2)
        return mf_table_rows_launch(rp[2], rm[2], rv[2], rp[3], rm[3], rv[3], n_users, n_items, L, uid_all, iid_all, gu[1],
                                    gi[1], w.tag[0], w.tag[1], tb_on ? w.ctag[0] : nullptr, tb_on ? w.ctag[1] : nullptr,
                                    B_all, (int)adam_step, sc, st, tb_on ? &tb[1] : nullptr);
    return R4R_OK;
}
