// Fused native training step for the ID-only recommenders of pytorch_models/MF.py --
// model_type 'MF_dot' (MF.py:41-58: bias gathers, two ID-embedding gathers, dropout on each,
// row dot product) and 'bias_only' (MF.py:39-46) -- with the loss (loss.py:7-11), the backward
// pass and the dense Adam update (main.py:94-96,60) in TWO launches:
//
//   1. mf_fwd_bwd_kernel   one wave per rating: gathers, Philox dropout, dot, prediction, SE,
//                          and the rating's gradient rows kept COMPACT ([B, D] per table + the
//                          scalar d loss / d pred); marks the rows it touched with the step's tag
//   2. mf_adam_kernel      one streaming pass over every parameter (every row moves every step:
//                          L2 weight decay).  The dense table gradient is never
//                          materialised: a row whose tag is not this step's has gradient zero
//                          (24 B/element: read p, m, v, write p, m, v), a tagged row sums its
//                          compact entries in ascending batch order (deterministic, no atomics).
//
// The op-by-op module path needs ~25 launches and a zero-filled dense gradient per table for the
// same step (28 B/element + the fill); on Amazon-Electronics-sized tables (16.6 M parameters)
// the sweep is the whole cost, so this kernel is the HBM-bound leg of SURVEY.md 8d for MF.
#include "adam_device.h"
#include "common.h"

namespace r4r {

constexpr int MF_MAX_D = 256;          // latent size: <= 4 elements per lane of the rating's wave
constexpr int MF_SLOTS = 5;            // user table, item table, user bias, item bias, global bias
constexpr int MF_MAX_B = 1024;         // a tagged row scans the batch for its entries

struct MfStep {
    const int64_t *uid, *iid;          // [B]
    const float *y;                    // [B] or NULL
    float *p[MF_SLOTS], *m[MF_SLOTS], *v[MF_SLOTS];
    int64_t rows[MF_SLOTS];            // U+1, I+1, U+1, I+1, 1
    int width[MF_SLOTS];               // D, D, 1, 1, 1
    // workspace
    float *gu, *gi;                    // [B, D] compact gradient rows
    float *g;                          // [B] d mean(SE) / d pred
    float *mult;                       // [B, 2D] dropout multipliers
    int *tag_u, *tag_i;                // [U+1], [I+1]: step tag of the last step that touched the row
    float *pred, *se, *sse_accum;
    int64_t B;
    int D, training, want_grad, tag;
    float p_drop, inv_denom;
    uint64_t seed, offset;
};

__global__ __launch_bounds__(256) void mf_fwd_bwd_kernel(MfStep a) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= a.B) return;                                   // whole wave
    const int D = a.D;
    const int64_t u = a.uid[b], i = a.iid[b];
    const float base = (a.p[2][u] + a.p[3][i]) + a.p[4][0];
    float xu[MF_MAX_D / 64], xi[MF_MAX_D / 64], mu[MF_MAX_D / 64], mi[MF_MAX_D / 64];
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < MF_MAX_D / 64; ++k) {
        const int d = lane + 64 * k;
        xu[k] = xi[k] = 0.f;
        mu[k] = mi[k] = 1.f;
        if (d < D) {
            xu[k] = a.p[0][u * D + d];
            xi[k] = a.p[1][i * D + d];
            if (a.training && a.p_drop > 0.f) {
                const float keep = 1.f / (1.f - a.p_drop);
                const uint32_t ru = philox_first_word(a.offset + (uint64_t)(b * 2 * D + d), a.seed);
                const uint32_t ri = philox_first_word(a.offset + (uint64_t)(b * 2 * D + D + d), a.seed);
                mu[k] = ((float)(ru >> 8) * (1.0f / 16777216.0f) >= a.p_drop) ? keep : 0.f;
                mi[k] = ((float)(ri >> 8) * (1.0f / 16777216.0f) >= a.p_drop) ? keep : 0.f;
            }
            xu[k] *= mu[k];
            xi[k] *= mi[k];
            part = fmaf(xu[k], xi[k], part);
            if (a.mult) { a.mult[b * 2 * D + d] = mu[k]; a.mult[b * 2 * D + D + d] = mi[k]; }
        }
    }
    const float pred = D > 0 ? base + wave_sum(part) : base;
    if (lane == 0) a.pred[b] = pred;
    if (!a.y) return;
    const float d = pred - a.y[b];
    if (lane == 0) a.se[b] = d * d;
    if (!a.want_grad) return;
    const float g = 2.f * d * a.inv_denom;
    if (lane == 0) {
        a.g[b] = g;
        a.tag_u[u] = a.tag;
        a.tag_i[i] = a.tag;
    }
#pragma unroll
    for (int k = 0; k < MF_MAX_D / 64; ++k) {
        const int dd = lane + 64 * k;
        if (dd < D) {
            a.gu[b * D + dd] = g * mu[k] * xi[k];          // d pred / d U[u, d] = mult_u * (dropped item value)
            a.gi[b * D + dd] = g * mi[k] * xu[k];
        }
    }
}

constexpr int MF_CHUNK = 8192, MF_THREADS = 256;

constexpr int MF_CHUNK_BIAS = 1024;    // bias vectors: one float4 per thread, so their workgroups are not the tail

// Scalar fields only: an array member indexed by the workgroup's slot number (even through a
// chain of constant-index selects, which LLVM turns back into an indexed access) is copied to
// scratch by hipcc, and a kernel that owns scratch streamed at 3.7 instead of 5+ TB/s.
struct MfSweep {
    float *p0, *p1, *p2, *p3, *p4;
    float *m0, *m1, *m2, *m3, *m4;
    float *v0, *v1, *v2, *v3, *v4;
    int64_t n0, n1, n2, n3, n4;        // elements per slot
    int cb1, cb2, cb3, cb4;            // first workgroup of slots 1..4 (slot 0 starts at 0)
    const int64_t *uid, *iid;
    const float *gu, *gi, *g, *se;
    float *sse_accum;
    const int *tag_u, *tag_i;
    int64_t B;
    int D, now;
    AdamScalars s;
};

__host__ __device__ inline int mf_chunk(int t) { return t < 2 ? MF_CHUNK : MF_CHUNK_BIAS; }

// What one workgroup of the sweep needs, picked out of the kernel arguments with unrolled
// compares and passed BY VALUE: indexing the argument struct with a runtime slot number (or
// handing helpers a reference to it) makes hipcc copy the whole struct to scratch, and a kernel
// that owns scratch streamed at 3.7 instead of 5+ TB/s.
struct MfSlot {
    float *p, *m, *v;                  // this workgroup's chunk
    const int *tag;                    // row tags of the slot's side (NULL: global bias)
    const int64_t *ids;                // uid / iid
    const float *grow;                 // compact gradient rows [B, D] (tables) or NULL (bias vectors: g)
    const float *g;                    // [B]
    int64_t B;
    int D, now;
};

// gradient of element (row, col): zero unless the row carries this step's tag
__device__ __forceinline__ float mf_grad(const MfSlot s, int64_t row, int col) {
    float acc = 0.f;
    if (!s.tag) {                                           // global bias: every rating contributes
        for (int64_t b = 0; b < s.B; ++b) acc += s.g[b];
        return acc;
    }
    if (s.tag[row] != s.now) return 0.f;
    for (int64_t b = 0; b < s.B; ++b)                       // ascending batch order: deterministic
        if (s.ids[b] == row) acc += s.grow ? s.grow[b * s.D + col] : s.g[b];
    return acc;
}

// four consecutive columns of a tagged table row, one scan of the batch
__device__ __forceinline__ float4 mf_grad4(const MfSlot s, int64_t row, int col) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t b = 0; b < s.B; ++b)
        if (s.ids[b] == row) {
            const float4 r = *reinterpret_cast<const float4 *>(s.grow + b * s.D + col);
            acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
        }
    return acc;
}

__global__ __launch_bounds__(MF_THREADS) void mf_adam_kernel(MfSweep w) {
    const int bx = (int)blockIdx.x;
    // the workgroup's slot: 0 user table, 1 item table, 2 user bias, 3 item bias, 4 global bias
    const int t = (bx >= w.cb1) + (bx >= w.cb2) + (bx >= w.cb3) + (bx >= w.cb4);
    float *bp = w.p0, *bm = w.m0, *bv = w.v0;
    int64_t numel = w.n0;
    int cb = 0;
    if (t == 1) { bp = w.p1; bm = w.m1; bv = w.v1; numel = w.n1; cb = w.cb1; }
    else if (t == 2) { bp = w.p2; bm = w.m2; bv = w.v2; numel = w.n2; cb = w.cb2; }
    else if (t == 3) { bp = w.p3; bm = w.m3; bv = w.v3; numel = w.n3; cb = w.cb3; }
    else if (t == 4) { bp = w.p4; bm = w.m4; bv = w.v4; numel = w.n4; cb = w.cb4; }
    const int W = t < 2 ? w.D : 1;
    const int64_t start = (int64_t)(bx - cb) * mf_chunk(t);
    int64_t cnt = numel - start;
    if (cnt > mf_chunk(t)) cnt = mf_chunk(t);
    float *p = bp + start, *m = bm + start, *v = bv + start;
    const bool user_side = (t == 0 || t == 2);
    MfSlot a;
    a.p = p; a.m = m; a.v = v;
    a.tag = t == 4 ? nullptr : (user_side ? w.tag_u : w.tag_i);
    a.ids = user_side ? w.uid : w.iid;
    a.grow = t == 0 ? w.gu : (t == 1 ? w.gi : nullptr);
    a.g = w.g; a.B = w.B; a.D = w.D; a.now = w.now;
    if (t == 4 && threadIdx.x == 0 && w.sse_accum) {        // the running metric (main.py:57), same launch
        float s = 0.f;
        for (int64_t b = 0; b < w.B; ++b) s += w.se[b];
        w.sse_accum[0] += s;
    }
    const bool vec = (W % 4 == 0) && (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) |
                                        reinterpret_cast<uintptr_t>(v)) & 15) == 0);
    // (row, column) of a thread's element advance incrementally: a 64-bit division per element
    // would cost more than the 24 bytes the element moves
    const int64_t row_start = start / W;
    const int col_start = (int)(start - row_start * W);
    if (vec) {                                              // a float4 never straddles a row
        const int64_t nvec = cnt >> 2;
        const int *tag = a.tag;
        const unsigned first = col_start + threadIdx.x * 4u;
        int64_t row = row_start + first / (unsigned)W;
        int col = (int)(first % (unsigned)W);
        const int step_row = (MF_THREADS * 4) / W, step_col = (MF_THREADS * 4) % W;
        // two float4 per round, every load of the round (p, m, v and the row tags) issued before
        // the first use: one memory round trip per round, eight requests in flight per lane
        int64_t i = threadIdx.x;
        for (; i + MF_THREADS < nvec; i += 2 * MF_THREADS) {
            int64_t row1 = row + step_row;
            int col1 = col + step_col;
            if (col1 >= W) { col1 -= W; ++row1; }
            const int64_t j = i + MF_THREADS;
            float4 P0 = reinterpret_cast<float4 *>(p)[i], P1 = reinterpret_cast<float4 *>(p)[j];
            float4 M0 = reinterpret_cast<float4 *>(m)[i], M1 = reinterpret_cast<float4 *>(m)[j];
            float4 V0 = reinterpret_cast<float4 *>(v)[i], V1 = reinterpret_cast<float4 *>(v)[j];
            const int t0 = tag[row], t1 = tag[row1];
            float4 G0 = make_float4(0.f, 0.f, 0.f, 0.f), G1 = G0;
            if (t0 == a.now) G0 = mf_grad4(a, row, col);
            if (t1 == a.now) G1 = mf_grad4(a, row1, col1);
            adam_elem(P0.x, G0.x, M0.x, V0.x, w.s); adam_elem(P0.y, G0.y, M0.y, V0.y, w.s);
            adam_elem(P0.z, G0.z, M0.z, V0.z, w.s); adam_elem(P0.w, G0.w, M0.w, V0.w, w.s);
            adam_elem(P1.x, G1.x, M1.x, V1.x, w.s); adam_elem(P1.y, G1.y, M1.y, V1.y, w.s);
            adam_elem(P1.z, G1.z, M1.z, V1.z, w.s); adam_elem(P1.w, G1.w, M1.w, V1.w, w.s);
            reinterpret_cast<float4 *>(p)[i] = P0; reinterpret_cast<float4 *>(p)[j] = P1;
            reinterpret_cast<float4 *>(m)[i] = M0; reinterpret_cast<float4 *>(m)[j] = M1;
            reinterpret_cast<float4 *>(v)[i] = V0; reinterpret_cast<float4 *>(v)[j] = V1;
            row = row1 + step_row;
            col = col1 + step_col;
            if (col >= W) { col -= W; ++row; }
        }
        for (; i < nvec; i += MF_THREADS) {
            float4 P = reinterpret_cast<float4 *>(p)[i];
            float4 M = reinterpret_cast<float4 *>(m)[i];
            float4 V = reinterpret_cast<float4 *>(v)[i];
            float4 G = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tag[row] == a.now) G = mf_grad4(a, row, col);
            adam_elem(P.x, G.x, M.x, V.x, w.s);
            adam_elem(P.y, G.y, M.y, V.y, w.s);
            adam_elem(P.z, G.z, M.z, V.z, w.s);
            adam_elem(P.w, G.w, M.w, V.w, w.s);
            reinterpret_cast<float4 *>(p)[i] = P;
            reinterpret_cast<float4 *>(m)[i] = M;
            reinterpret_cast<float4 *>(v)[i] = V;
            row += step_row;
            col += step_col;
            if (col >= W) { col -= W; ++row; }
        }
    } else if (W == 1 && (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) |
                             reinterpret_cast<uintptr_t>(v)) & 15) == 0) && t < 4) {
        // bias vectors: four rows per float4, their four tags in one int4 (tags are 256-B aligned,
        // start is a multiple of 4)
        const int *tag = a.tag;
        const int64_t *ids = a.ids;
        const int64_t nvec = cnt >> 2;
        for (int64_t i = threadIdx.x; i < nvec; i += MF_THREADS) {
            const int64_t r0 = start + i * 4;
            float4 P = reinterpret_cast<float4 *>(p)[i];
            float4 M = reinterpret_cast<float4 *>(m)[i];
            float4 V = reinterpret_cast<float4 *>(v)[i];
            const int4 T = *reinterpret_cast<const int4 *>(tag + r0);
            float G0 = 0.f, G1 = 0.f, G2 = 0.f, G3 = 0.f;     // (no runtime-indexed array: that would be scratch)
            if (T.x == a.now || T.y == a.now || T.z == a.now || T.w == a.now)
                for (int64_t b = 0; b < a.B; ++b) {           // ascending batch order: deterministic
                    const int64_t d = ids[b] - r0;
                    const float gb = a.g[b];
                    if (d == 0) G0 += gb;
                    if (d == 1) G1 += gb;
                    if (d == 2) G2 += gb;
                    if (d == 3) G3 += gb;
                }
            adam_elem(P.x, G0, M.x, V.x, w.s);
            adam_elem(P.y, G1, M.y, V.y, w.s);
            adam_elem(P.z, G2, M.z, V.z, w.s);
            adam_elem(P.w, G3, M.w, V.w, w.s);
            reinterpret_cast<float4 *>(p)[i] = P;
            reinterpret_cast<float4 *>(m)[i] = M;
            reinterpret_cast<float4 *>(v)[i] = V;
        }
        for (int64_t i = (nvec << 2) + threadIdx.x; i < cnt; i += MF_THREADS) {      // < 4 leftover rows
            float P = p[i], M = m[i], V = v[i];
            adam_elem(P, mf_grad(a, start + i, 0), M, V, w.s);
            p[i] = P; m[i] = M; v[i] = V;
        }
    } else {
        const unsigned first = col_start + threadIdx.x;
        int64_t row = row_start + first / (unsigned)W;
        int col = (int)(first % (unsigned)W);
        const int step_row = MF_THREADS / W, step_col = MF_THREADS % W;
        for (int64_t i = threadIdx.x; i < cnt; i += MF_THREADS) {
            float P = p[i], M = m[i], V = v[i];
            adam_elem(P, mf_grad(a, row, col), M, V, w.s);
            p[i] = P; m[i] = M; v[i] = V;
            row += step_row;
            col += step_col;
            if (col >= W) { col -= W; ++row; }
        }
    }
}

struct MfWs {
    float *gu, *gi, *g, *mult;
    int *tag_u, *tag_i;
    size_t bytes;
};

static size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }

static MfWs mf_carve(void *ws, int64_t B, int D, int64_t n_users, int64_t n_items) {
    char *p = static_cast<char *>(ws);
    size_t o = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + o : nullptr; o += a256(bytes); return r; };
    MfWs w;
    w.tag_u = reinterpret_cast<int *>(take((size_t)n_users * 4));   // tags first: they must persist (zeroed once)
    w.tag_i = reinterpret_cast<int *>(take((size_t)n_items * 4));
    w.gu = reinterpret_cast<float *>(take((size_t)B * D * 4));
    w.gi = reinterpret_cast<float *>(take((size_t)B * D * 4));
    w.g = reinterpret_cast<float *>(take((size_t)B * 4));
    w.mult = reinterpret_cast<float *>(take((size_t)B * 2 * D * 4));
    w.bytes = o;
    return w;
}

}  // namespace r4r

using namespace r4r;

extern "C" size_t r4r_mf_ws_bytes(int64_t B, int D, int64_t n_users, int64_t n_items) {
    if (B < 0 || D < 0 || n_users <= 0 || n_items <= 0) return 0;
    return mf_carve(nullptr, B, D, n_users, n_items).bytes;
}

extern "C" size_t r4r_mf_ws_mult_offset(int64_t B, int D, int64_t n_users, int64_t n_items) {
    const MfWs w = mf_carve(reinterpret_cast<void *>(256), B, D, n_users, n_items);
    return (size_t)(reinterpret_cast<char *>(w.mult) - reinterpret_cast<char *>(256));
}

extern "C" size_t r4r_mf_ws_grad_offset(int64_t B, int D, int64_t n_users, int64_t n_items, int which) {
    const MfWs w = mf_carve(reinterpret_cast<void *>(256), B, D, n_users, n_items);
    const char *q = which == 0 ? reinterpret_cast<char *>(w.gu) : which == 1 ? reinterpret_cast<char *>(w.gi)
                                                                              : reinterpret_cast<char *>(w.g);
    return (size_t)(q - reinterpret_cast<char *>(256));
}

extern "C" int r4r_mf_step(const int64_t *uid, const int64_t *iid, const float *y,
                           const uint64_t *p, const uint64_t *m, const uint64_t *v,
                           int64_t n_users, int64_t n_items, int D,
                           float *pred, float *se, float *sse_accum, void *ws, size_t ws_bytes, int64_t B,
                           float dropout_p, int training, uint64_t seed, uint64_t offset, float inv_denom,
                           float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                           void *stream) {
    R4R_REQUIRE(uid && iid && p && pred && ws, "mf_step: null pointer");
    R4R_REQUIRE(n_users > 0 && n_items > 0 && B >= 0, "mf_step: bad sizes");
    R4R_REQUIRE(D >= 0 && D <= MF_MAX_D, "mf_step: latent_size %d outside 0..%d", D, MF_MAX_D);
    R4R_REQUIRE(!m == !v, "mf_step: m and v go together");
    R4R_REQUIRE(!m || (y && se && adam_step >= 1), "mf_step: a training step needs ratings, the se buffer and "
                                                   "adam_step >= 1");
    R4R_REQUIRE(!y || se, "mf_step: se buffer required when y is given");
    R4R_REQUIRE(!m || B <= MF_MAX_B, "mf_step: batch %lld > %d (a touched row scans the batch for its entries; use "
                                     "the module path for larger batches)", (long long)B, MF_MAX_B);
    R4R_REQUIRE(adam_step < (1ll << 31), "mf_step: step tag overflow");
    R4R_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "mf_step: dropout %f outside [0,1)", (double)dropout_p);
    if (ws_bytes < r4r_mf_ws_bytes(B, D, n_users, n_items)) {
        set_error("mf_step: workspace %zu < %zu bytes", ws_bytes, r4r_mf_ws_bytes(B, D, n_users, n_items));
        return R4R_ERR_WORKSPACE;
    }
    for (int k = (D > 0 ? 0 : 2); k < MF_SLOTS; ++k) {
        R4R_REQUIRE(p[k] && (!m || (m[k] && v[k])), "mf_step: slot %d: null parameter / moment pointer", k);
    }
    if (B == 0) return R4R_OK;
    hipStream_t st = as_stream(stream);
    const MfWs w = mf_carve(ws, B, D, n_users, n_items);
    MfStep a;
    a.uid = uid; a.iid = iid; a.y = y;
    const int64_t rows[MF_SLOTS] = {n_users, n_items, n_users, n_items, 1};
    const int width[MF_SLOTS] = {D, D, 1, 1, 1};
    for (int k = 0; k < MF_SLOTS; ++k) {
        a.p[k] = reinterpret_cast<float *>(p[k]);
        a.m[k] = m ? reinterpret_cast<float *>(m[k]) : nullptr;
        a.v[k] = v ? reinterpret_cast<float *>(v[k]) : nullptr;
        a.rows[k] = rows[k]; a.width[k] = width[k];
    }
    a.gu = w.gu; a.gi = w.gi; a.g = w.g; a.mult = w.mult; a.tag_u = w.tag_u; a.tag_i = w.tag_i;
    a.pred = pred; a.se = se; a.sse_accum = sse_accum;
    a.B = B; a.D = D; a.training = training; a.want_grad = m != nullptr; a.tag = (int)adam_step;
    a.p_drop = dropout_p; a.inv_denom = inv_denom; a.seed = seed; a.offset = offset;
    mf_fwd_bwd_kernel<<<(unsigned)cdiv(B, 4), 256, 0, st>>>(a);
    if (!m) return check_launch("mf_step(forward)");
    MfSweep sw;
    sw.p0 = a.p[0]; sw.p1 = a.p[1]; sw.p2 = a.p[2]; sw.p3 = a.p[3]; sw.p4 = a.p[4];
    sw.m0 = a.m[0]; sw.m1 = a.m[1]; sw.m2 = a.m[2]; sw.m3 = a.m[3]; sw.m4 = a.m[4];
    sw.v0 = a.v[0]; sw.v1 = a.v[1]; sw.v2 = a.v[2]; sw.v3 = a.v[3]; sw.v4 = a.v[4];
    int64_t numel[MF_SLOTS], begin[MF_SLOTS + 1], chunks = 0;
    for (int k = 0; k < MF_SLOTS; ++k) {
        numel[k] = rows[k] * width[k];
        begin[k] = chunks;
        chunks += cdiv(numel[k], mf_chunk(k));
        R4R_REQUIRE(chunks < (1ll << 31), "mf_step: too many chunks");
    }
    sw.n0 = numel[0]; sw.n1 = numel[1]; sw.n2 = numel[2]; sw.n3 = numel[3]; sw.n4 = numel[4];
    sw.cb1 = (int)begin[1]; sw.cb2 = (int)begin[2]; sw.cb3 = (int)begin[3]; sw.cb4 = (int)begin[4];
    sw.uid = uid; sw.iid = iid; sw.gu = w.gu; sw.gi = w.gi; sw.g = w.g; sw.se = se; sw.sse_accum = sse_accum;
    sw.tag_u = w.tag_u; sw.tag_i = w.tag_i; sw.B = B; sw.D = D; sw.now = (int)adam_step;
    sw.s = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    {
        ScopedTiming tm(R4R_TIMING_ADAM, st);
        mf_adam_kernel<<<(unsigned)chunks, MF_THREADS, 0, st>>>(sw);
    }
    return check_launch("mf_step");
}
