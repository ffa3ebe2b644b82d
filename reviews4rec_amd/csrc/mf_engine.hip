// Fused native training step for the ID-only recommenders of pytorch_models/MF.py --
// model_type 'MF_dot' (MF.py:41-58: bias gathers, two ID-embedding gathers, dropout on each,
// row dot product) and 'bias_only' (MF.py:39-46) -- with the loss (loss.py:7-11), the backward
// pass and the dense Adam update (main.py:94-96,60) in TWO launches:
//
//   1. mf_fwd_bwd_kernel   one wave per rating: gathers, Philox dropout, dot, prediction, SE,
//                          and the rating's gradient rows kept COMPACT ([B, D] per table + the
//                          scalar d loss / d pred); marks the rows it touched with the step's tag
//   2. mf_adam_kernel      every parameter moves every step (L2 weight decay), but
//                          the dense table gradient is never materialised.  Two kinds of workgroup:
//                          SWEEP workgroups stream every element (24 B: read p, m, v, write p, m, v)
//                          and give the rows no rating touched (tag != this step) the gradient-zero
//                          update; ENTRY waves, one per rating and side, scan the batch's ids (LDS),
//                          and the first entry of a row sums that row's gradient rows -- four
//                          interleaved accumulators of entries taken in ascending order, combined in
//                          a fixed order: deterministic, no atomics -- and updates the table row and
//                          its bias element.
//
// The op-by-op module path needs ~25 launches and a zero-filled dense gradient per table for the
// same step (28 B/element + the fill); on Amazon-Electronics-sized tables (16.6 M parameters)
// the sweep is the whole cost, so this kernel is the HBM-bound leg of SURVEY.md 8d for MF.
#include "adam_device.h"
#include "common.h"

namespace r4r {

constexpr int MF_MAX_D = 256;          // latent size: <= 4 elements per lane of the rating's wave
constexpr int MF_SLOTS = 5;            // user table, item table, user bias, item bias, global bias
constexpr int MF_MAX_B = 16384;        // the entry waves keep a side's ids in LDS (4 B each)

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

constexpr int MF_CHUNK_BIAS = 1024;    // bias vectors: short workgroups, so they are not the tail

// Scalar fields only: an array member indexed by the workgroup's slot number (even through a
// chain of constant-index selects, which LLVM turns back into an indexed access) is copied to
// scratch by hipcc, and a kernel that owns scratch streamed at 3.7 instead of 5+ TB/s.
struct MfSweep {
    float *p0, *p1, *p2, *p3, *p4;
    float *m0, *m1, *m2, *m3, *m4;
    float *v0, *v1, *v2, *v3, *v4;
    int64_t n0, n1, n2, n3;            // elements of the tables and bias vectors
    int cb1, cb2, cb3, cb_global, cb_entries;   // first workgroup of slots 1..3, of the global-bias group, of the entry waves
    const int64_t *uid, *iid;
    const float *gu, *gi, *g, *se;
    float *sse_accum;
    const int *tag_u, *tag_i;
    int64_t B;
    int D, now;
    AdamScalars s;
};

__host__ __device__ inline int mf_chunk(int t) { return t < 2 ? MF_CHUNK : MF_CHUNK_BIAS; }

__global__ __launch_bounds__(MF_THREADS) void mf_adam_kernel(MfSweep w) {
    extern __shared__ int sid[];                            // entry waves: the side's ids
    __shared__ float red[MF_THREADS];
    const int bx = (int)blockIdx.x, tid = threadIdx.x;
    if (bx >= w.cb_entries) {
        // ---- entry waves: 4 per workgroup, all of one side (user side's groups first)
        const int lane = tid & 63;
        const int groups = (int)((w.B + 3) / 4);
        int gi = bx - w.cb_entries;
        const int t = gi >= groups;
        if (t) gi -= groups;
        const int64_t *ids = t ? w.iid : w.uid;
        for (int64_t j = tid; j < w.B; j += MF_THREADS) sid[j] = (int)ids[j];
        __syncthreads();
        const int64_t k = (int64_t)gi * 4 + (tid >> 6);
        if (k >= w.B) return;                               // whole wave
        const int row = sid[k];
        const int nch = (int)((w.B + 63) / 64), kc = (int)(k / 64);
        // is k the first entry of its row?  (chunks up to k's own)
        for (int c = 0; c <= kc; ++c) {
            const int j = c * 64 + lane;
            const unsigned long long mask = __ballot(j < w.B && sid[j] == row);
            if (mask && (int64_t)c * 64 + (__ffsll((long long)mask) - 1) < k) return;   // an earlier entry owns the row
        }
        // owner: the row's entries in ascending order, dealt round-robin to four accumulators whose
        // loads are in flight together; lanes are columns (lane, lane + 64, ...)
        const int D = w.D;
        const float *rows = t ? w.gi : w.gu;
        float acc[4][MF_MAX_D / 64], gsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int x = 0; x < MF_MAX_D / 64; ++x) acc[q][x] = 0.f;
        // (entries are popped four at a time into named slots: a runtime-indexed pending list
        // would live in scratch, and a kernel that owns scratch streams slower -- see MfSweep)
        int carry0 = -1, carry1 = -1, carry2 = -1;          // < 4 entries left over from the previous chunk
        auto add4 = [&](int e0, int e1, int e2, int e3) {   // e_q < 0: slot q unused
            float tmp[4][MF_MAX_D / 64], tg[4];
            const int es[4] = {e0, e1, e2, e3};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = es[q] < 0 ? 0 : es[q];
#pragma unroll
                for (int x = 0; x < MF_MAX_D / 64; ++x)
                    tmp[q][x] = (es[q] >= 0 && lane + 64 * x < D) ? rows[(int64_t)e * D + lane + 64 * x] : 0.f;
                tg[q] = es[q] >= 0 ? w.g[e] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int x = 0; x < MF_MAX_D / 64; ++x) acc[q][x] += tmp[q][x];
                gsum[q] += tg[q];
            }
        };
        auto pop = [](unsigned long long &mask, int base) {
            if (!mask) return -1;
            const int e = base + (__ffsll((long long)mask) - 1);
            mask &= mask - 1;
            return e;
        };
        for (int c = kc; c < nch; ++c) {
            const int j = c * 64 + lane;
            unsigned long long mask = __ballot(j < w.B && sid[j] == row);
            // complete the carried group first (keeps every entry's accumulator = its rank mod 4)
            if (carry0 >= 0 && mask) {
                const int n = carry2 >= 0 ? 3 : (carry1 >= 0 ? 2 : 1);
                const int a1 = n >= 2 ? carry1 : pop(mask, c * 64), a2 = n >= 3 ? carry2 : pop(mask, c * 64);
                const int a3 = pop(mask, c * 64);
                if (a3 >= 0) { add4(carry0, a1, a2, a3); carry0 = carry1 = carry2 = -1; }
                else { carry1 = a1; carry2 = a2; }          // still short of four
            }
            while (carry0 < 0 && mask) {
                const int e0 = pop(mask, c * 64), e1 = pop(mask, c * 64), e2 = pop(mask, c * 64), e3 = pop(mask, c * 64);
                if (e3 >= 0) add4(e0, e1, e2, e3);
                else { carry0 = e0; carry1 = e1; carry2 = e2; }
            }
        }
        if (carry0 >= 0) add4(carry0, carry1, carry2, -1);
        const float gb = (gsum[0] + gsum[1]) + (gsum[2] + gsum[3]);
        if (D > 0) {
            float *bp = t ? w.p1 : w.p0, *bm = t ? w.m1 : w.m0, *bv = t ? w.v1 : w.v0;
#pragma unroll
            for (int x = 0; x < MF_MAX_D / 64; ++x) {
                const int col = lane + 64 * x;
                if (col < D) {
                    const int64_t o = (int64_t)row * D + col;
                    const float G = (acc[0][x] + acc[1][x]) + (acc[2][x] + acc[3][x]);
                    float P = bp[o], M = bm[o], V = bv[o];
                    adam_elem(P, G, M, V, w.s);
                    bp[o] = P; bm[o] = M; bv[o] = V;
                }
            }
        }
        if (lane == 0) {                                    // the row's bias element
            float *bp = t ? w.p3 : w.p2, *bm = t ? w.m3 : w.m2, *bv = t ? w.v3 : w.v2;
            float P = bp[row], M = bm[row], V = bv[row];
            adam_elem(P, gb, M, V, w.s);
            bp[row] = P; bm[row] = M; bv[row] = V;
        }
        return;
    }
    if (bx >= w.cb_global) {
        // ---- global bias (gradient = sum of d loss / d pred over the batch) + the running SE:
        // strided per-thread sums, then a fixed tree -- deterministic
        float a = 0.f, e = 0.f;
        for (int64_t b = tid; b < w.B; b += MF_THREADS) { a += w.g[b]; e += w.se[b]; }
        red[tid] = a;
        __syncthreads();
        for (int off = MF_THREADS / 2; off > 0; off >>= 1) { if (tid < off) red[tid] += red[tid + off]; __syncthreads(); }
        const float gsum = red[0];
        __syncthreads();
        red[tid] = e;
        __syncthreads();
        for (int off = MF_THREADS / 2; off > 0; off >>= 1) { if (tid < off) red[tid] += red[tid + off]; __syncthreads(); }
        if (tid == 0) {
            if (w.p4) {                                     // (absent when the caller keeps its global bias elsewhere)
                float P = w.p4[0], M = w.m4[0], V = w.v4[0];
                adam_elem(P, gsum, M, V, w.s);
                w.p4[0] = P; w.m4[0] = M; w.v4[0] = V;
            }
            if (w.sse_accum) w.sse_accum[0] += red[0];
        }
        return;
    }
    // ---- sweep workgroups: slot 0 user table, 1 item table, 2 user bias, 3 item bias
    const int t = (bx >= w.cb1) + (bx >= w.cb2) + (bx >= w.cb3);
    float *bp = w.p0, *bm = w.m0, *bv = w.v0;
    int64_t numel = w.n0;
    int cb = 0;
    if (t == 1) { bp = w.p1; bm = w.m1; bv = w.v1; numel = w.n1; cb = w.cb1; }
    else if (t == 2) { bp = w.p2; bm = w.m2; bv = w.v2; numel = w.n2; cb = w.cb2; }
    else if (t == 3) { bp = w.p3; bm = w.m3; bv = w.v3; numel = w.n3; cb = w.cb3; }
    const int W = t < 2 ? w.D : 1;
    const int64_t start = (int64_t)(bx - cb) * mf_chunk(t);
    int64_t cnt = numel - start;
    if (cnt > mf_chunk(t)) cnt = mf_chunk(t);
    float *p = bp + start, *m = bm + start, *v = bv + start;
    const int *tag = (t == 0 || t == 2) ? w.tag_u : w.tag_i;
    const bool aligned = (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0);
    // (row, column) of a thread's element advance incrementally: a 64-bit division per element
    // would cost more than the 24 bytes the element moves
    const int64_t row_start = start / W;
    const int col_start = (int)(start - row_start * W);
    if (W % 4 == 0 && aligned) {                            // table rows: a float4 never straddles a row
        const int64_t nvec = cnt >> 2;
        const unsigned first = col_start + tid * 4u;
        int64_t row = row_start + first / (unsigned)W;
        int col = (int)(first % (unsigned)W);
        const int step_row = (MF_THREADS * 4) / W, step_col = (MF_THREADS * 4) % W;
        // two float4 per round, every load of the round (p, m, v and the row tags) issued before
        // the first use: one memory round trip per round, eight requests in flight per lane
        int64_t i = tid;
        for (; i + MF_THREADS < nvec; i += 2 * MF_THREADS) {
            int64_t row1 = row + step_row;
            int col1 = col + step_col;
            if (col1 >= W) { col1 -= W; ++row1; }
            const int64_t j = i + MF_THREADS;
            float4 P0 = reinterpret_cast<float4 *>(p)[i], P1 = reinterpret_cast<float4 *>(p)[j];
            float4 M0 = reinterpret_cast<float4 *>(m)[i], M1 = reinterpret_cast<float4 *>(m)[j];
            float4 V0 = reinterpret_cast<float4 *>(v)[i], V1 = reinterpret_cast<float4 *>(v)[j];
            const int t0 = tag[row], t1 = tag[row1];
            if (t0 != w.now) {                              // touched rows belong to their entry wave
                adam_elem(P0.x, 0.f, M0.x, V0.x, w.s); adam_elem(P0.y, 0.f, M0.y, V0.y, w.s);
                adam_elem(P0.z, 0.f, M0.z, V0.z, w.s); adam_elem(P0.w, 0.f, M0.w, V0.w, w.s);
                reinterpret_cast<float4 *>(p)[i] = P0; reinterpret_cast<float4 *>(m)[i] = M0;
                reinterpret_cast<float4 *>(v)[i] = V0;
            }
            if (t1 != w.now) {
                adam_elem(P1.x, 0.f, M1.x, V1.x, w.s); adam_elem(P1.y, 0.f, M1.y, V1.y, w.s);
                adam_elem(P1.z, 0.f, M1.z, V1.z, w.s); adam_elem(P1.w, 0.f, M1.w, V1.w, w.s);
                reinterpret_cast<float4 *>(p)[j] = P1; reinterpret_cast<float4 *>(m)[j] = M1;
                reinterpret_cast<float4 *>(v)[j] = V1;
            }
            row = row1 + step_row;
            col = col1 + step_col;
            if (col >= W) { col -= W; ++row; }
        }
        for (; i < nvec; i += MF_THREADS) {
            float4 P = reinterpret_cast<float4 *>(p)[i];
            float4 M = reinterpret_cast<float4 *>(m)[i];
            float4 V = reinterpret_cast<float4 *>(v)[i];
            if (tag[row] != w.now) {
                adam_elem(P.x, 0.f, M.x, V.x, w.s); adam_elem(P.y, 0.f, M.y, V.y, w.s);
                adam_elem(P.z, 0.f, M.z, V.z, w.s); adam_elem(P.w, 0.f, M.w, V.w, w.s);
                reinterpret_cast<float4 *>(p)[i] = P; reinterpret_cast<float4 *>(m)[i] = M;
                reinterpret_cast<float4 *>(v)[i] = V;
            }
            row += step_row;
            col += step_col;
            if (col >= W) { col -= W; ++row; }
        }
    } else {                                                // bias vectors (W = 1) and unaligned tables
        const unsigned first = col_start + tid;
        int64_t row = row_start + first / (unsigned)W;
        int col = (int)(first % (unsigned)W);
        const int step_row = MF_THREADS / W, step_col = MF_THREADS % W;
        for (int64_t i = tid; i < cnt; i += MF_THREADS) {
            float P = p[i], M = m[i], V = v[i];
            if (tag[row] != w.now) {
                adam_elem(P, 0.f, M, V, w.s);
                p[i] = P; m[i] = M; v[i] = V;
            }
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

// Adam on two ID bias vectors whose gradient is d loss / d pred of the ratings that name the id
// (DeepCoNN++'s user_bias / item_bias, DeepCoNN.py:69-71): the D = 0 form of the sweep above --
// untouched elements take the gradient-zero update, a touched element the fixed-order sum of its
// ratings.  `tag_*`: per-element step tags the caller's forward kernel set to `now`.
int mf_bias_rows_launch(float *ub, float *ub_m, float *ub_v, float *ib, float *ib_m, float *ib_v,
                        int64_t n_users, int64_t n_items, const int64_t *uid, const int64_t *iid, const float *g,
                        const int *tag_u, const int *tag_i, int64_t B, int now, const AdamScalars &sc, hipStream_t st) {
    if (B > MF_MAX_B) {
        set_error("bias rows: batch %lld > %d", (long long)B, MF_MAX_B);
        return R4R_ERR_ARG;
    }
    MfSweep sw{};
    sw.p2 = ub; sw.m2 = ub_m; sw.v2 = ub_v; sw.p3 = ib; sw.m3 = ib_m; sw.v3 = ib_v;
    sw.n0 = sw.n1 = 0; sw.n2 = n_users; sw.n3 = n_items;
    int64_t chunks = 0;
    sw.cb1 = sw.cb2 = 0;                                    // no tables: slots 0, 1 are empty
    chunks += cdiv(n_users, mf_chunk(2));
    sw.cb3 = (int)chunks;
    chunks += cdiv(n_items, mf_chunk(3));
    sw.cb_global = (int)chunks;                             // no global-bias workgroup either
    sw.cb_entries = (int)chunks;
    chunks += 2 * cdiv(B, 4);
    sw.uid = uid; sw.iid = iid; sw.g = g; sw.se = nullptr; sw.sse_accum = nullptr;
    sw.tag_u = tag_u; sw.tag_i = tag_i; sw.B = B; sw.D = 0; sw.now = now; sw.s = sc;
    mf_adam_kernel<<<(unsigned)chunks, MF_THREADS, (size_t)B * sizeof(int), st>>>(sw);
    return check_launch("bias rows");
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
    R4R_REQUIRE(!m || B <= MF_MAX_B, "mf_step: batch %lld > %d (the entry waves keep a side's ids in LDS; use the "
                                     "module path for larger batches)", (long long)B, MF_MAX_B);
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
    int64_t numel[4], begin[4], chunks = 0;
    for (int k = 0; k < 4; ++k) {                           // sweep workgroups: tables, bias vectors
        numel[k] = rows[k] * width[k];
        begin[k] = chunks;
        chunks += cdiv(numel[k], mf_chunk(k));
    }
    sw.n0 = numel[0]; sw.n1 = numel[1]; sw.n2 = numel[2]; sw.n3 = numel[3];
    sw.cb1 = (int)begin[1]; sw.cb2 = (int)begin[2]; sw.cb3 = (int)begin[3];
    sw.cb_global = (int)chunks;                             // one workgroup: global bias + running SE
    chunks += 1;
    sw.cb_entries = (int)chunks;                            // entry waves: 4 per workgroup, per side
    chunks += 2 * cdiv(B, 4);
    R4R_REQUIRE(chunks < (1ll << 31), "mf_step: too many workgroups");
    sw.uid = uid; sw.iid = iid; sw.gu = w.gu; sw.gi = w.gi; sw.g = w.g; sw.se = se; sw.sse_accum = sse_accum;
    sw.tag_u = w.tag_u; sw.tag_i = w.tag_i; sw.B = B; sw.D = D; sw.now = (int)adam_step;
    sw.s = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    {
        ScopedTiming tm(R4R_TIMING_ADAM, st);
        mf_adam_kernel<<<(unsigned)chunks, MF_THREADS, (size_t)B * sizeof(int), st>>>(sw);
    }
    return check_launch("mf_step");
}
