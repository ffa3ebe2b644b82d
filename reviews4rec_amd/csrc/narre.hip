// NARRE review-level attention for gfx950 (fused scorer MLP + softmax over the
// R reviews + weighted sum), forward and backward.
//
// Reference behaviour restated: NARRE.attention, pytorch_models/NARRE.py:53-64,
// with the scorer of NARRE.py:24-36 (Linear(2L->L), ReLU, Dropout, Linear(L->1)).
// Padded (all-zero) reviews are NOT masked out of the softmax, like the reference.
//
// One workgroup per example, thread (l, r) owns hidden unit l of review r; the
// [R][L] hidden tile, the R scores and the R attention weights live in LDS.  The
// whole problem is R*L ~ 100 values per example: launch/HBM bound, no MFMA.
#include "common.h"

namespace r4r {

constexpr int MAXR = 32, MAXL = 32;

__global__ void narre_attn_fwd_kernel(const float *__restrict__ x, const float *__restrict__ other,
                                      const float *__restrict__ W0, const float *__restrict__ b0,
                                      const float *__restrict__ w3, const float *__restrict__ b3,
                                      const float *__restrict__ mult,
                                      float *__restrict__ out, float *__restrict__ h_save,
                                      float *__restrict__ a_save, int R, int L) {
    __shared__ float hs[MAXR][MAXL + 1];
    __shared__ float sc[MAXR];
    const int l = threadIdx.x, r = threadIdx.y;
    const int64_t n = blockIdx.x;
    const float *xr = x + (n * R + r) * L;
    const float *orow = other + (n * R + r) * L;
    const float *wrow = W0 + (size_t)l * 2 * L;
    float s = 0.f;
    for (int c = 0; c < L; ++c) s = fmaf(wrow[c], xr[c], s);
    for (int c = 0; c < L; ++c) s = fmaf(wrow[L + c], orow[c], s);
    s += b0[l];
    s = s > 0.f ? s : 0.f;
    if (mult) s *= mult[(n * R + r) * L + l];
    hs[r][l] = s;
    h_save[(n * R + r) * L + l] = s;
    __syncthreads();
    if (l == 0) {
        float t = 0.f;
        for (int k = 0; k < L; ++k) t = fmaf(w3[k], hs[r][k], t);
        sc[r] = t + b3[0];
    }
    __syncthreads();
    float m = -INFINITY;
    for (int k = 0; k < R; ++k) m = fmaxf(m, sc[k]);
    float den = 0.f;
    for (int k = 0; k < R; ++k) den += expf(sc[k] - m);
    if (l == 0) a_save[n * R + r] = expf(sc[r] - m) / den;
    if (r == 0) {
        float o = 0.f;
        for (int k = 0; k < R; ++k) o = fmaf(expf(sc[k] - m) / den, x[(n * R + k) * L + l], o);
        out[n * L + l] = o;
    }
}

// Per-example part of the backward: g_x, g_other, plus g_pre [N,R,L] (gradient at
// the scorer's pre-activation) and g_sc [N,R] (gradient at the scores) for the
// parameter reduction kernel below.
__global__ void narre_attn_bwd_kernel(const float *__restrict__ x, const float *__restrict__ W0,
                                      const float *__restrict__ w3, const float *__restrict__ mult,
                                      const float *__restrict__ h_save, const float *__restrict__ a_save,
                                      const float *__restrict__ g_out,
                                      float *__restrict__ g_x, float *__restrict__ g_other,
                                      float *__restrict__ g_pre, float *__restrict__ g_sc, int R, int L) {
    __shared__ float gp[MAXR][MAXL + 1];
    __shared__ float ga[MAXR];
    __shared__ float gs[MAXR];
    const int l = threadIdx.x, r = threadIdx.y;
    const int64_t n = blockIdx.x;
    const int64_t row = n * R + r;
    if (l == 0) {
        float t = 0.f;
        for (int k = 0; k < L; ++k) t = fmaf(g_out[n * L + k], x[row * L + k], t);
        ga[r] = t;
    }
    __syncthreads();
    float dot = 0.f;
    for (int k = 0; k < R; ++k) dot = fmaf(a_save[n * R + k], ga[k], dot);
    const float a = a_save[row];
    const float gsc = a * (ga[r] - dot);
    if (l == 0) { gs[r] = gsc; g_sc[row] = gsc; }
    const float h = h_save[row * L + l];
    float gpre = 0.f;
    if (h > 0.f) gpre = gsc * w3[l] * (mult ? mult[row * L + l] : 1.f);
    gp[r][l] = gpre;
    g_pre[row * L + l] = gpre;
    __syncthreads();
    // g_cat[r][c] = sum_k g_pre[r][k] W0[k][c]; c = l -> x half, c = L + l -> other half
    float sx = 0.f, so = 0.f;
    for (int k = 0; k < L; ++k) {
        sx = fmaf(gp[r][k], W0[(size_t)k * 2 * L + l], sx);
        so = fmaf(gp[r][k], W0[(size_t)k * 2 * L + L + l], so);
    }
    g_x[row * L + l] = sx + a * g_out[n * L + l];
    g_other[row * L + l] = so;
}

// Parameter gradients, reduced over all N*R rows in a fixed order.
// grid = (L + 1, nsplit): block l < L -> row l of g_W0 (2L columns) and g_b0[l]; block L ->
// g_w3 (L columns) and g_b3; workgroup (., s) reduces its slice of rows, a finish kernel adds
// the nsplit partials (deterministic, no atomics).  blockDim = (128, 4): threadIdx.x = column,
// threadIdx.y = one of 4 row groups.
__global__ void narre_attn_bwd_p_kernel(const float *__restrict__ x, const float *__restrict__ other,
                                        const float *__restrict__ h_save, const float *__restrict__ g_pre,
                                        const float *__restrict__ g_sc, float *__restrict__ part,
                                        int64_t rows, int L, int per_split) {
    __shared__ float red[4][128];
    const int c = threadIdx.x, rg = threadIdx.y;
    const int blk = blockIdx.x, sp = blockIdx.y;
    const int64_t r0 = (int64_t)sp * per_split, r1 = min(rows, r0 + (int64_t)per_split);
    float acc = 0.f;
    if (blk < L) {
#pragma unroll 4
        for (int64_t i = r0 + rg; i < r1; i += 4) {
            const float g = g_pre[i * L + blk];
            float val = 0.f;
            if (c < L) val = x[i * L + c];
            else if (c < 2 * L) val = other[i * L + c - L];
            else if (c == 2 * L) val = 1.f;
            acc = fmaf(g, val, acc);
        }
    } else {
#pragma unroll 4
        for (int64_t i = r0 + rg; i < r1; i += 4) {
            const float g = g_sc[i];
            float val = 0.f;
            if (c < L) val = h_save[i * L + c];
            else if (c == L) val = 1.f;
            acc = fmaf(g, val, acc);
        }
    }
    red[rg][c] = acc;
    __syncthreads();
    if (rg == 0) part[((size_t)sp * (L + 1) + blk) * 128 + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

__global__ void narre_attn_bwd_p_finish_kernel(const float *__restrict__ part, float *__restrict__ g_W0,
                                               float *__restrict__ g_b0, float *__restrict__ g_w3,
                                               float *__restrict__ g_b3, int L, int nsplit) {
    const int blk = blockIdx.x, c = threadIdx.x;            // blockDim = 128
    float t = 0.f;
    for (int s = 0; s < nsplit; ++s) t += part[((size_t)s * (L + 1) + blk) * 128 + c];
    if (blk < L) {
        if (c < 2 * L) g_W0[(size_t)blk * 2 * L + c] = t;
        else if (c == 2 * L) g_b0[blk] = t;
    } else {
        if (c < L) g_w3[c] = t;
        else if (c == L) g_b3[0] = t;
    }
}

static inline int attn_splits(int64_t rows) {
    int s = (int)(rows / 128);
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    return s;
}

}  // namespace r4r

using namespace r4r;

extern "C" size_t r4r_narre_attn_ws_bytes(int64_t N, int R, int L) {
    return (size_t)(N * R * L + N * R + (int64_t)attn_splits(N * R) * (L + 1) * 128) * sizeof(float);
}

extern "C" int r4r_narre_attn_fwd(const float *x, const float *other, const float *W0, const float *b0,
                                  const float *w3, const float *b3, const float *mult,
                                  float *out, float *h_save, float *a_save,
                                  int64_t N, int R, int L, void *stream) {
    R4R_REQUIRE(x && other && W0 && b0 && w3 && b3 && out && h_save && a_save, "narre_attn_fwd: null pointer");
    R4R_REQUIRE(R > 0 && R <= MAXR && L > 0 && L <= MAXL, "narre_attn_fwd: R=%d L=%d outside 1..32", R, L);
    if (N <= 0) return R4R_OK;
    narre_attn_fwd_kernel<<<(unsigned)N, dim3(L, R), 0, as_stream(stream)>>>(x, other, W0, b0, w3, b3, mult,
                                                                           out, h_save, a_save, R, L);
    return check_launch("narre_attn_fwd");
}

extern "C" int r4r_narre_attn_bwd(const float *x, const float *other, const float *W0, const float *w3,
                                  const float *mult, const float *h_save, const float *a_save,
                                  const float *g_out,
                                  float *g_x, float *g_other, float *g_W0, float *g_b0, float *g_w3, float *g_b3,
                                  void *ws, size_t ws_bytes,
                                  int64_t N, int R, int L, void *stream) {
    R4R_REQUIRE(x && other && W0 && w3 && h_save && a_save && g_out && g_x && g_other && g_W0 && g_b0 &&
                    g_w3 && g_b3 && ws, "narre_attn_bwd: null pointer");
    R4R_REQUIRE(R > 0 && R <= MAXR && L > 0 && L <= MAXL, "narre_attn_bwd: R=%d L=%d outside 1..32", R, L);
    if (ws_bytes < r4r_narre_attn_ws_bytes(N, R, L)) {
        set_error("narre_attn_bwd: workspace %zu < %zu bytes", ws_bytes, r4r_narre_attn_ws_bytes(N, R, L));
        return R4R_ERR_WORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    float *g_pre = static_cast<float *>(ws);
    float *g_sc = g_pre + N * R * L;
    if (N > 0)
        narre_attn_bwd_kernel<<<(unsigned)N, dim3(L, R), 0, st>>>(x, W0, w3, mult, h_save, a_save, g_out,
                                                                  g_x, g_other, g_pre, g_sc, R, L);
    float *part = g_sc + N * R;
    const int ns = attn_splits(N * R);
    const int per_split = (int)cdiv(N * R > 0 ? N * R : 1, ns);
    narre_attn_bwd_p_kernel<<<dim3(L + 1, ns), dim3(128, 4), 0, st>>>(x, other, h_save, g_pre, g_sc, part,
                                                                     N * R, L, per_split);
    narre_attn_bwd_p_finish_kernel<<<L + 1, 128, 0, st>>>(part, g_W0, g_b0, g_w3, g_b3, L, ns);
    return check_launch("narre_attn_bwd");
}
