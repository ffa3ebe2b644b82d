// Small fp32 ops of the rating heads for gfx950: tiny linear layers, dropout,
// factorisation machine, ID-embedding gathers / dense scatter-add, dot + bias
// heads, squared error.  All of them are launch-latency / HBM-bound, so the
// rules that matter are coalescing and one pass over the data; none of this is
// reshaped into MFMA work.
//
// Reference behaviour restated (file:line under the reference root) is cited
// per entry point in include/r4r.h.
#include "common.h"

namespace r4r {

// ------------------------------------------------------------------ linear
// y[n][o] = act(b[o] + sum_i x[n][i] * w[o][i]);  one thread per output.
__global__ void linear_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                  const float *__restrict__ b, float *__restrict__ y,
                                  int64_t N, int n_in, int n_out, int relu) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * n_out) return;
    const int64_t n = i / n_out;
    const int o = (int)(i - n * n_out);
    const float *xr = x + n * n_in, *wr = w + (size_t)o * n_in;
    float s = 0.f;
    for (int k = 0; k < n_in; ++k) s = fmaf(xr[k], wr[k], s);
    s += b[o];
    y[i] = (relu && s < 0.f) ? 0.f : s;
}

// g_x[n][i] = sum_o geff[n][o] * w[o][i],  geff = relu ? g_y * (y > 0) : g_y
__global__ void linear_bwd_x_kernel(const float *__restrict__ w, const float *__restrict__ y,
                                    const float *__restrict__ gy, float *__restrict__ gx,
                                    int64_t N, int n_in, int n_out, int relu) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * n_in) return;
    const int64_t n = i / n_in;
    const int k = (int)(i - n * n_in);
    float s = 0.f;
    for (int o = 0; o < n_out; ++o) {
        float g = gy[n * n_out + o];
        if (relu && !(y[n * n_out + o] > 0.f)) g = 0.f;
        s = fmaf(g, w[(size_t)o * n_in + k], s);
    }
    gx[i] = s;
}

// g_w[o][i] = sum_n geff[n][o] * x[n][i]; g_b[o] = sum_n geff[n][o].
// grid = (n_out, nsplit): workgroup (o, s) reduces its slice of rows with LB_ROWS row groups
// (thread k < n_in owns column k, thread n_in the bias), combined through LDS in a fixed
// order; a second tiny kernel adds the nsplit partials in a fixed order (deterministic, no
// atomics).  With nsplit == 1 the first kernel writes the result directly.
constexpr int LB_ROWS = 4;
__global__ void linear_bwd_w_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                    const float *__restrict__ gy, float *__restrict__ gw,
                                    float *__restrict__ gb, float *__restrict__ part,
                                    int64_t N, int n_in, int n_out, int relu, int per_split) {
    extern __shared__ float red[];   // [LB_ROWS][n_in + 1]
    const int o = blockIdx.x, sp = blockIdx.y;
    const int k = threadIdx.x, rg = threadIdx.y;
    const int64_t r0 = (int64_t)sp * per_split, r1 = min(N, r0 + (int64_t)per_split);
    float s = 0.f;
    if (k <= n_in) {
#pragma unroll 4
        for (int64_t n = r0 + rg; n < r1; n += LB_ROWS) {
            float g = gy[n * n_out + o];
            if (relu && !(y[n * n_out + o] > 0.f)) g = 0.f;
            s = (k < n_in) ? fmaf(g, x[n * n_in + k], s) : s + g;
        }
        red[rg * (n_in + 1) + k] = s;
    }
    __syncthreads();
    if (rg == 0 && k <= n_in) {
        float t = red[k];
        for (int r = 1; r < LB_ROWS; ++r) t += red[r * (n_in + 1) + k];
        if (part) part[((size_t)sp * n_out + o) * (n_in + 1) + k] = t;
        else if (k < n_in) gw[(size_t)o * n_in + k] = t;
        else gb[o] = t;
    }
}

__global__ void linear_bwd_w_finish_kernel(const float *__restrict__ part, float *__restrict__ gw,
                                           float *__restrict__ gb, int n_in, int n_out, int nsplit) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = n_out * (n_in + 1);
    if (i >= per) return;
    float t = 0.f;
    for (int s = 0; s < nsplit; ++s) t += part[(size_t)s * per + i];
    const int o = i / (n_in + 1), k = i - o * (n_in + 1);
    if (k < n_in) gw[(size_t)o * n_in + k] = t;
    else gb[o] = t;
}

// ----------------------------------------------------------------- dropout
// Philox4x32-10, one 128-bit draw per 4 consecutive elements.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ void dropout_fwd_kernel(const float *__restrict__ x, float *__restrict__ y,
                                   float *__restrict__ mult, int64_t n, float p, float scale,
                                   uint64_t seed, uint64_t offset, const uint64_t *__restrict__ offset_dev) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // group of 4 elements
    const int64_t i0 = g * 4;
    if (i0 >= n) return;
    const uint64_t ctr = offset + (offset_dev ? offset_dev[0] : 0) + (uint64_t)g;
    uint32_t r[4];
    philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = i0 + j;
        if (i < n) {
            const float u = (float)(r[j] >> 8) * (1.0f / 16777216.0f);   // [0, 1)
            const float m = (u >= p) ? scale : 0.f;
            mult[i] = m;
            y[i] = x[i] * m;
        }
    }
}

// counter[0] += delta: advances a device-resident Philox offset / step count AFTER the kernels
// that read it (same stream), so a captured hipGraph replays with fresh values.
__global__ void counter_add_kernel(uint64_t *__restrict__ ctr, uint64_t delta) {
    if (threadIdx.x == 0 && blockIdx.x == 0) ctr[0] += delta;
}

__global__ void mul_kernel(const float *__restrict__ a, const float *__restrict__ b,
                           float *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] * b[i];
}

__global__ void add_kernel(const float *__restrict__ a, const float *__restrict__ b,
                           float *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

// ---------------------------------------------------------------------- FM
// One wave per example: lane l holds inputs l, l + 64, ... (FM_MAX_SLOTS of them: n <= 512; the reference puts
// no bound on latent_size, hyper_params.py:63, and the FM reads 2 x latent_size inputs, DeepCoNN.py:32);
// s_k = sum_i x_i V_ik by a wave reduction (K12 of the survey: "wavefront reductions for the FM second-order
// term").  With n <= 64 every lane holds one input and the arithmetic is the single-slot form's, bit for bit.
constexpr int FM_MAX_SLOTS = 8;

__global__ void fm_fwd_kernel(const float *__restrict__ x, const float *__restrict__ V,
                              const float *__restrict__ lw, const float *__restrict__ lb,
                              float *__restrict__ out, int64_t N, int n, int k) {
    const int lane = threadIdx.x & 63;
    const int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= N) return;
    float xi[FM_MAX_SLOTS];
#pragma unroll
    for (int t = 0; t < FM_MAX_SLOTS; ++t) xi[t] = (lane + 64 * t < n) ? x[b * n + lane + 64 * t] : 0.f;
    float inter = 0.f;
    for (int kk = 0; kk < k; ++kk) {
        float a = 0.f, a2 = 0.f;
#pragma unroll
        for (int t = 0; t < FM_MAX_SLOTS; ++t) {
            if (64 * t < n) {                               // uniform
                const float v = (lane + 64 * t < n) ? V[(lane + 64 * t) * k + kk] : 0.f;
                a += xi[t] * v;
                a2 += xi[t] * xi[t] * v * v;
            }
        }
        const float s = wave_sum(a);
        const float s2 = wave_sum(a2);
        inter += s * s - s2;
    }
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < FM_MAX_SLOTS; ++t)
        if (lane + 64 * t < n) l += xi[t] * lw[lane + 64 * t];
    const float lin = wave_sum(l);
    if (lane == 0) out[b] = 0.5f * inter + lin + lb[0];
}

// g_x[b][i] = g_b * (sum_k (s_k V_ik - x_i V_ik^2) + w_i)
__global__ void fm_bwd_x_kernel(const float *__restrict__ x, const float *__restrict__ V,
                                const float *__restrict__ lw, const float *__restrict__ gout,
                                float *__restrict__ gx, int64_t N, int n, int k) {
    const int lane = threadIdx.x & 63;
    const int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= N) return;
    float xi[FM_MAX_SLOTS], acc[FM_MAX_SLOTS];
#pragma unroll
    for (int t = 0; t < FM_MAX_SLOTS; ++t) {
        xi[t] = (lane + 64 * t < n) ? x[b * n + lane + 64 * t] : 0.f;
        acc[t] = 0.f;
    }
    for (int kk = 0; kk < k; ++kk) {
        float v[FM_MAX_SLOTS], a = 0.f;
#pragma unroll
        for (int t = 0; t < FM_MAX_SLOTS; ++t) {
            v[t] = (64 * t < n && lane + 64 * t < n) ? V[(lane + 64 * t) * k + kk] : 0.f;
            a += xi[t] * v[t];
        }
        const float s = wave_sum(a);
#pragma unroll
        for (int t = 0; t < FM_MAX_SLOTS; ++t) acc[t] += s * v[t] - xi[t] * v[t] * v[t];
    }
#pragma unroll
    for (int t = 0; t < FM_MAX_SLOTS; ++t)
        if (lane + 64 * t < n) gx[b * n + lane + 64 * t] = gout[b] * (acc[t] + lw[lane + 64 * t]);
}

// g_V[i][kk] = sum_b g_b (s_bk x_bi - x_bi^2 V_ik); g_lw[i] = sum_b g_b x_bi; g_lb = sum_b g_b.
// One workgroup of 64 x 4 threads per (kk column, block of 64 inputs) plus the linear part's (kk == k);
// s_bk (over ALL n inputs) is recomputed per example by a wave reduction; rows of 4 example groups are combined
// through LDS in a fixed order.
__global__ void fm_bwd_p_kernel(const float *__restrict__ x, const float *__restrict__ V,
                                const float *__restrict__ gout, float *__restrict__ gV,
                                float *__restrict__ glw, float *__restrict__ glb,
                                int64_t N, int n, int k) {
    __shared__ float red[4][65];
    const int lane = threadIdx.x, rg = threadIdx.y;   // blockDim = (64, 4)
    const int kk = blockIdx.x;                        // kk == k -> linear part
    const int i0 = blockIdx.y * 64, mine = i0 + lane; // this workgroup's inputs
    float acc = 0.f, accb = 0.f;
    float v[FM_MAX_SLOTS];
#pragma unroll
    for (int t = 0; t < FM_MAX_SLOTS; ++t) v[t] = (kk < k && lane + 64 * t < n) ? V[(lane + 64 * t) * k + kk] : 0.f;
    const float vm = (kk < k && mine < n) ? V[mine * k + kk] : 0.f;
    for (int64_t b = rg; b < N; b += 4) {
        const float xm = (mine < n) ? x[b * n + mine] : 0.f;
        const float g = gout[b];
        if (kk < k) {
            float a = 0.f;
#pragma unroll
            for (int t = 0; t < FM_MAX_SLOTS; ++t)
                if (64 * t < n) a += ((lane + 64 * t < n) ? x[b * n + lane + 64 * t] : 0.f) * v[t];
            const float s = wave_sum(a);
            acc += g * (s * xm - xm * xm * vm);
        } else {
            acc += g * xm;
            accb += g;
        }
    }
    red[rg][lane] = acc;
    if (lane == 0) red[rg][64] = accb;
    __syncthreads();
    if (rg == 0) {
        const float t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        if (kk < k) { if (mine < n) gV[mine * k + kk] = t; }
        else {
            if (mine < n) glw[mine] = t;
            if (lane == 0 && blockIdx.y == 0) glb[0] = red[0][64] + red[1][64] + red[2][64] + red[3][64];
        }
    }
}

// ------------------------------------------------------------ embed gather
__global__ void embed_gather_kernel(const float *__restrict__ table, const int64_t *__restrict__ idx,
                                    float *__restrict__ out, int D, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * D) return;
    const int64_t r = i / D;
    const int d = (int)(i - r * D);
    out[i] = table[idx[r] * D + d];
}

__global__ void fill_zero_kernel(float *__restrict__ p, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0.f;
}

__global__ void embed_scatter_add_kernel(const float *__restrict__ gout, const int64_t *__restrict__ idx,
                                         float *__restrict__ gtable, int D, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * D) return;
    const int64_t r = i / D;
    const int d = (int)(i - r * D);
    atomicAdd(gtable + idx[r] * D + d, gout[i]);
}

// Deterministic variant: one wave per entry; the FIRST occurrence of a row (the "leader") adds
// the gradient rows of all its later duplicates in ascending entry order and owns the store --
// no atomics, the same bits on every run and on every rank.  O(n^2) index compares, meant for the
// compact (row-id, grad-row) lists of a batch (n = global batch size), not for bulk scatters.
__global__ void embed_scatter_add_ordered_kernel(const float *__restrict__ gout, const int64_t *__restrict__ idx,
                                                 float *__restrict__ gtable, int D, int64_t n) {
    const int lane = threadIdx.x & 63;
    const int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (e >= n) return;
    const int64_t row = idx[e];
    if (row < 0) return;                                     // padding entry of an all-gathered list
    bool dup = false;
    for (int64_t k = lane; k < e; k += 64) dup |= (idx[k] == row);
    if (__any(dup)) return;                                  // an earlier entry leads this row
    for (int d0 = 0; d0 < D; d0 += 64) {
        const int d = d0 + lane;
        float acc = (d < D) ? gout[e * D + d] : 0.f;
        for (int64_t k = e + 1; k < n; ++k)
            if (idx[k] == row && d < D) acc += gout[k * D + d];
        if (d < D) gtable[row * D + d] = acc;
    }
}

// -------------------------------------------------------------- rating head
__global__ void rowdot_fwd_kernel(const float *__restrict__ a, const float *__restrict__ c,
                                  float *__restrict__ out, int64_t N, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= N) return;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s = fmaf(a[b * D + d], c[b * D + d], s);
    s = wave_sum(s);
    if (lane == 0) out[b] = s;
}

__global__ void rowdot_bwd_kernel(const float *__restrict__ a, const float *__restrict__ c,
                                  const float *__restrict__ gout, float *__restrict__ ga,
                                  float *__restrict__ gc, int64_t N, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * D) return;
    const float g = gout[i / D];
    ga[i] = g * c[i];
    gc[i] = g * a[i];
}

__global__ void bias_head_fwd_kernel(const float *__restrict__ r, const float *__restrict__ ub,
                                     const float *__restrict__ ib, const float *__restrict__ gb,
                                     const int64_t *__restrict__ uid, const int64_t *__restrict__ iid,
                                     float *__restrict__ out, int64_t N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    // same association order as the reference: ((rating + user_bias) + item_bias) + global_bias;
    // without ID biases (DeepCoNN 'deepconn' mode, DeepCoNN.py:65): rating + global_bias.
    float s;
    if (!ub) s = r[i];
    else if (r) s = (r[i] + ub[uid[i]]) + ib[iid[i]];
    else s = ub[uid[i]] + ib[iid[i]];
    out[i] = s + gb[0];
}

// g_global = sum_b g_out[b], single workgroup, fixed order.
__global__ void sum_kernel(const float *__restrict__ g, float *__restrict__ out, int64_t N) {
    __shared__ float red[256];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < N; i += 256) s += g[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

__global__ void mse_kernel(const float *__restrict__ out, const float *__restrict__ y,
                           float *__restrict__ se, float *__restrict__ gout, int64_t N, float inv_denom) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float d = out[i] - y[i];
    se[i] = d * d;
    if (gout) gout[i] = 2.f * d * inv_denom;
}

// out[0] = (1/N) sum_{n,l} (a-b)^2, single workgroup, fixed order.
__global__ void sqdist_mean_fwd_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                       float *__restrict__ out, int64_t total, float inv_n) {
    __shared__ float red[256];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < total; i += 256) { const float d = a[i] - b[i]; s = fmaf(d, d, s); }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] * inv_n;
}

__global__ void sqdist_mean_bwd_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                       const float *__restrict__ gout, float *__restrict__ ga,
                                       float *__restrict__ gb, int64_t total, float inv_n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float g = gout[0] * 2.f * inv_n * (a[i] - b[i]);
    ga[i] = g;
    gb[i] = -g;
}

static inline unsigned blocks_for(int64_t n, int per = 256) { return (unsigned)((n + per - 1) / per); }

}  // namespace r4r

using namespace r4r;

extern "C" int r4r_linear_fwd(const float *x, const float *w, const float *b, float *y,
                              int64_t N, int n_in, int n_out, int relu, void *stream) {
    R4R_REQUIRE(x && w && b && y, "linear_fwd: null pointer");
    R4R_REQUIRE(N >= 0 && n_in > 0 && n_out > 0, "linear_fwd: bad sizes");
    if (N == 0) return R4R_OK;
    linear_fwd_kernel<<<blocks_for(N * n_out), 256, 0, as_stream(stream)>>>(x, w, b, y, N, n_in, n_out, relu);
    return check_launch("linear_fwd");
}

static inline int linear_bwd_splits(int64_t N) {
    int s = (int)(N / 64);                 // >= 64 rows per workgroup
    if (s > 32) s = 32;
    if (s < 1) s = 1;
    return s;
}

extern "C" size_t r4r_linear_bwd_ws_bytes(int64_t N, int n_in, int n_out) {
    const int ns = linear_bwd_splits(N);
    return ns > 1 ? (size_t)ns * n_out * (n_in + 1) * sizeof(float) : 0;
}

extern "C" int r4r_linear_bwd(const float *x, const float *w, const float *y, const float *g_y,
                              float *g_x, float *g_w, float *g_b, void *ws, size_t ws_bytes,
                              int64_t N, int n_in, int n_out, int relu, void *stream) {
    R4R_REQUIRE(x && w && g_y && g_w && g_b, "linear_bwd: null pointer");
    R4R_REQUIRE(!relu || y, "linear_bwd: relu needs the saved output");
    R4R_REQUIRE(N >= 0 && n_in > 0 && n_in <= 255 && n_out > 0, "linear_bwd: n_in %d out of range (1..255)", n_in);
    if (ws_bytes < r4r_linear_bwd_ws_bytes(N, n_in, n_out) || (r4r_linear_bwd_ws_bytes(N, n_in, n_out) && !ws)) {
        set_error("linear_bwd: workspace %zu < %zu bytes", ws_bytes, r4r_linear_bwd_ws_bytes(N, n_in, n_out));
        return R4R_ERR_WORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    if (g_x && N > 0)
        linear_bwd_x_kernel<<<blocks_for(N * n_in), 256, 0, st>>>(w, y, g_y, g_x, N, n_in, n_out, relu);
    const int tx = ((n_in + 1 + 63) / 64) * 64;
    const int ns = linear_bwd_splits(N);
    const int per_split = (int)cdiv(N > 0 ? N : 1, ns);
    float *part = ns > 1 ? static_cast<float *>(ws) : nullptr;
    linear_bwd_w_kernel<<<dim3(n_out, ns), dim3(tx, LB_ROWS), LB_ROWS * (n_in + 1) * sizeof(float), st>>>(
        x, y, g_y, g_w, g_b, part, N, n_in, n_out, relu, per_split);
    if (ns > 1) {
        const int per = n_out * (n_in + 1);
        linear_bwd_w_finish_kernel<<<(per + 255) / 256, 256, 0, st>>>(part, g_w, g_b, n_in, n_out, ns);
    }
    return check_launch("linear_bwd");
}

extern "C" int r4r_dropout_fwd(const float *x, float *y, float *mult, int64_t n, float p,
                               uint64_t seed, uint64_t offset, uint64_t *offset_dev, void *stream) {
    R4R_REQUIRE(x && y && mult, "dropout_fwd: null pointer");
    R4R_REQUIRE(p >= 0.f && p < 1.f, "dropout_fwd: p=%f outside [0,1)", (double)p);
    if (n <= 0) return R4R_OK;
    hipStream_t st = as_stream(stream);
    dropout_fwd_kernel<<<blocks_for((n + 3) / 4), 256, 0, st>>>(x, y, mult, n, p, 1.f / (1.f - p), seed, offset,
                                                                offset_dev);
    if (offset_dev) counter_add_kernel<<<1, 64, 0, st>>>(offset_dev, (uint64_t)((n + 3) / 4));
    return check_launch("dropout_fwd");
}

extern "C" int r4r_counter_add(uint64_t *counter, uint64_t delta, void *stream) {
    R4R_REQUIRE(counter, "counter_add: null pointer");
    counter_add_kernel<<<1, 64, 0, as_stream(stream)>>>(counter, delta);
    return check_launch("counter_add");
}

extern "C" int r4r_mul(const float *a, const float *b, float *out, int64_t n, void *stream) {
    R4R_REQUIRE(a && b && out, "mul: null pointer");
    if (n <= 0) return R4R_OK;
    mul_kernel<<<blocks_for(n), 256, 0, as_stream(stream)>>>(a, b, out, n);
    return check_launch("mul");
}

extern "C" int r4r_add(const float *a, const float *b, float *out, int64_t n, void *stream) {
    R4R_REQUIRE(a && b && out, "add: null pointer");
    if (n <= 0) return R4R_OK;
    add_kernel<<<blocks_for(n), 256, 0, as_stream(stream)>>>(a, b, out, n);
    return check_launch("add");
}

extern "C" int r4r_fm_fwd(const float *x, const float *V, const float *lin_w, const float *lin_b,
                          float *out, int64_t N, int n, int k, void *stream) {
    R4R_REQUIRE(x && V && lin_w && lin_b && out, "fm_fwd: null pointer");
    R4R_REQUIRE(n > 0 && n <= 64 * FM_MAX_SLOTS && k > 0 && k <= 4096, "fm_fwd: n=%d outside 1..%d or k=%d outside 1..4096", n, 64 * FM_MAX_SLOTS, k);
    if (N <= 0) return R4R_OK;
    fm_fwd_kernel<<<blocks_for(N * 64), 256, 0, as_stream(stream)>>>(x, V, lin_w, lin_b, out, N, n, k);
    return check_launch("fm_fwd");
}

extern "C" int r4r_fm_bwd(const float *x, const float *V, const float *lin_w, const float *g_out,
                          float *g_x, float *g_V, float *g_lin_w, float *g_lin_b,
                          int64_t N, int n, int k, void *stream) {
    R4R_REQUIRE(x && V && lin_w && g_out && g_x && g_V && g_lin_w && g_lin_b, "fm_bwd: null pointer");
    R4R_REQUIRE(n > 0 && n <= 64 * FM_MAX_SLOTS && k > 0 && k <= 4096, "fm_bwd: n=%d outside 1..%d or k=%d outside 1..4096", n, 64 * FM_MAX_SLOTS, k);
    hipStream_t st = as_stream(stream);
    if (N > 0) fm_bwd_x_kernel<<<blocks_for(N * 64), 256, 0, st>>>(x, V, lin_w, g_out, g_x, N, n, k);
    fm_bwd_p_kernel<<<dim3(k + 1, (n + 63) / 64), dim3(64, 4), 0, st>>>(x, V, g_out, g_V, g_lin_w, g_lin_b, N, n, k);
    return check_launch("fm_bwd");
}

extern "C" int r4r_embed_gather(const float *table, const int64_t *idx, float *out,
                                int64_t R, int D, int64_t n, void *stream) {
    R4R_REQUIRE(table && idx && out, "embed_gather: null pointer");
    R4R_REQUIRE(R > 0 && D > 0, "embed_gather: bad sizes");
    if (n <= 0) return R4R_OK;
    embed_gather_kernel<<<blocks_for(n * D), 256, 0, as_stream(stream)>>>(table, idx, out, D, n);
    return check_launch("embed_gather");
}

extern "C" int r4r_embed_scatter_add(const float *g_out, const int64_t *idx, float *g_table,
                                     int64_t R, int D, int64_t n, void *stream) {
    R4R_REQUIRE(g_out && idx && g_table, "embed_scatter_add: null pointer");
    R4R_REQUIRE(R > 0 && D > 0, "embed_scatter_add: bad sizes");
    hipStream_t st = as_stream(stream);
    const int64_t tot = R * D;
    unsigned zb = blocks_for(tot);
    if (zb > 4096) zb = 4096;
    fill_zero_kernel<<<zb, 256, 0, st>>>(g_table, tot);
    if (n > 0) embed_scatter_add_kernel<<<blocks_for(n * D), 256, 0, st>>>(g_out, idx, g_table, D, n);
    return check_launch("embed_scatter_add");
}

extern "C" int r4r_embed_scatter_add_ordered(const float *g_out, const int64_t *idx, float *g_table,
                                             int64_t R, int D, int64_t n, void *stream) {
    R4R_REQUIRE(g_out && idx && g_table, "embed_scatter_add_ordered: null pointer");
    R4R_REQUIRE(R > 0 && D > 0 && n >= 0 && n <= (1 << 20), "embed_scatter_add_ordered: bad sizes");
    hipStream_t st = as_stream(stream);
    const int64_t tot = R * D;
    unsigned zb = blocks_for(tot);
    if (zb > 4096) zb = 4096;
    fill_zero_kernel<<<zb, 256, 0, st>>>(g_table, tot);
    if (n > 0) embed_scatter_add_ordered_kernel<<<blocks_for(n * 64), 256, 0, st>>>(g_out, idx, g_table, D, n);
    return check_launch("embed_scatter_add_ordered");
}

extern "C" int r4r_rowdot_fwd(const float *a, const float *c, float *out, int64_t N, int D, void *stream) {
    R4R_REQUIRE(a && c && out && D > 0, "rowdot_fwd: bad arguments");
    if (N <= 0) return R4R_OK;
    rowdot_fwd_kernel<<<blocks_for(N * 64), 256, 0, as_stream(stream)>>>(a, c, out, N, D);
    return check_launch("rowdot_fwd");
}

extern "C" int r4r_rowdot_bwd(const float *a, const float *c, const float *g_out, float *g_a, float *g_c,
                              int64_t N, int D, void *stream) {
    R4R_REQUIRE(a && c && g_out && g_a && g_c && D > 0, "rowdot_bwd: bad arguments");
    if (N <= 0) return R4R_OK;
    rowdot_bwd_kernel<<<blocks_for(N * D), 256, 0, as_stream(stream)>>>(a, c, g_out, g_a, g_c, N, D);
    return check_launch("rowdot_bwd");
}

extern "C" int r4r_bias_head_fwd(const float *r, const float *user_bias, const float *item_bias,
                                 const float *global_bias, const int64_t *uid, const int64_t *iid,
                                 float *out, int64_t N, void *stream) {
    R4R_REQUIRE(global_bias && out, "bias_head_fwd: null pointer");
    R4R_REQUIRE(user_bias ? (item_bias && uid && iid) : (r != nullptr),
                "bias_head_fwd: need (user_bias, item_bias, uid, iid) or, without ID biases, r");
    if (N <= 0) return R4R_OK;
    bias_head_fwd_kernel<<<blocks_for(N), 256, 0, as_stream(stream)>>>(r, user_bias, item_bias, global_bias,
                                                                      uid, iid, out, N);
    return check_launch("bias_head_fwd");
}

extern "C" int r4r_bias_head_bwd(const float *g_out, const int64_t *uid, const int64_t *iid,
                                 float *g_user_bias, float *g_item_bias, float *g_global,
                                 int64_t RU, int64_t RI, int64_t N, void *stream) {
    R4R_REQUIRE(g_out && g_global, "bias_head_bwd: null pointer");
    hipStream_t st = as_stream(stream);
    if (g_user_bias) {
        R4R_REQUIRE(uid && iid && g_item_bias, "bias_head_bwd: null pointer");
        if (int rc = r4r_embed_scatter_add(g_out, uid, g_user_bias, RU, 1, N, stream)) return rc;
        if (int rc = r4r_embed_scatter_add(g_out, iid, g_item_bias, RI, 1, N, stream)) return rc;
    }
    sum_kernel<<<1, 256, 0, st>>>(g_out, g_global, N);
    return check_launch("bias_head_bwd");
}

extern "C" int r4r_mse_fwd_bwd(const float *out, const float *y, float *se, float *g_out,
                               int64_t N, float denom, void *stream) {
    R4R_REQUIRE(out && y && se, "mse_fwd_bwd: null pointer");
    R4R_REQUIRE(denom > 0.f, "mse_fwd_bwd: denom must be positive");
    if (N <= 0) return R4R_OK;
    mse_kernel<<<blocks_for(N), 256, 0, as_stream(stream)>>>(out, y, se, g_out, N, 1.f / denom);
    return check_launch("mse_fwd_bwd");
}

extern "C" int r4r_sqdist_mean_fwd(const float *a, const float *b, float *out, int64_t N, int L, void *stream) {
    R4R_REQUIRE(a && b && out && N > 0 && L > 0, "sqdist_mean_fwd: bad arguments");
    sqdist_mean_fwd_kernel<<<1, 256, 0, as_stream(stream)>>>(a, b, out, N * L, 1.f / (float)N);
    return check_launch("sqdist_mean_fwd");
}

extern "C" int r4r_sqdist_mean_bwd(const float *a, const float *b, const float *g_out, float *g_a, float *g_b,
                                   int64_t N, int L, void *stream) {
    R4R_REQUIRE(a && b && g_out && g_a && g_b && N > 0 && L > 0, "sqdist_mean_bwd: bad arguments");
    sqdist_mean_bwd_kernel<<<blocks_for(N * L), 256, 0, as_stream(stream)>>>(a, b, g_out, g_a, g_b, N * L,
                                                                            1.f / (float)N);
    return check_launch("sqdist_mean_bwd");
}
