// TextCNN tower for gfx950: word gather -> 3 x E convolution -> relu -> global
// max-pool (+argmax), and the argmax-sparse weight gradient.
//
// Reference behaviour restated (file:line under the reference root):
//   common_pytorch_models.py:14-17  Conv2d(1, 100, [3, E], padding=(2, 0))
//   common_pytorch_models.py:29-31  relu -> max_pool1d over all T+2 positions
//   DeepCoNN.py:53-54               word2vec(idx) feeding the conv
//
// Forward as an implicit GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32,
// exact fp32 == an fmaf chain):
//   M = conv positions p in [0, T+2) of one document, tiled by 128
//   N = filters, padded 100 -> 112 (7 tiles of 16)
//   K = 3 taps x E, walked as (E-chunk of 32) x (tap) x (16-wide k block)
// The A operand is never materialised: a tile stages the 130 gathered word rows
// it needs ONCE per E-chunk in LDS and tap j reads row (i + j) -- the sliding
// window is an LDS row offset.  The [N,T,E] activations and [N,F,T+2] conv
// output never touch HBM; the epilogue keeps a running (max, first-argmax) per
// (document, filter) in registers and writes one partial per tile.
//
// LDS image (floats): X [130][40] + W [112][104]; both row strides are
// == 8 (mod 16) so the ds_read_b128 fragment reads (16 rows x 4 k-quads per
// wave) are bank-conflict free (MI355X_MICROARCH.md, LDS table).
#include <stdlib.h>

#include "common.h"

namespace r4r {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MT = 128;            // conv positions per workgroup tile
constexpr int XR = MT + 2;         // staged word rows (2-row halo)
constexpr int EC = 32;             // embedding columns per K chunk
constexpr int XS = EC + 8;         // X row stride in LDS (floats)
constexpr int NP = 112;            // filters padded to 7 x 16
constexpr int NT = NP / 16;
constexpr int WS = 3 * EC + 8;     // W row stride in LDS (floats)
constexpr int FWD_THREADS = 256;
constexpr int FWD_LDS_BYTES = (XR * XS + NP * WS) * 4 + XR * 4;

static inline int n_chunks(int E) { return (E + EC - 1) / EC; }
static inline int tiles_per_doc(int T) { return (T + 2 + MT - 1) / MT; }

// ---------------------------------------------------------------------------
// Pack conv weight [F][3][E] into the per-chunk LDS image [chunk][112][104]:
//   Wp[c][n][j*32 + ee] = W[n][j][c*32 + ee]   (0 for n >= F, e >= E, pad cols)
// ---------------------------------------------------------------------------
__global__ void textcnn_pack_w_kernel(const float *__restrict__ w, float *__restrict__ wp,
                                      int E, int F, int nchunk) {
    const int total = nchunk * NP * WS;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int col = i % WS;
        const int n = (i / WS) % NP;
        const int c = i / (WS * NP);
        float v = 0.f;
        if (col < 3 * EC && n < F) {
            const int j = col / EC, e = c * EC + col % EC;
            if (e < E) v = w[((size_t)n * 3 + j) * E + e];
        }
        wp[i] = v;
    }
}

// ---------------------------------------------------------------------------
// Forward tile kernel.  grid = N * tiles_per_doc, block = 256 (4 waves); wave w
// owns conv positions [32w, 32w+32) x all 112 filters: 2 x 7 accumulators.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(FWD_THREADS, 2) void textcnn_fwd_kernel(
    const float *__restrict__ table, const int64_t *__restrict__ idx,
    const float *__restrict__ wp, const float *__restrict__ bias,
    float *__restrict__ pmax, int *__restrict__ parg,
    int T, int E, int F, int tiles, int nchunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Xs = reinterpret_cast<float *>(smem);       // [XR][XS]
    float *Wl = Xs + XR * XS;                          // [NP][WS]
    int *tok = reinterpret_cast<int *>(Wl + NP * WS);  // [XR] token id or -1 (zero row)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int doc = blockIdx.x / tiles, tile = blockIdx.x - doc * tiles;
    const int p0 = tile * MT;
    const int P = T + 2;

    for (int r = tid; r < XR; r += FWD_THREADS) {
        const int t = p0 - 2 + r;
        tok[r] = (t >= 0 && t < T) ? (int)idx[(size_t)doc * T + t] : -1;
    }
    __syncthreads();

    f32x4 acc[2][NT];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int lrow = lane & 15, q = lane >> 4;
    const int e16 = (E + 15) & ~15;   // K per tap rounded up to the 16-wide block

    for (int c = 0; c < nchunk; ++c) {
        const int e0 = c * EC;
        // ---- stage the gathered word rows of this E-chunk: XR rows x 8 float4
        for (int i = tid; i < XR * (EC / 4); i += FWD_THREADS) {
            const int r = i >> 3, c4 = i & 7;
            const int e = e0 + c4 * 4;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int tk = tok[r];
            if (tk >= 0 && e < E) v = *reinterpret_cast<const f32x4 *>(table + (size_t)tk * E + e);
            *reinterpret_cast<f32x4 *>(Xs + r * XS + c4 * 4) = v;
        }
        // ---- stage the weight chunk: a linear copy of the packed LDS image
        const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(wp + (size_t)c * NP * WS);
        for (int i = tid; i < NP * WS / 4; i += FWD_THREADS)
            reinterpret_cast<f32x4 *>(Wl)[i] = wsrc[i];
        __syncthreads();

        const int nblk = min(EC / 16, (e16 - e0) / 16);
        for (int j = 0; j < 3; ++j) {
            for (int g = 0; g < nblk; ++g) {
                f32x4 a[2], b[NT];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    a[mi] = *reinterpret_cast<const f32x4 *>(
                        Xs + (wave * 32 + mi * 16 + lrow + j) * XS + g * 16 + q * 4);
#pragma unroll
                for (int ni = 0; ni < NT; ++ni)
                    b[ni] = *reinterpret_cast<const f32x4 *>(
                        Wl + (ni * 16 + lrow) * WS + j * EC + g * 16 + q * 4);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NT; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                a[mi][kk], b[ni][kk], acc[mi][ni], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: (max, first argmax) over this tile's positions, per filter.
    // C layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg.
    float *redv = Xs;                                   // [4 waves][NP]
    int *redp = reinterpret_cast<int *>(Xs + 4 * NP);   // [4 waves][NP]
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        const int col = ni * 16 + lrow;
        const float bc = (col < F) ? bias[col] : 0.f;
        float best = -INFINITY;
        int bp = 0x7fffffff;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = p0 + wave * 32 + mi * 16 + q * 4 + r;
                const float v = acc[mi][ni][r] + bc;
                if (p < P && v > best) { best = v; bp = p; }
            }
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const float ov = __shfl_xor(best, off);
            const int op = __shfl_xor(bp, off);
            if (ov > best || (ov == best && op < bp)) { best = ov; bp = op; }
        }
        if (q == 0) { redv[wave * NP + col] = best; redp[wave * NP + col] = bp; }
    }
    __syncthreads();
    if (tid < NP) {
        float best = redv[tid];
        int bp = redp[tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float ov = redv[w * NP + tid];
            const int op = redp[w * NP + tid];
            if (ov > best || (ov == best && op < bp)) { best = ov; bp = op; }
        }
        pmax[(size_t)blockIdx.x * NP + tid] = best;
        parg[(size_t)blockIdx.x * NP + tid] = bp;
    }
}


// ---------------------------------------------------------------------------
// Forward tile kernel v2: same tiling, plus a register prefetch of the NEXT
// E-chunk (issue the global loads early, write them to LDS late): the gather
// and weight-image loads of chunk c+1 are in flight while chunk c's 336 MFMAs
// per wave run, so the only exposed staging cost per chunk is the ds_write pass
// and its two barriers -- which the co-resident second workgroup covers.
// ---------------------------------------------------------------------------
constexpr int W_VEC = NP * WS / 4;                       // 2912 float4 per weight chunk
constexpr int W_PER_THREAD = (W_VEC + FWD_THREADS - 1) / FWD_THREADS;   // 12

struct EpilogueOut { float *pmax; int *parg; };

__device__ __forceinline__ void tile_compute(const float *Xs, const float *Wl, f32x4 (&acc)[2][NT],
                                             int wave, int lrow, int q, int nblk) {
    for (int j = 0; j < 3; ++j) {
        for (int g = 0; g < nblk; ++g) {
            f32x4 a[2], b[NT];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                a[mi] = *reinterpret_cast<const f32x4 *>(Xs + (wave * 32 + mi * 16 + lrow + j) * XS + g * 16 + q * 4);
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
                b[ni] = *reinterpret_cast<const f32x4 *>(Wl + (ni * 16 + lrow) * WS + j * EC + g * 16 + q * 4);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][kk], b[ni][kk], acc[mi][ni], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ void tile_epilogue(float *Xs, const f32x4 (&acc)[2][NT], const float *bias,
                                              float *pmax, int *parg, int p0, int P, int F,
                                              int tid, int wave, int lrow, int q) {
    float *redv = Xs;                                   // [4 waves][NP]
    int *redp = reinterpret_cast<int *>(Xs + 4 * NP);   // [4 waves][NP]
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        const int col = ni * 16 + lrow;
        const float bc = (col < F) ? bias[col] : 0.f;
        float best = -INFINITY;
        int bp = 0x7fffffff;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = p0 + wave * 32 + mi * 16 + q * 4 + r;
                const float v = acc[mi][ni][r] + bc;
                if (p < P && v > best) { best = v; bp = p; }
            }
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const float ov = __shfl_xor(best, off);
            const int op = __shfl_xor(bp, off);
            if (ov > best || (ov == best && op < bp)) { best = ov; bp = op; }
        }
        if (q == 0) { redv[wave * NP + col] = best; redp[wave * NP + col] = bp; }
    }
    __syncthreads();
    if (tid < NP) {
        float best = redv[tid];
        int bp = redp[tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float ov = redv[w * NP + tid];
            const int op = redp[w * NP + tid];
            if (ov > best || (ov == best && op < bp)) { best = ov; bp = op; }
        }
        pmax[(size_t)blockIdx.x * NP + tid] = best;
        parg[(size_t)blockIdx.x * NP + tid] = bp;
    }
}

__global__ __launch_bounds__(FWD_THREADS, 2) void textcnn_fwd_kernel_v2(
    const float *__restrict__ table, const int64_t *__restrict__ idx,
    const float *__restrict__ wp, const float *__restrict__ bias,
    float *__restrict__ pmax, int *__restrict__ parg,
    int T, int E, int F, int tiles, int nchunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Xs = reinterpret_cast<float *>(smem);       // [XR][XS]
    float *Wl = Xs + XR * XS;                          // [NP][WS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int doc = blockIdx.x / tiles, tile = blockIdx.x - doc * tiles;
    const int p0 = tile * MT;
    const int P = T + 2;
    const int lrow = lane & 15, q = lane >> 4;
    const int e16 = (E + 15) & ~15;

    // This thread stages float4 column c4 of rows (tid>>3) + 32k, k = 0..3, and -- for
    // tid < 16 -- of the two halo rows 128, 129.  Row offsets into the table (in floats),
    // -1 for rows outside the document (zero rows).
    const int c4 = tid & 7;
    long xoff[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int r = (k < 4) ? (tid >> 3) + 32 * k : 128 + (tid >> 3);
        const int t = p0 - 2 + r;
        const bool live = (k < 4 || tid < 16) && t >= 0 && t < T;
        xoff[k] = live ? (long)idx[(size_t)doc * T + t] * E : -1;
    }

    f32x4 xr[5], wr[W_PER_THREAD];
    auto issue_loads = [&](int c) {
        const int e = c * EC + c4 * 4;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            xr[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (xoff[k] >= 0 && e < E) xr[k] = *reinterpret_cast<const f32x4 *>(table + xoff[k] + e);
        }
        const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(wp + (size_t)c * NP * WS);
#pragma unroll
        for (int k = 0; k < W_PER_THREAD; ++k) {
            const int i = tid + k * FWD_THREADS;
            if (i < W_VEC) wr[k] = wsrc[i];
        }
    };
    auto write_lds = [&]() {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int r = (k < 4) ? (tid >> 3) + 32 * k : 128 + (tid >> 3);
            if (k < 4 || tid < 16) *reinterpret_cast<f32x4 *>(Xs + r * XS + c4 * 4) = xr[k];
        }
#pragma unroll
        for (int k = 0; k < W_PER_THREAD; ++k) {
            const int i = tid + k * FWD_THREADS;
            if (i < W_VEC) reinterpret_cast<f32x4 *>(Wl)[i] = wr[k];
        }
    };

    f32x4 acc[2][NT];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue_loads(0);
    write_lds();
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        if (c + 1 < nchunk) issue_loads(c + 1);          // in flight during the MFMAs below
        const int nblk = min(EC / 16, (e16 - c * EC) / 16);
        tile_compute(Xs, Wl, acc, wave, lrow, q, nblk);
        __syncthreads();                                 // every wave is done reading chunk c
        if (c + 1 < nchunk) {
            write_lds();
            __syncthreads();
        }
    }
    tile_epilogue(Xs, acc, bias, pmax, parg, p0, P, F, tid, wave, lrow, q);
}


// ---------------------------------------------------------------------------
// Forward tile kernel v3: LDS double buffering.  PMC on v2 showed the matrix pipe
// 75 % busy: the two co-resident workgroups run in phase and both sit in the
// "barrier -> ds_write -> barrier" bubble at the same time.  Here the E-chunk is 16
// columns so TWO (X, W) images fit: chunk c+1 is written into the other buffer at
// the START of chunk c's compute (the ds_writes drain under the MFMAs), chunk c+2's
// global loads are issued right after, and there is ONE barrier per chunk and no
// exposed staging at all.
//   NW = 8: 256 positions / workgroup, 1 workgroup per CU (2 waves per SIMD),
//           weight image staged once per 256 rows
//   NW = 4: 128 positions / workgroup, 2 workgroups per CU (NARRE's short reviews)
// LDS per buffer (floats): X [32 NW + 2][24] + W [112][56]; strides == 8 (mod 16).
// ---------------------------------------------------------------------------
constexpr int EC3 = 16;
constexpr int XS3 = EC3 + 8;            // 24
constexpr int WS3 = 3 * EC3 + 8;        // 56
constexpr int W3_VEC = NP * WS3 / 4;    // 1568 float4 per weight chunk

template <int NW>
struct V3 {
    static constexpr int THREADS = 64 * NW;
    static constexpr int MTILE = 32 * NW;
    static constexpr int XROWS = MTILE + 2;
    static constexpr int BUF_FLOATS = XROWS * XS3 + NP * WS3;
    static constexpr int LDS_BYTES = 2 * BUF_FLOATS * 4;
    static constexpr int XK = MTILE * 4 / THREADS;                    // = 2 float4 of X per thread (+ halo)
    static constexpr int WK = (W3_VEC + THREADS - 1) / THREADS;       // 4 (NW=8) or 7 (NW=4)
};

__global__ void textcnn_pack_w3_kernel(const float *__restrict__ w, float *__restrict__ wp,
                                       int E, int F, int nchunk) {
    const int total = nchunk * NP * WS3;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int col = i % WS3;
        const int n = (i / WS3) % NP;
        const int c = i / (WS3 * NP);
        float v = 0.f;
        if (col < 3 * EC3 && n < F) {
            const int j = col / EC3, e = c * EC3 + col % EC3;
            if (e < E) v = w[((size_t)n * 3 + j) * E + e];
        }
        wp[i] = v;
    }
}

// ABL (timing-only ablations, results are WRONG when ABL != 0; used by scratch/bench_conv.py):
//   1 = no staging inside the loop, 2 = also no barrier, 3 = also no LDS fragment reads
template <int NW, int ABL = 0>
__global__ __launch_bounds__(64 * NW, 2) void textcnn_fwd_kernel_v3(
    const float *__restrict__ table, const int64_t *__restrict__ idx,
    const float *__restrict__ wp, const float *__restrict__ bias,
    float *__restrict__ pmax, int *__restrict__ parg,
    int T, int E, int F, int tiles, int nchunk) {
    using C = V3<NW>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lds = reinterpret_cast<float *>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int doc = blockIdx.x / tiles, tile = blockIdx.x - doc * tiles;
    const int p0 = tile * C::MTILE;
    const int P = T + 2;
    const int lrow = lane & 15, q = lane >> 4;

    // staging role of this thread: float4 column c4 (of 4) of rows (tid>>2) + (THREADS/4) k,
    // k < XK, plus -- for tid < 8 -- of the two halo rows MTILE, MTILE+1
    const int c4 = tid & 3;
    long xoff[C::XK + 1];
#pragma unroll
    for (int k = 0; k <= C::XK; ++k) {
        const int r = (k < C::XK) ? (tid >> 2) + (C::THREADS / 4) * k : C::MTILE + (tid >> 2);
        const int t = p0 - 2 + r;
        const bool live = (k < C::XK || tid < 8) && t >= 0 && t < T;
        xoff[k] = live ? (long)idx[(size_t)doc * T + t] * E : -1;
    }

    f32x4 xr[C::XK + 1], wr[C::WK];
    auto issue_loads = [&](int c) {
        const int e = c * EC3 + c4 * 4;
#pragma unroll
        for (int k = 0; k <= C::XK; ++k) {
            xr[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (xoff[k] >= 0 && e < E) xr[k] = *reinterpret_cast<const f32x4 *>(table + xoff[k] + e);
        }
        const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(wp + (size_t)c * NP * WS3);
#pragma unroll
        for (int k = 0; k < C::WK; ++k) {
            const int i = tid + k * C::THREADS;
            if (i < W3_VEC) wr[k] = wsrc[i];
        }
    };
    auto write_lds = [&](float *buf) {
        float *Xs = buf, *Wl = buf + C::XROWS * XS3;
#pragma unroll
        for (int k = 0; k <= C::XK; ++k) {
            const int r = (k < C::XK) ? (tid >> 2) + (C::THREADS / 4) * k : C::MTILE + (tid >> 2);
            if (k < C::XK || tid < 8) *reinterpret_cast<f32x4 *>(Xs + r * XS3 + c4 * 4) = xr[k];
        }
#pragma unroll
        for (int k = 0; k < C::WK; ++k) {
            const int i = tid + k * C::THREADS;
            if (i < W3_VEC) reinterpret_cast<f32x4 *>(Wl)[i] = wr[k];
        }
    };

    f32x4 acc[2][NT];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue_loads(0);
    write_lds(lds);
    if (nchunk > 1) issue_loads(1);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        float *cur = lds + ((ABL ? 0 : c) & 1) * C::BUF_FLOATS;
        if (ABL == 0 && c + 1 < nchunk) {
            write_lds(lds + ((c + 1) & 1) * C::BUF_FLOATS);   // chunk c+1 -> the other buffer
            if (c + 2 < nchunk) issue_loads(c + 2);            // in flight for a whole chunk
        }
        const float *Xs = cur, *Wl = cur + C::XROWS * XS3;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            f32x4 a[2], b[NT];
            if (ABL == 3 && (c > 0 || j > 0)) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) { a[mi] = acc[mi][0]; asm volatile("" : "+v"(a[mi])); }
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) { b[ni] = acc[0][ni]; asm volatile("" : "+v"(b[ni])); }
            } else {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                a[mi] = *reinterpret_cast<const f32x4 *>(Xs + (wave * 32 + mi * 16 + lrow + j) * XS3 + q * 4);
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
                b[ni] = *reinterpret_cast<const f32x4 *>(Wl + (ni * 16 + lrow) * WS3 + j * EC3 + q * 4);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][kk], b[ni][kk], acc[mi][ni], 0, 0, 0);
        }
        if (ABL < 2) __syncthreads();
    }

    // ---- epilogue (NW waves)
    float *redv = lds;                                    // [NW][NP]
    int *redp = reinterpret_cast<int *>(lds + NW * NP);   // [NW][NP]
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        const int col = ni * 16 + lrow;
        const float bc = (col < F) ? bias[col] : 0.f;
        float best = -INFINITY;
        int bp = 0x7fffffff;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = p0 + wave * 32 + mi * 16 + q * 4 + r;
                const float v = acc[mi][ni][r] + bc;
                if (p < P && v > best) { best = v; bp = p; }
            }
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const float ov = __shfl_xor(best, off);
            const int op = __shfl_xor(bp, off);
            if (ov > best || (ov == best && op < bp)) { best = ov; bp = op; }
        }
        if (q == 0) { redv[wave * NP + col] = best; redp[wave * NP + col] = bp; }
    }
    __syncthreads();
    if (tid < NP) {
        float best = redv[tid];
        int bp = redp[tid];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const float ov = redv[w * NP + tid];
            const int op = redp[w * NP + tid];
            if (ov > best || (ov == best && op < bp)) { best = ov; bp = op; }
        }
        pmax[(size_t)blockIdx.x * NP + tid] = best;
        parg[(size_t)blockIdx.x * NP + tid] = bp;
    }
}

// Combine the per-tile partials of one document, apply relu:
//   pooled = max(0, max_p conv), argmax = first p of the max, -1 if pooled == 0.
__global__ void textcnn_pool_finish_kernel(const float *__restrict__ pmax, const int *__restrict__ parg,
                                           float *__restrict__ pooled, int *__restrict__ argmax,
                                           int64_t N, int F, int tiles) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * F) return;
    const int64_t doc = i / F;
    const int f = (int)(i - doc * F);
    float best = -INFINITY;
    int bp = -1;
    for (int t = 0; t < tiles; ++t) {
        const float v = pmax[((size_t)doc * tiles + t) * NP + f];
        if (v > best) { best = v; bp = parg[((size_t)doc * tiles + t) * NP + f]; }
    }
    if (best > 0.f) { pooled[i] = best; argmax[i] = bp; }
    else { pooled[i] = 0.f; argmax[i] = -1; }
}

// ---------------------------------------------------------------------------
// Argmax-sparse weight gradient.  grid = (F, nsplit); block = 256 threads, each
// owning one float4 column group of the [3][E] window (looped if 3E/4 > 256).
// Workgroup (f, s) sums its slice of documents; a second kernel adds the nsplit
// partials in a fixed order (deterministic, no atomics).
// ---------------------------------------------------------------------------
constexpr int WG_THREADS = 256;

__global__ __launch_bounds__(WG_THREADS) void textcnn_wgrad_kernel(
    const float *__restrict__ table, const int64_t *__restrict__ idx,
    const float *__restrict__ gp, const int *__restrict__ argmax,
    float *__restrict__ part_w, float *__restrict__ part_b,
    int64_t N, int T, int E, int F, int per_split) {
    const int f = blockIdx.x, s = blockIdx.y;
    const int64_t n0 = (int64_t)s * per_split;
    const int64_t n1 = min(N, n0 + (int64_t)per_split);
    const int nvec = 3 * E / 4;
    for (int v = threadIdx.x; v < nvec; v += WG_THREADS) {
        const int j = (v * 4) / E;
        const int e = v * 4 - j * E;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int64_t n = n0; n < n1; ++n) {
            const int p = argmax[n * F + f];
            if (p < 0) continue;
            const int t = p - 2 + j;
            if (t < 0 || t >= T) continue;
            const float g = gp[n * F + f];
            const int64_t tk = idx[n * T + t];
            const f32x4 x = *reinterpret_cast<const f32x4 *>(table + (size_t)tk * E + e);
            acc += g * x;
        }
        *reinterpret_cast<f32x4 *>(part_w + ((size_t)s * F + f) * 3 * E + v * 4) = acc;
    }
    if (threadIdx.x == 0) {
        float sb = 0.f;
        for (int64_t n = n0; n < n1; ++n)
            if (argmax[n * F + f] >= 0) sb += gp[n * F + f];
        part_b[(size_t)s * F + f] = sb;
    }
}

__global__ void textcnn_wgrad_reduce_kernel(const float *__restrict__ part_w, const float *__restrict__ part_b,
                                            float *__restrict__ dw, float *__restrict__ db,
                                            int E, int F, int nsplit) {
    const int nw = F * 3 * E;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nw) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += part_w[(size_t)k * nw + i];
        dw[i] = s;
    } else if (i < nw + F) {
        const int f = i - nw;
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += part_b[(size_t)k * F + f];
        db[f] = s;
    }
}

static inline int wgrad_splits(int64_t N) {
    // >= 1024 workgroups when the batch allows it, at least 8 documents per split
    int s = (int)cdiv(N, 8);
    if (s > 16) s = 16;
    if (s < 1) s = 1;
    return s;
}

// Kernel generation used by r4r_textcnn_fwd.  R4R_TEXTCNN_FWD=<n> pins one for A/B runs:
//   1 = v1 (no prefetch), 2 = v2 (register prefetch), 3 = v3<8> (double-buffered, 256-row
//   tiles), 4 = v3<4> (double-buffered, 128-row tiles)
static int fwd_variant(int T) {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("R4R_TEXTCNN_FWD");
        v = e ? atoi(e) : 0;
    }
    if (v >= 1 && v <= 7) return v;        // 5..7: timing-only ablations of v3<4> (wrong results)
    return (T + 2 > 160) ? 3 : 4;          // long documents: 256-row tiles; short reviews: 128
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace r4r

using namespace r4r;

extern "C" size_t r4r_textcnn_ws_bytes(int64_t N, int T, int E, int F) {
    if (N < 0 || T <= 0 || E <= 0 || F <= 0) return 0;
    const size_t img_v2 = (size_t)n_chunks(E) * NP * WS, img_v3 = (size_t)((E + EC3 - 1) / EC3) * NP * WS3;
    const size_t fwd = align256((img_v2 > img_v3 ? img_v2 : img_v3) * 4) +
                       2 * align256((size_t)N * tiles_per_doc(T) * NP * 4);   // tiles of the smallest MT
    const int ns = wgrad_splits(N);
    const size_t bwd = align256((size_t)ns * F * 3 * E * 4) + align256((size_t)ns * F * 4);
    return fwd > bwd ? fwd : bwd;
}

static int check_tower_args(const void *table, int64_t V, const void *idx, int64_t N, int T, int E, int F) {
    R4R_REQUIRE(table && idx, "textcnn: null table/idx");
    R4R_REQUIRE(V > 0 && N >= 0 && T > 0, "textcnn: bad sizes V=%lld N=%lld T=%d", (long long)V, (long long)N, T);
    R4R_REQUIRE(E > 0 && E % 4 == 0, "textcnn: word_embed_size %d must be a positive multiple of 4 "
                                     "(pad the frozen table on the host otherwise)", E);
    R4R_REQUIRE(F > 0 && F <= NP, "textcnn: %d filters > %d supported", F, NP);
    R4R_REQUIRE(N * (int64_t)tiles_per_doc(T) < (1ll << 31), "textcnn: grid too large");
    return R4R_OK;
}

extern "C" int r4r_textcnn_fwd(const float *table, int64_t V, const int64_t *idx,
                               const float *conv_w, const float *conv_b,
                               float *pooled, int32_t *argmax,
                               void *ws, size_t ws_bytes,
                               int64_t N, int T, int E, int F, void *stream) {
    if (int rc = check_tower_args(table, V, idx, N, T, E, F)) return rc;
    R4R_REQUIRE(conv_w && conv_b && pooled && argmax && ws, "textcnn_fwd: null pointer");
    if (ws_bytes < r4r_textcnn_ws_bytes(N, T, E, F)) {
        set_error("textcnn_fwd: workspace %zu < %zu bytes", ws_bytes, r4r_textcnn_ws_bytes(N, T, E, F));
        return R4R_ERR_WORKSPACE;
    }
    if (N == 0) return R4R_OK;
    hipStream_t st = as_stream(stream);
    const int variant = fwd_variant(T);
    const bool v3 = variant >= 3;
    const int mtile = (variant == 3) ? 256 : 128;
    const int nchunk = v3 ? (E + EC3 - 1) / EC3 : n_chunks(E);
    const int tiles = (T + 2 + mtile - 1) / mtile;
    const size_t img = (size_t)nchunk * NP * (v3 ? WS3 : WS);
    char *base = static_cast<char *>(ws);
    float *wp = reinterpret_cast<float *>(base);
    base += align256(img * 4);
    float *pmax = reinterpret_cast<float *>(base);
    base += align256((size_t)N * tiles * NP * 4);
    int *parg = reinterpret_cast<int *>(base);

    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(textcnn_fwd_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, FWD_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(textcnn_fwd_kernel_v2),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, FWD_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(textcnn_fwd_kernel_v3<8>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, V3<8>::LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(textcnn_fwd_kernel_v3<4>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, V3<4>::LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(textcnn_fwd_kernel_v3<4, 1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, V3<4>::LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(textcnn_fwd_kernel_v3<4, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, V3<4>::LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(textcnn_fwd_kernel_v3<4, 3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, V3<4>::LDS_BYTES);
        attr_set = true;
    }
    if (v3)
        textcnn_pack_w3_kernel<<<(unsigned)((img + 255) / 256), 256, 0, st>>>(conv_w, wp, E, F, nchunk);
    else
        textcnn_pack_w_kernel<<<(unsigned)((img + 255) / 256), 256, 0, st>>>(conv_w, wp, E, F, nchunk);
    {
        ScopedTiming tm(R4R_TIMING_TEXTCNN_FWD, st);
        const unsigned grid = (unsigned)(N * tiles);
        if (variant == 1)
            textcnn_fwd_kernel<<<grid, FWD_THREADS, FWD_LDS_BYTES, st>>>(
                table, idx, wp, conv_b, pmax, parg, T, E, F, tiles, nchunk);
        else if (variant == 2)
            textcnn_fwd_kernel_v2<<<grid, FWD_THREADS, FWD_LDS_BYTES, st>>>(
                table, idx, wp, conv_b, pmax, parg, T, E, F, tiles, nchunk);
        else if (variant == 3)
            textcnn_fwd_kernel_v3<8><<<grid, V3<8>::THREADS, V3<8>::LDS_BYTES, st>>>(
                table, idx, wp, conv_b, pmax, parg, T, E, F, tiles, nchunk);
        else if (variant == 4)
            textcnn_fwd_kernel_v3<4><<<grid, V3<4>::THREADS, V3<4>::LDS_BYTES, st>>>(
                table, idx, wp, conv_b, pmax, parg, T, E, F, tiles, nchunk);
        else if (variant == 5)
            textcnn_fwd_kernel_v3<4, 1><<<grid, V3<4>::THREADS, V3<4>::LDS_BYTES, st>>>(
                table, idx, wp, conv_b, pmax, parg, T, E, F, tiles, nchunk);
        else if (variant == 6)
            textcnn_fwd_kernel_v3<4, 2><<<grid, V3<4>::THREADS, V3<4>::LDS_BYTES, st>>>(
                table, idx, wp, conv_b, pmax, parg, T, E, F, tiles, nchunk);
        else
            textcnn_fwd_kernel_v3<4, 3><<<grid, V3<4>::THREADS, V3<4>::LDS_BYTES, st>>>(
                table, idx, wp, conv_b, pmax, parg, T, E, F, tiles, nchunk);
    }
    textcnn_pool_finish_kernel<<<(unsigned)cdiv(N * F, 256), 256, 0, st>>>(pmax, parg, pooled, argmax, N, F, tiles);
    return check_launch("textcnn_fwd");
}

extern "C" int r4r_textcnn_wgrad(const float *table, int64_t V, const int64_t *idx,
                                 const float *g_pooled, const int32_t *argmax,
                                 float *d_conv_w, float *d_conv_b,
                                 void *ws, size_t ws_bytes,
                                 int64_t N, int T, int E, int F, void *stream) {
    if (int rc = check_tower_args(table, V, idx, N, T, E, F)) return rc;
    R4R_REQUIRE(g_pooled && argmax && d_conv_w && d_conv_b && ws, "textcnn_wgrad: null pointer");
    if (ws_bytes < r4r_textcnn_ws_bytes(N, T, E, F)) {
        set_error("textcnn_wgrad: workspace %zu < %zu bytes", ws_bytes, r4r_textcnn_ws_bytes(N, T, E, F));
        return R4R_ERR_WORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    const int ns = wgrad_splits(N);
    const int per_split = (int)cdiv(N > 0 ? N : 1, ns);
    char *base = static_cast<char *>(ws);
    float *part_w = reinterpret_cast<float *>(base);
    base += align256((size_t)ns * F * 3 * E * 4);
    float *part_b = reinterpret_cast<float *>(base);
    {
        ScopedTiming tm(R4R_TIMING_TEXTCNN_WGRAD, st);
        textcnn_wgrad_kernel<<<dim3(F, ns), WG_THREADS, 0, st>>>(table, idx, g_pooled, argmax, part_w, part_b,
                                                                N, T, E, F, per_split);
    }
    const int tot = F * 3 * E + F;
    textcnn_wgrad_reduce_kernel<<<(tot + 255) / 256, 256, 0, st>>>(part_w, part_b, d_conv_w, d_conv_b, E, F, ns);
    return check_launch("textcnn_wgrad");
}
