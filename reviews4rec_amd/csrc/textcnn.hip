// TextCNN tower for gfx950: word gather -> 3 x E convolution -> relu -> global
// max-pool (+argmax), and the argmax-sparse weight gradient.
//
// Reference behaviour restated (file:line under the reference root):
//   common_pytorch_models.py:14-17  Conv2d(1, 100, [3, E], padding=(2, 0))
//   common_pytorch_models.py:29-31  relu -> max_pool1d over all T+2 positions
//   DeepCoNN.py:53-54               word2vec(idx) feeding the conv
//
// Forward = implicit GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact
// fp32 == an fmaf chain):
//   M = conv positions p in [0, T+2) of one document, tiled by 32 x NW (NW waves)
//   N = filters, padded 100 -> 112 (7 column tiles of 16)
//   K = 3 taps x E, walked as (E-chunk of 16) x (tap)
// The A operand is never materialised: a tile stages the gathered word rows it
// needs ONCE per E-chunk in LDS and tap j reads row (i + j) -- the sliding window
// is an LDS row offset.  The [N,T,E] activations and [N,F,T+2] conv output never
// touch HBM; the epilogue keeps a running (max, first-argmax) per (document,
// filter) in registers and writes one partial per tile.
//
// Pipeline (measured history in DESIGN.md): LDS is double buffered.  Chunk c+1 is
// written into the other buffer at the START of chunk c's compute (the ds_writes
// drain under the MFMAs), chunk c+2's global loads are issued right after and stay
// in flight for a whole chunk, and there is ONE barrier per chunk.
//   NW = 8: 256 positions / workgroup, 1 workgroup per CU (2 waves per SIMD)
//   NW = 4: 128 positions / workgroup, 2 workgroups per CU (NARRE's short reviews)
// LDS per buffer (floats): X [32 NW + 2][24] + W [112][56]; both row strides are
// == 8 (mod 16), which makes the ds_read_b128 fragment reads (16 rows x 4 k-quads
// per wave) bank-conflict free -- SQ_LDS_BANK_CONFLICT measures 0.
//
// The ds_read_b128 trick: lane (row = l & 15, quad = l >> 4) reads 4 consecutive k
// of its row; register r of quad q then holds k = 4q + r.  MFMA k-slot q of step r
// therefore sees k = 4q + r for BOTH operands, a permutation of the 16 k's of the
// block -- the dot product does not care -- so one 16-byte LDS read feeds 4 MFMAs.
#include <stdlib.h>

#include <math.h>
#include "textcnn.h"
#include "wgrad_device.h"

namespace r4r {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = NP / 16;
constexpr int EC = 16;                 // embedding columns per K chunk
constexpr int XS = EC + 8;             // X row stride in LDS (floats) = 24
constexpr int WS = 3 * EC + 8;         // W row stride in LDS (floats) = 56
constexpr int W_VEC = NP * WS / 4;     // 1568 float4 per weight chunk

template <int NW>
struct Cfg {
    static constexpr int THREADS = 64 * NW;
    static constexpr int MTILE = 32 * NW;
    static constexpr int XROWS = MTILE + 2;
    static constexpr int BUF_FLOATS = XROWS * XS + NP * WS;
    static constexpr int LDS_BYTES = 2 * BUF_FLOATS * 4;
    static constexpr int XK = MTILE * 4 / THREADS;                  // 2 float4 of X per thread (+ halo)
    static constexpr int WK = (W_VEC + THREADS - 1) / THREADS;      // 4 (NW=8) or 7 (NW=4)
};

static inline int n_chunks(int E) { return (E + EC - 1) / EC; }

struct FwdArgs {
    FwdTower t[MAX_TOWERS];
    const float *table;
    int T, E, F, tiles, nchunk;
};

struct PackArgs {
    const float *w[MAX_TOWERS];
    float *wp[MAX_TOWERS];
    int E, F, nchunk;
};

// ---------------------------------------------------------------------------
// Pack conv weight [F][3][E] into the per-chunk LDS image [chunk][112][56]:
//   Wp[c][n][j*16 + ee] = W[n][j][c*16 + ee]   (0 for n >= F, e >= E, pad cols)
// so the kernel's weight staging is a linear float4 copy.  blockIdx.y = tower.
// ---------------------------------------------------------------------------
__global__ void textcnn_pack_w_kernel(PackArgs a) {
    const float *__restrict__ w = a.w[blockIdx.y];
    float *__restrict__ wp = a.wp[blockIdx.y];
    const int total = a.nchunk * NP * WS;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int col = i % WS;
        const int n = (i / WS) % NP;
        const int c = i / (WS * NP);
        float v = 0.f;
        if (col < 3 * EC && n < a.F) {
            const int j = col / EC, e = c * EC + col % EC;
            if (e < a.E) v = w[((size_t)n * 3 + j) * a.E + e];
        }
        wp[i] = v;
    }
}

// ---------------------------------------------------------------------------
// Forward tile kernel.  grid = (N * tiles, ntower); wave w owns conv positions
// [32w, 32w+32) x all 112 filters: 2 x 7 accumulators of 16x16.
// ---------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void textcnn_fwd_kernel(FwdArgs args) {
    using C = Cfg<NW>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lds = reinterpret_cast<float *>(smem);

    const FwdTower &tw = args.t[blockIdx.y];
    const float *__restrict__ table = args.table;
    const int64_t *__restrict__ idx = tw.idx;
    const float *__restrict__ wp = tw.wp;
    const int T = args.T, E = args.E, F = args.F, tiles = args.tiles, nchunk = args.nchunk;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int doc = blockIdx.x / tiles, tile = blockIdx.x - doc * tiles;
    const int p0 = tile * C::MTILE;
    const int P = T + 2;
    const int lrow = lane & 15, q = lane >> 4;

    // staging role of this thread: float4 column c4 (of 4) of rows (tid>>2) + (THREADS/4) k,
    // k < XK, plus -- for tid < 8 -- of the two halo rows MTILE, MTILE+1.  Row offsets into
    // the table in floats; -1 = a row outside the document (the conv's zero padding).
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
        const int e = c * EC + c4 * 4;
#pragma unroll
        for (int k = 0; k <= C::XK; ++k) {
            xr[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (xoff[k] >= 0 && e < E) xr[k] = *reinterpret_cast<const f32x4 *>(table + xoff[k] + e);
        }
        const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(wp + (size_t)c * NP * WS);
#pragma unroll
        for (int k = 0; k < C::WK; ++k) {
            const int i = tid + k * C::THREADS;
            if (i < W_VEC) wr[k] = wsrc[i];
        }
    };
    auto write_lds = [&](float *buf) {
        float *Xs = buf, *Wl = buf + C::XROWS * XS;
#pragma unroll
        for (int k = 0; k <= C::XK; ++k) {
            const int r = (k < C::XK) ? (tid >> 2) + (C::THREADS / 4) * k : C::MTILE + (tid >> 2);
            if (k < C::XK || tid < 8) *reinterpret_cast<f32x4 *>(Xs + r * XS + c4 * 4) = xr[k];
        }
#pragma unroll
        for (int k = 0; k < C::WK; ++k) {
            const int i = tid + k * C::THREADS;
            if (i < W_VEC) reinterpret_cast<f32x4 *>(Wl)[i] = wr[k];
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
        const float *cur = lds + (c & 1) * C::BUF_FLOATS;
        if (c + 1 < nchunk) {
            write_lds(lds + ((c + 1) & 1) * C::BUF_FLOATS);   // chunk c+1 -> the other buffer
            if (c + 2 < nchunk) issue_loads(c + 2);            // in flight for a whole chunk
        }
        const float *Xs = cur, *Wl = cur + C::XROWS * XS;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            f32x4 a[2], b[NT];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                a[mi] = *reinterpret_cast<const f32x4 *>(Xs + (wave * 32 + mi * 16 + lrow + j) * XS + q * 4);
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
                b[ni] = *reinterpret_cast<const f32x4 *>(Wl + (ni * 16 + lrow) * WS + j * EC + q * 4);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][kk], b[ni][kk], acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: (max, first argmax) over this tile's positions, per filter.
    // C layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg.
    float *redv = lds;                                    // [NW][NP]
    int *redp = reinterpret_cast<int *>(lds + NW * NP);   // [NW][NP]
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        const int col = ni * 16 + lrow;
        const float bc = (col < F) ? tw.conv_b[col] : 0.f;
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
        tw.pmax[(size_t)blockIdx.x * NP + tid] = best;
        tw.parg[(size_t)blockIdx.x * NP + tid] = bp;
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
// Argmax-sparse weight gradient.  The reference pays a dense wgrad GEMM
// (convolution_backward); because of the global max-pool only ONE window per
// (document, filter) carries gradient, so this is a gather-weighted sum:
//   dW[f, j, :] = sum_n g[n,f] * table[idx[n, argmax[n,f] - 2 + j], :]
// grid = (F, nsplit, ntower); block = 256 threads, each owning one float4 column
// group of the [3][E] window (looped if 3E/4 > 256).  Workgroup (f, s) sums its
// slice of documents; a second kernel adds the nsplit partials in a fixed order
// (deterministic, no atomics).
// ---------------------------------------------------------------------------
// (WgradArgs and the per-workgroup body live in wgrad_device.h, shared with engine.hip)

__global__ __launch_bounds__(WG_THREADS) void textcnn_wgrad_kernel(WgradArgs a) {
    wgrad_block(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

__global__ void textcnn_wgrad_reduce_kernel(WgradArgs a) { wgrad_reduce_block(a, blockIdx.y, blockIdx.x); }

// ----------------------------------------------------------------- launchers
size_t textcnn_wp_floats(int E) { return (size_t)n_chunks(E) * NP * WS; }

// R4R_TEXTCNN_TILE=128|256 pins the tile height for A/B runs.
int textcnn_tile_rows(int T) {
    static int pin = -1;
    if (pin < 0) {
        const char *e = getenv("R4R_TEXTCNN_TILE");
        pin = e ? atoi(e) : 0;
    }
    if (pin == 128 || pin == 256) return pin;
    return (T + 2 > 160) ? 256 : 128;      // long documents: 256-row tiles; short reviews: 128
}

int textcnn_tiles(int T) {
    const int m = textcnn_tile_rows(T);
    return (T + 2 + m - 1) / m;
}

// Which conv algorithm to run.  R4R_CONV_ALGO=direct|project pins it (A/B runs, tests).
// AUTO (measured crossover): project-then-gather wins at every batch size for
// wide windows (E >= 128: 0.080 vs 0.135 ms/step at B=2, E=300) and from ~64 k positions up
// for narrow ones (E=64: 0.092 vs 0.154 ms at B=128; 0.069 vs 0.063 ms at B=32).
int textcnn_pick_algo(int requested, int64_t N, int T, int E, int F) {
    static int pin = -1;
    if (pin < 0) {
        const char *e = getenv("R4R_CONV_ALGO");
        pin = !e ? 0 : (e[0] == 'd' ? R4R_CONV_DIRECT : (e[0] == 'p' ? R4R_CONV_PROJECT : 0));
    }
    int algo = pin ? pin : requested;
    if (algo != R4R_CONV_DIRECT && algo != R4R_CONV_PROJECT)
        algo = (E >= 128 || N * (int64_t)(T + 2) >= 65536) ? R4R_CONV_PROJECT : R4R_CONV_DIRECT;
    if (F != 100) algo = R4R_CONV_DIRECT;                   // the projection kernels are built for 100 filters
    return algo;
}

int textcnn_wgrad_splits(int64_t N) {
    // Documents per split: 16 = one phase-1 round of the wgrad workgroup.  Measured at B=128 (backward +
    // reduce kernels): 8 docs 14.0 + 8.3 us, 16 docs 11.0 + 5.1, 32 docs 14.6 + 4.9, 64 docs 25.2 + 4.4.
    // R4R_WGRAD_DOCS pins another value for A/B runs.
    static int docs = -1;
    if (docs < 0) {
        const char *e = getenv("R4R_WGRAD_DOCS");
        docs = e ? atoi(e) : 16;
        if (docs < 1) docs = 16;
    }
    static int cap = -1;
    if (cap < 0) {
        const char *e = getenv("R4R_WGRAD_MAX_SPLITS");
        cap = e ? atoi(e) : 16;
        if (cap < 1) cap = 16;
    }
    int s = (int)cdiv(N, docs);
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    return s;
}

int textcnn_fwd_launch(const float *table, const FwdTower *tw, int ntower,
                       int64_t N, int T, int E, int F, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(textcnn_fwd_kernel<8>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<8>::LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(textcnn_fwd_kernel<4>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<4>::LDS_BYTES);
        attr_set = true;
    }
    const int nchunk = n_chunks(E);
    PackArgs pa;
    FwdArgs fa;
    for (int k = 0; k < MAX_TOWERS; ++k) {
        const FwdTower &t = tw[k < ntower ? k : 0];
        pa.w[k] = t.conv_w;
        pa.wp[k] = t.wp;
        fa.t[k] = t;
    }
    pa.E = E; pa.F = F; pa.nchunk = nchunk;
    const int img = nchunk * NP * WS;
    textcnn_pack_w_kernel<<<dim3((img + 255) / 256, ntower), 256, 0, st>>>(pa);

    const int mtile = textcnn_tile_rows(T);
    fa.table = table; fa.T = T; fa.E = E; fa.F = F; fa.nchunk = nchunk;
    fa.tiles = (T + 2 + mtile - 1) / mtile;
    const dim3 grid((unsigned)(N * fa.tiles), ntower);
    {
        ScopedTiming tm(R4R_TIMING_TEXTCNN_FWD, st);
        if (mtile == 256)
            textcnn_fwd_kernel<8><<<grid, Cfg<8>::THREADS, Cfg<8>::LDS_BYTES, st>>>(fa);
        else
            textcnn_fwd_kernel<4><<<grid, Cfg<4>::THREADS, Cfg<4>::LDS_BYTES, st>>>(fa);
    }
    return check_launch("textcnn_fwd");
}

int textcnn_pool_finish_launch(const float *pmax, const int *parg, float *pooled, int *argmax,
                               int64_t N, int tiles, int F, hipStream_t st) {
    textcnn_pool_finish_kernel<<<(unsigned)cdiv(N * F, 256), 256, 0, st>>>(pmax, parg, pooled, argmax, N, F, tiles);
    return check_launch("textcnn_pool_finish");
}

int textcnn_wgrad_launch(const float *table, const WgradTower *tw, int ntower,
                         int64_t N, int T, int E, int F, hipStream_t st, int64_t table_bytes) {
    WgradArgs a;
    for (int k = 0; k < MAX_TOWERS; ++k) a.t[k] = tw[k < ntower ? k : 0];
    a.table = table; a.N = N; a.T = T; a.E = E; a.F = F;
    a.table_bytes = table_bytes;                            // (0: the wide form keeps its per-row loads)
    a.nsplit = textcnn_wgrad_splits(N);
    a.per_split = (int)cdiv(N > 0 ? N : 1, a.nsplit);
    {
        ScopedTiming tm(R4R_TIMING_TEXTCNN_WGRAD, st);
        textcnn_wgrad_kernel<<<dim3(F, a.nsplit, ntower), WG_THREADS, 0, st>>>(a);
    }
    return textcnn_wgrad_reduce_launch(tw, ntower, N, E, F, st);
}

int textcnn_wgrad_reduce_launch(const WgradTower *tw, int ntower, int64_t N, int E, int F, hipStream_t st) {
    WgradArgs a;
    for (int k = 0; k < MAX_TOWERS; ++k) a.t[k] = tw[k < ntower ? k : 0];
    a.table = nullptr; a.N = N; a.T = 0; a.E = E; a.F = F;
    a.nsplit = textcnn_wgrad_splits(N);
    a.per_split = (int)cdiv(N > 0 ? N : 1, a.nsplit);
    const int tot = F * 3 * E + F;
    textcnn_wgrad_reduce_kernel<<<dim3((tot + 255) / 256, ntower), 256, 0, st>>>(a);
    return check_launch("textcnn_wgrad");
}

}  // namespace r4r

using namespace r4r;

// Workspace: [weight image][pmax][parg] for the forward, [part_w][part_b] for the
// backward; sized for the smaller (128-row) tile so either tile height fits.
extern "C" size_t r4r_textcnn_ws_bytes(int64_t N, int T, int E, int F, int64_t V) {
    if (N < 0 || T <= 0 || E <= 0 || F <= 0 || V <= 0) return 0;
    const size_t tiles128 = (size_t)(T + 2 + 127) / 128;
    const size_t partials = 2 * align256((size_t)N * tiles128 * NP * 4);
    const size_t fwd = align256(textcnn_wp_floats(E) * 4) + partials;
    const size_t proj = partials + 2 * align256((size_t)(V + 4) * 4) + align256((size_t)proj_row_capacity(N, T, V) * 4) +
                        align256(256) + align256(proj_ptab_floats(N, T, V) * 4) + align256(textcnn_wp_floats(E) * 4);
    const int ns = textcnn_wgrad_splits(N);
    const size_t bwd = align256((size_t)ns * F * 3 * E * 4) + align256((size_t)ns * F * 4);
    size_t m = fwd > bwd ? fwd : bwd;
    return m > proj ? m : proj;
}

static int check_tower_args(const void *table, int64_t V, const void *idx, int64_t N, int T, int E, int F) {
    R4R_REQUIRE(table && idx, "textcnn: null table/idx");
    R4R_REQUIRE(V > 0 && N >= 0 && T > 0, "textcnn: bad sizes V=%lld N=%lld T=%d", (long long)V, (long long)N, T);
    R4R_REQUIRE(E > 0 && E % 4 == 0, "textcnn: word_embed_size %d must be a positive multiple of 4 "
                                     "(pad the frozen table on the host otherwise)", E);
    R4R_REQUIRE(F > 0 && F <= NP, "textcnn: %d filters > %d supported", F, NP);
    R4R_REQUIRE(N * (int64_t)((T + 2 + 127) / 128) < (1ll << 31), "textcnn: grid too large");
    return R4R_OK;
}

extern "C" int r4r_textcnn_fwd(const float *table, int64_t V, const int64_t *idx,
                               const float *conv_w, const float *conv_b,
                               float *pooled, int32_t *argmax,
                               void *ws, size_t ws_bytes,
                               int64_t N, int T, int E, int F, void *stream) {
    if (int rc = check_tower_args(table, V, idx, N, T, E, F)) return rc;
    R4R_REQUIRE(conv_w && conv_b && pooled && argmax && ws, "textcnn_fwd: null pointer");
    if (ws_bytes < r4r_textcnn_ws_bytes(N, T, E, F, V)) {
        set_error("textcnn_fwd: workspace %zu < %zu bytes", ws_bytes, r4r_textcnn_ws_bytes(N, T, E, F, V));
        return R4R_ERR_WORKSPACE;
    }
    if (N == 0) return R4R_OK;
    hipStream_t st = as_stream(stream);
    char *base = static_cast<char *>(ws);
    auto take = [&](size_t nbytes) { char *r = base; base += align256(nbytes); return r; };
    const size_t tiles128 = (size_t)(T + 2 + 127) / 128;
    float *pmax = reinterpret_cast<float *>(take((size_t)N * tiles128 * NP * 4));
    int *parg = reinterpret_cast<int *>(take((size_t)N * tiles128 * NP * 4));
    if (textcnn_pick_algo(R4R_CONV_AUTO, N, T, E, F) == R4R_CONV_PROJECT) {
        ProjTower pt;
        pt.idx = idx; pt.conv_w = conv_w; pt.conv_b = conv_b; pt.pmax = pmax; pt.parg = parg;
        pt.flags = reinterpret_cast<int *>(take((size_t)(V + 4) * 4));
        pt.slot = reinterpret_cast<int *>(take((size_t)(V + 4) * 4));
        pt.list = reinterpret_cast<int *>(take((size_t)proj_row_capacity(N, T, V) * 4));
        pt.count = reinterpret_cast<int *>(take(256));
        pt.ptab = reinterpret_cast<float *>(take(proj_ptab_floats(N, T, V) * 4));
        // scratch of the opt-in fp16-split GEMM: handed over only in mode 2 (the caller vouches that the scales given
        // to r4r_gemm_math describe THIS table and these weights; mode 1 is for the fused steps, whose host refreshes them)
        float *wimg = reinterpret_cast<float *>(take(textcnn_wp_floats(E) * 4));
        pt.wimg = proj_gemm_math_mode() == 2 ? wimg : nullptr;
        if (int rc = textcnn_proj_fwd_launch(table, V, &pt, 1, N, T, E, F, /*zero_state=*/true, st)) return rc;
        return textcnn_pool_finish_launch(pmax, parg, pooled, argmax, N, proj_tiles(T), F, st);
    }
    FwdTower tw;
    tw.idx = idx; tw.conv_w = conv_w; tw.conv_b = conv_b; tw.pmax = pmax; tw.parg = parg;
    tw.wp = reinterpret_cast<float *>(take(textcnn_wp_floats(E) * 4));
    if (int rc = textcnn_fwd_launch(table, &tw, 1, N, T, E, F, st)) return rc;
    return textcnn_pool_finish_launch(pmax, parg, pooled, argmax, N, textcnn_tiles(T), F, st);
}

// Which algorithm a request runs (the static rule above + the R4R_CONV_ALGO pin): hosts that keep
// per-algorithm state (prepared token buffers) ask instead of restating the rule.
extern "C" int r4r_conv_algo(int requested, int64_t N, int T, int E, int F) {
    return textcnn_pick_algo(requested, N, T, E, F);
}

// Measured rule: given what a batch actually holds -- `rows` distinct tokens (summed over the towers)
// in `docs` documents of T words -- is the projection (GEMM over the distinct rows + gather-add-max +
// token compaction over V words) or the direct conv (every position) the faster forward?  Cost model
// fitted to MI355X measurements at E = 64 and 300 (DESIGN.md 4.1c lists them, profiles/r02_conv_rule.txt):
//   direct      0.12 + 0.0051 E ns per position, positions rounded up to 128 per document and to launches'
//               waves of 65,536 positions (B = 8 .. 32 at T = 1000 all take the 110 us of one wave at E = 300)
//   GEMM        (11 + 0.14 E) us per round of 256 row tiles of 128 rows; the last, partial round (9.3 + 0.044 E) /
//               (13.8 + 0.071 E) us when it is cut into 4 / 2 column parts (at most a quarter / half of the grid:
//               project.hip; fitted at E = 64 and 300: 12 / 18 and 22.5 / 35 us)
//   gather      0.085 ns + 0.0006 ns per MB of projected rows (1200 B each), per position; at least 10 us
//   tokens      4 us + 0.027 us per 1000 words of vocabulary
extern "C" int r4r_conv_pick(int E, int T, int64_t docs, int64_t rows, int64_t V) {
    if (E <= 0 || T <= 0 || docs <= 0 || rows <= 0) return R4R_CONV_PROJECT;
    const double P = (double)(T + 2);
    const double padded = (double)docs * (double)(((int64_t)P + 127) / 128 * 128);
    const double t_direct = ceil(padded / 65536.0) * 65536.0 * (0.12 + 0.0051 * E) * 1e-3;   // us
    const int64_t tiles = (rows + 127) / 128 + 1, tail = tiles % 256;
    const double round = 11.0 + 0.14 * E;
    const double last = tail == 0 ? 0.0 : (tail * 4 <= 256 ? 9.3 + 0.044 * E : (tail * 2 <= 256 ? 13.8 + 0.071 * E : round));
    const double t_gemm = (double)(tiles / 256) * round + last;
    double t_gather = (double)docs * P * (0.085 + 0.0006 * ((double)rows * 1200.0 / 1e6)) * 1e-3;
    if (t_gather < 10.0) t_gather = 10.0;                                                  // a launch's latency floor
    const double t_tokens = 4.0 + 0.027 * ((double)V / 1000.0);
    return (t_gemm + t_gather + t_tokens < t_direct) ? R4R_CONV_PROJECT : R4R_CONV_DIRECT;
}

extern "C" int r4r_gemm_math(int mode, float table_maxabs, float weight_maxabs) {
    R4R_REQUIRE(mode >= 0 && mode <= 2, "gemm_math: mode %d (0 fp32, 1 f16x2 in the fused steps, 2 also in r4r_textcnn_fwd)", mode);
    R4R_REQUIRE(mode == 0 || (table_maxabs > 0.f && table_maxabs < 3.0e38f && weight_maxabs > 0.f && weight_maxabs < 3.0e38f),
                "gemm_math: f16x2 needs max |table| > 0 and max |conv weights| > 0");
    proj_gemm_set_math(mode, table_maxabs, weight_maxabs);
    return R4R_OK;
}

extern "C" int r4r_gemm_form(int balanced) {
    proj_gemm_set_form(balanced);
    return R4R_OK;
}

extern "C" int r4r_textcnn_wgrad(const float *table, int64_t V, const int64_t *idx,
                                 const float *g_pooled, const int32_t *argmax,
                                 float *d_conv_w, float *d_conv_b,
                                 void *ws, size_t ws_bytes,
                                 int64_t N, int T, int E, int F, void *stream) {
    if (int rc = check_tower_args(table, V, idx, N, T, E, F)) return rc;
    R4R_REQUIRE(g_pooled && argmax && d_conv_w && d_conv_b && ws, "textcnn_wgrad: null pointer");
    if (ws_bytes < r4r_textcnn_ws_bytes(N, T, E, F, V)) {
        set_error("textcnn_wgrad: workspace %zu < %zu bytes", ws_bytes, r4r_textcnn_ws_bytes(N, T, E, F, V));
        return R4R_ERR_WORKSPACE;
    }
    const int ns = textcnn_wgrad_splits(N);
    char *base = static_cast<char *>(ws);
    WgradTower tw;
    tw.idx = idx; tw.g_pooled = g_pooled; tw.argmax = argmax; tw.d_w = d_conv_w; tw.d_b = d_conv_b;
    tw.part_w = reinterpret_cast<float *>(base);
    base += align256((size_t)ns * F * 3 * E * 4);
    tw.part_b = reinterpret_cast<float *>(base);
    // R4R_WGRAD_ROWS=loop: the wide form's per-row loads (what tables of 4 GB and more get) instead of the buffer-resource
    // batches -- for the test that holds the two to the same bits
    const char *rows = getenv("R4R_WGRAD_ROWS");
    const bool loop = rows && rows[0] == 'l';
    return textcnn_wgrad_launch(table, &tw, 1, N, T, E, F, as_stream(stream), loop ? 0 : (int64_t)V * E * 4);
}
