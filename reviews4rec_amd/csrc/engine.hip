// Fused DeepCoNN training / inference step for gfx950 ('deepconn' mode: two TextCNN
// towers -> FC -> dropout -> concat -> FM + global bias -> squared error).
//
// One C call enqueues the whole step -- 6 launches instead of the ~45 of the
// op-by-op autograd path:
//   1 pack both conv weight images            (textcnn.hip)
//   2 conv + relu + max-pool, BOTH towers in one grid   (textcnn.hip, fp32 MFMA)
//   3 head: pool-finish + FC + dropout + FM + SE, and the head's backward down to
//     g_pooled, one wave per rating            (here)
//   4 head parameter gradients, reduced over the batch in a fixed order (here)
//   5 argmax-sparse conv wgrad, both towers    (textcnn.hip)
//   6 wgrad partial reduce -> flat gradient buffer
// All trainable parameters live in ONE flat fp32 buffer (layout below) and so do the
// gradients: data parallelism is a single RCCL all-reduce over `flat_g`, and Adam is
// a single r4r_adam_multi sweep over one "tensor".
//
// Reference behaviour restated (file:line under the reference root):
//   DeepCoNN.forward                     DeepCoNN.py:37-66
//   TextCNN.forward (fc + dropout)       common_pytorch_models.py:33-37
//   TorchFM.forward                      common_pytorch_models.py:49-57
//   MSELoss + mean + backward            loss.py:7-11, main.py:56-59
// 'deepconn' mode never touches `final`, `user_bias`, `item_bias` (DeepCoNN.py:64-66,
// SURVEY.md fact 7), so they are not in the flat buffer and Adam never sees them.
#include <stdlib.h>

#include "trace_device.h"
#include "textcnn.h"
#include "wgrad_device.h"
#include "tokens_device.h"
#include "adam_device.h"

namespace r4r {

constexpr int F_CONV = 100;     // common_pytorch_models.py:11
constexpr int FM_K = 8;         // DeepCoNN.py:32
constexpr int MAX_L = 128;         // (the FM takes 2 L inputs: ceil(2 L / 64) per lane of its wave)

enum { P_UCW = 0, P_UCB, P_UFW, P_UFB, P_ICW, P_ICB, P_IFW, P_IFB, P_FMV, P_FMLW, P_FMLB, P_GB, P_COUNT };

struct Layout {
    int64_t off[P_COUNT], size[P_COUNT], total;
};

static Layout make_layout(int E, int L) {
    Layout lay;
    const int64_t sz[P_COUNT] = {(int64_t)F_CONV * 3 * E, F_CONV, (int64_t)L * F_CONV, L,
                                 (int64_t)F_CONV * 3 * E, F_CONV, (int64_t)L * F_CONV, L,
                                 (int64_t)2 * L * FM_K, 2 * L, 1, 1};
    int64_t o = 0;
    for (int i = 0; i < P_COUNT; ++i) {
        lay.off[i] = o;
        lay.size[i] = sz[i];
        o += (sz[i] + 31) & ~(int64_t)31;          // 128-byte aligned slots (whole cache lines for the GEMM's weight-row pieces); pad floats stay 0 forever
    }
    lay.total = o;
    return lay;
}

struct HeadArgs {
    // inputs
    const float *pmax[2]; const int *parg[2];     // conv partials per tower [B, tiles, NP]
    const float *fc_w[2]; const float *fc_b[2];   // [L,100], [L]
    const float *V, *lin_w, *lin_b, *gbias;       // [2L,8], [2L], [1], [1]
    const float *y;                               // [B] or NULL
    // outputs
    float *pooled[2]; int *argmax[2];             // [B,100]
    float *g_pooled[2];                           // [B,100]   (training only)
    float *mult;                                  // [B, 2L] dropout multipliers
    float *x, *s, *g, *gz;                        // [B,2L], [B,8], [B], [B,2L]
    float *pred, *se;                             // [B]
    int64_t B;
    int L, tiles, training, want_grad;
    float p_drop, inv_denom;
    uint64_t seed, offset;
};


constexpr int HEAD_MAX_TILES = 8;

// ---- the head, ONE WORKGROUP PER RATING.  Wave w finishes the pool of (tower w >> 1, filters 64 (w & 1) ..): 16
// partial loads per lane, every load of the prologue requested before anything waits; the FC layers' 2 L x 100
// products are spread over all 256 threads (PARTS threads per output, summed in a fixed order through LDS:
// deterministic); wave 0 alone runs the FM's cross-lane sums; the backward to g_pooled is one output per thread.
// (A four-ratings-per-workgroup form put a batch of 128 on 32 CUs of 256: 11 us against 8; removed in round 5.)
// ML: compile-time cap of the latent size (LDS arrays, register arrays and unrolled staging loops are sized by it).
// (a pointer array of the kernel arguments indexed by a run-time tower makes every lane FETCH the pointer from the
// argument segment, a round trip in front of the access it serves: select between the constant-index elements instead)
#define SEL2(arr, s) ((s) ? (arr)[1] : (arr)[0])
HEAD_TRACE_DEFINE(r4r_debug_dc_head_trace)
template <int ML>
__global__ __launch_bounds__(256) void deepconn_head_wg_kernel(HeadArgs a) {
    HEAD_STAMP(0)
    constexpr int N2 = 2 * ML, PARTS = 256 / N2;      // FC outputs (both towers); threads per output: 8 | 4
    __shared__ float sw[2][ML][F_CONV + 1];            // FC weights, +1 pad
    __shared__ float sfb[N2], slw[N2], sz[N2];
    __shared__ float sV[N2][FM_K];
    __shared__ float sp[2][F_CONV + 4];                 // pooled
    __shared__ float red[N2][PARTS + 1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t b = blockIdx.x;
    const int L = a.L, n = 2 * L;
    const float lin_b0 = a.lin_b[0], gbias0 = a.gbias[0], yb = a.y ? a.y[b] : 0.f;

    // every load of the prologue is issued before anything waits: the FC matrices, FM V / lin, and this lane's partials
    // (the FC matrices as 16-byte units, F_CONV / 4 = 25 filter quads per row: a quarter of the loads and index computations)
    constexpr int FQ = F_CONV / 4;
    typedef float hq4 __attribute__((ext_vector_type(4)));
    constexpr int WREGS = (2 * ML * FQ + 255) / 256;
    hq4 wreg[WREGS];
    const int wtot = 2 * L * FQ;
#pragma unroll
    for (int k = 0; k < WREGS; ++k) {
        // (unconditional, clamped: behind a uniform `if (256 k < wtot)` hipcc put a branch and a full vmcnt(0) around every
        // one of these loads -- four dependent round trips in front of the launch's first barrier)
        const int i = min(tid + 256 * k, wtot - 1);
        const int t = i >= L * FQ;
        wreg[k] = *reinterpret_cast<const hq4 *>(SEL2(a.fc_w, t) + 4 * (i - t * L * FQ));
    }
    const int tn = min(tid, n - 1);
    const float fbreg = SEL2(a.fc_b, tn >= L)[tn - (tn >= L ? L : 0)];
    const float lwreg = a.lin_w[tn];
    constexpr int VREGS = (N2 * FM_K + 255) / 256;
    float vreg[VREGS];
#pragma unroll
    for (int k = 0; k < VREGS; ++k) vreg[k] = a.V[min(tid + 256 * k, n * FM_K - 1)];

    // ---- pool finish of (tower t, filter f): max over the tiles, relu, first argmax
    {
        const int t = w >> 1, f = lane + 64 * (w & 1), fc = min(f, F_CONV - 1);
        float best = -INFINITY;
        int bp = -1;
        if (a.tiles <= HEAD_MAX_TILES) {
            float v[HEAD_MAX_TILES];
            int pp[HEAD_MAX_TILES];
#pragma unroll
            for (int k = 0; k < HEAD_MAX_TILES; ++k) {
                const size_t o = ((size_t)b * a.tiles + min(k, a.tiles - 1)) * NP + fc;
                v[k] = SEL2(a.pmax, t)[o];
                pp[k] = SEL2(a.parg, t)[o];
            }
#pragma unroll
            for (int k = 0; k < HEAD_MAX_TILES; ++k)           // a clamped duplicate never wins (strict >)
                if (v[k] > best) { best = v[k]; bp = pp[k]; }
        } else {
            for (int k0 = 0; k0 < a.tiles; k0 += HEAD_MAX_TILES) {
                float v[HEAD_MAX_TILES];
                int pp[HEAD_MAX_TILES];
#pragma unroll
                for (int k = 0; k < HEAD_MAX_TILES; ++k) {
                    const bool in = k0 + k < a.tiles;
                    const size_t o = ((size_t)b * a.tiles + (in ? k0 + k : 0)) * NP + fc;
                    v[k] = in ? SEL2(a.pmax, t)[o] : -INFINITY;
                    pp[k] = SEL2(a.parg, t)[o];
                }
#pragma unroll
                for (int k = 0; k < HEAD_MAX_TILES; ++k)
                    if (v[k] > best) { best = v[k]; bp = pp[k]; }
            }
        }
        if (!(best > 0.f)) { best = 0.f; bp = -1; }
        if (f < F_CONV) {
            sp[t][f] = best;
            SEL2(a.pooled, t)[b * F_CONV + f] = best;
            SEL2(a.argmax, t)[b * F_CONV + f] = bp;
        }
    }
    HEAD_STAMP(1)
#pragma unroll
    for (int k = 0; k < WREGS; ++k) {
        const int i = tid + 256 * k;
        if (i < wtot) {
            const int t = i >= L * FQ, r = i - t * L * FQ, l = r / FQ, f0 = 4 * (r - l * FQ);
#pragma unroll
            for (int c = 0; c < 4; ++c) sw[t][l][f0 + c] = wreg[k][c];          // (rows of F_CONV + 1 floats: four 4-byte writes)
        }
    }
    if (tid < n) { sfb[tid] = fbreg; slw[tid] = lwreg; }
#pragma unroll
    for (int k = 0; k < VREGS; ++k) {
        const int i = tid + 256 * k;
        if (i < n * FM_K) sV[i / FM_K][i % FM_K] = vreg[k];
    }
    __syncthreads();
    HEAD_STAMP(2)

    // ---- FC: output o = tid / PARTS (tower o / L, row o % L), PARTS partial sums of ~100 / PARTS products each
    {
        const int o = tid / PARTS, part = tid - o * PARTS;
        if (o < n) {
            const int t = o / L, l = o - t * L;
            float acc = 0.f;
            for (int f = part; f < F_CONV; f += PARTS) acc = fmaf(sp[t][f], sw[t][l][f], acc);
            red[o][part] = acc;
        }
    }
    __syncthreads();
    const bool has_y = a.y != nullptr, want = has_y && a.want_grad;     // uniform across the grid
    if (w == 0) {
        // FM input i = lane + 64 u (NI inputs per lane: 2 at latent_size 33 .. 64, 4 at 65 .. 128)
        constexpr int NI = (N2 + 63) / 64;
        float xi[NI], mult[NI], lw[NI], gacc[NI];
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = lane + 64 * u;
            xi[u] = 0.f; mult[u] = 1.f; gacc[u] = 0.f;
            lw[u] = (i < n) ? slw[i] : 0.f;
            if (i < n) {
                float acc = 0.f;
#pragma unroll
                for (int q = 0; q < PARTS; ++q) acc += red[i][q];      // fixed order
                xi[u] = acc + sfb[i];
            }
            // ---- dropout on the FC output (common_pytorch_models.py:37)
            if (a.training && a.p_drop > 0.f && i < n) {
                const uint32_t r = philox_first_word(a.offset + (uint64_t)(b * n + i), a.seed);
                const float uu = (float)(r >> 8) * (1.0f / 16777216.0f);
                mult[u] = (uu >= a.p_drop) ? 1.f / (1.f - a.p_drop) : 0.f;
            }
            xi[u] *= mult[u];
        }
        // ---- FM (common_pytorch_models.py:49-57) + global bias
        float inter = 0.f;
        float sk_keep[FM_K];
#pragma unroll
        for (int k = 0; k < FM_K; ++k) {
            float pv = 0.f, pv2 = 0.f, vv[NI];
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                const int i = lane + 64 * u;
                vv[u] = (i < n) ? sV[i][k] : 0.f;
                pv += xi[u] * vv[u];
                pv2 += xi[u] * xi[u] * vv[u] * vv[u];
            }
            const float sk = wave_sum(pv);
            const float s2 = wave_sum(pv2);
            inter += sk * sk - s2;
#pragma unroll
            for (int u = 0; u < NI; ++u) gacc[u] += sk * vv[u] - xi[u] * vv[u] * vv[u];
            sk_keep[k] = sk;
        }
        float pl = 0.f;
#pragma unroll
        for (int u = 0; u < NI; ++u) pl += xi[u] * lw[u];
        const float lin = wave_sum(pl);
        const float pred = (0.5f * inter + lin + lin_b0) + gbias0;
        if (lane == 0) a.pred[b] = pred;
        if (has_y) {
            const float d = pred - yb;
            if (lane == 0) a.se[b] = d * d;
            if (want) {
                // ---- backward of the head down to gz
                const float g = 2.f * d * a.inv_denom;                 // d mean(SE) / d pred
#pragma unroll
                for (int u = 0; u < NI; ++u) {
                    const int i = lane + 64 * u;
                    const float gx = g * (gacc[u] + lw[u]);            // d / d x_i
                    const float gz = gx * mult[u];                     // through dropout
                    if (i < n) {
                        sz[i] = gz;
                        a.x[b * n + i] = xi[u];
                        a.gz[b * n + i] = gz;
                        a.mult[b * n + i] = mult[u];
                    }
                }
                if (lane < FM_K) {
                    float sv = 0.f;
#pragma unroll
                    for (int k = 0; k < FM_K; ++k) if (lane == k) sv = sk_keep[k];
                    a.s[b * FM_K + lane] = sv;
                }
                if (lane == 0) a.g[b] = g;
            }
        }
    }
    if (!want) return;                                     // uniform across the grid
    __syncthreads();
    HEAD_STAMP(3)
    if (tid < 2 * F_CONV) {
        const int t = tid >= F_CONV, f = tid - t * F_CONV;
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc = fmaf(sz[t * L + l], sw[t][l][f], acc);
        SEL2(a.g_pooled, t)[b * F_CONV + f] = acc;
    }
    HEAD_STAMP(4)
}

struct HeadGradArgs {
    const float *pooled[2];      // [B,100]
    const float *x, *s, *g, *gz; // [B,2L], [B,8], [B], [B,2L]
    const float *V;              // [2L,8]
    const float *se;             // [B]
    float *g_fc_w[2], *g_fc_b[2], *g_V, *g_lin_w, *g_lin_b, *g_gb;
    float *sse_accum;            // nullable: += sum_b se[b]
    int64_t B;
    int L;
};

// Output element o of the concatenated head-gradient vector, reduced over the batch by
// HG_ROWS row groups of one 256-thread workgroup (fixed order -> deterministic).  `blk`
// selects the HG_COLS outputs this workgroup owns: few outputs x many row groups keeps each
// thread's serial chain short (B / 16 terms).
constexpr int HG_ROWS = 16, HG_COLS = 16;
__device__ __forceinline__ void head_grad_block(const HeadGradArgs &a, int blk) {
    __shared__ float red[HG_ROWS][HG_COLS];
    const int ox = threadIdx.x & (HG_COLS - 1), rg = threadIdx.x / HG_COLS;
    const int L = a.L, n = 2 * L;
    const int n_fcw = L * F_CONV;
    const int seg[9] = {n_fcw, L, n_fcw, L, n * FM_K, n, 1, 1, 1};   // last: sse accumulator
    int o = blk * HG_COLS + ox;
    int which = -1, local = 0, acc_o = o;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        if (which < 0) {
            if (acc_o < seg[k]) { which = k; local = acc_o; }
            else acc_o -= seg[k];
        }
    }
    // Every output is a batch sum of  A[b] * (Bf[b] * C[b] - C[b] * C[b] * cv)  with absent factors 1 and cv 0:
    // the operand addresses are picked ONCE, and the loop requests HB terms' operands before it adds the first
    // (with the switch inside the loop hipcc kept the loads behind each other: 8 dependent round trips to data
    // another XCD wrote a moment ago -- the launch's longest chain, 9.4 us of tools/head_trace.py --backward).
    // Same ascending-b order per output as before.
    float s = 0.f;
    if (which >= 0) {
        // (plain selects, no switch: hipcc's lowering of a divergent switch over these pointer assignments lost the
        // first operand of the `item fc bias` outputs -- found by the golden trajectory test)
        const bool is_w = which == 0 || which == 2, is_b = which == 1 || which == 3, is_v = which == 4, is_lw = which == 5;
        const int t = which >> 1;                            // tower of an FC output
        const int wl = local / F_CONV, wf = local - wl * F_CONV;     // FC weight (l, f)
        const int vi = local / FM_K, vk = local - vi * FM_K;         // FM V (i, k)
        const float *pa = (is_w || is_b) ? a.gz : (which == 8 ? a.se : a.g);
        const int64_t sa = (is_w || is_b) ? n : 1;
        const int64_t oa = is_w ? t * L + wl : (is_b ? t * L + local : 0);
        const bool useb = is_w || is_v, usec = is_v || is_lw;
        const float *pb = is_w ? a.pooled[t & 1] : (is_v ? a.s : a.g);          // (dummies: any readable [B] array)
        const int64_t sb = is_w ? F_CONV : (is_v ? FM_K : 1), ob = is_w ? wf : (is_v ? vk : 0);
        const float *pc = usec ? a.x : a.g;
        const int64_t sc = usec ? n : 1, oc = is_v ? vi : (is_lw ? local : 0);
        const float cv = is_v ? a.V[local] : 0.f;
        constexpr int HB = 8;
        for (int64_t b0 = rg; b0 < a.B; b0 += (int64_t)HG_ROWS * HB) {
            float va[HB], vb[HB], vc[HB];
#pragma unroll
            for (int k = 0; k < HB; ++k) {
                const int64_t b = min(b0 + (int64_t)k * HG_ROWS, a.B - 1);      // clamped: never added past the end
                va[k] = pa[b * sa + oa];
                vb[k] = pb[b * sb + ob];
                vc[k] = pc[b * sc + oc];
            }
#pragma unroll
            for (int k = 0; k < HB; ++k) {
                const float B_ = useb ? vb[k] : 1.f, C_ = usec ? vc[k] : 1.f;
                const float term = va[k] * (B_ * C_ - C_ * C_ * cv);
                if (b0 + (int64_t)k * HG_ROWS < a.B) s += term;
            }
        }
    }
    red[rg][ox] = s;
    __syncthreads();
    if (rg == 0 && which >= 0) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < HG_ROWS; ++r) t += red[r][ox];
        switch (which) {
            case 0: a.g_fc_w[0][local] = t; break;
            case 1: a.g_fc_b[0][local] = t; break;
            case 2: a.g_fc_w[1][local] = t; break;
            case 3: a.g_fc_b[1][local] = t; break;
            case 4: a.g_V[local] = t; break;
            case 5: a.g_lin_w[local] = t; break;
            case 6: a.g_lin_b[0] = t; break;
            case 7: a.g_gb[0] = t; break;
            default: if (a.sse_accum) a.sse_accum[0] += t; break;
        }
    }
}

// The backward's two independent halves in ONE launch ("horizontal fusion"): z < 2 -> the
// argmax-sparse conv wgrad of tower z; z == 2 -> the head parameter gradients (strided
// over the first workgroups of that slice; the rest exit).  Neither depends on the other, both
// are latency-bound, so they overlap instead of running back to back.
// With a 4th z-slice the launch also marks the tokens of the NEXT batch (`nx`): that work depends
// only on the next batch's indices, the slice's workgroups fill CUs the latency-bound wgrad
// leaves idle, and the next step then starts at its GEMM.
BWD_TRACE_DEFINE(r4r_debug_dc_bwd_trace)
struct DcBackwardArgs {
    WgradArgs w;
    HeadGradArgs h;
    int hg_blocks;
    TokenArgs nx;
};
__global__ __launch_bounds__(WG_THREADS, 8) void deepconn_backward_kernel(DcBackwardArgs) {
    const DcBackwardArgs &A = kernel_args<DcBackwardArgs>();   // (each role's fields loaded in its own branch: common.h)
    const WgradArgs &w = A.w;
    const HeadGradArgs &h = A.h;
    const TokenArgs &nx = A.nx;
    const int hg_blocks = A.hg_blocks;
    BWD_STAMP(0, wall_clock64())
    BWD_STAMP(2, (unsigned long long)blockIdx.z + 1)
    // (slice 0 = the head gradients: its ~140 working workgroups are the longest chains of the launch and
    // start first; 1,600 wgrad workgroups follow, all resident at 8 waves per SIMD)
    if (blockIdx.z == 1 || blockIdx.z == 2) {
        wgrad_block(w, blockIdx.x, blockIdx.y, blockIdx.z - 1);
    } else if (blockIdx.z == 0) {
        for (int blk = blockIdx.y * gridDim.x + blockIdx.x; blk < hg_blocks; blk += gridDim.x * gridDim.y) {
            head_grad_block(h, blk);
            __syncthreads();
        }
    } else {
        token_mark_block(nx, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y, WG_THREADS);
    }
#ifdef R4R_TRACE
    __syncthreads();                                        // the workgroup's end, not thread 0's
#endif
    BWD_STAMP(1, wall_clock64())
}

// Second stage of the wgrad (fixed-order sum of the partials) and, in the extra workgroups, the
// compaction of the next batch's token marks.
// With `opt.on` the launch is also the optimiser: a conv-gradient element is updated as soon as
// it is summed, and a third group of workgroups updates the head parameters (their gradients
// were finished by the backward launch) -- one launch less per step, same Adam arithmetic as
// adam.hip.  Not under data parallelism: the all-reduce sits between gradient and update there.
constexpr int RED_THREADS = 256;
struct FusedAdam {
    float *p, *m, *v;            // flat parameter / moment buffers (layout of flat_g)
    const float *g;              // flat_g
    int64_t lo[2], hi[2];        // the two runs of non-conv parameters in the flat layout
    AdamScalars s;
    int on;
};
__global__ __launch_bounds__(RED_THREADS) void deepconn_reduce_kernel(WgradArgs w, int red_blocks, int comp_blocks,
                                                                      TokenArgs nx, FusedAdam opt) {
    const int bx = blockIdx.x;
    if (bx < red_blocks) {
        // the element's parameter and moments are requested WITH its partials (their addresses need nothing
        // from them), and Adam takes the sum from the register, not back from memory: one round trip, not three
        const WgradTower &tw = w.t[blockIdx.y];
        const int nw = w.F * 3 * w.E;
        const int i = bx * RED_THREADS + threadIdx.x;
        const float *gp0 = i < nw ? tw.d_w + i : (i < nw + w.F ? tw.d_b + (i - nw) : nullptr);
        const int64_t o = gp0 ? gp0 - opt.g : 0;
        float P = 0.f, M = 0.f, V = 0.f;
        if (opt.on && gp0) { P = opt.p[o]; M = opt.m[o]; V = opt.v[o]; }
        float *dst;
        const float g = wgrad_reduce_elem(w, blockIdx.y, i, dst);
        if (opt.on && dst) {
            adam_elem(P, g, M, V, opt.s);
            opt.p[o] = P; opt.m[o] = M; opt.v[o] = V;
        }
    } else if (bx < red_blocks + comp_blocks) {
        token_compact_auto<RED_THREADS / 64>(nx.t[blockIdx.y], nx.V, bx - red_blocks);
    } else {
        const int64_t o = opt.lo[blockIdx.y] + (int64_t)(bx - red_blocks - comp_blocks) * RED_THREADS + threadIdx.x;
        if (o < opt.hi[blockIdx.y]) {
            float P = opt.p[o], M = opt.m[o], V = opt.v[o];
            adam_elem(P, opt.g[o], M, V, opt.s);
            opt.p[o] = P; opt.m[o] = M; opt.v[o] = V;
        }
    }
}

__global__ void sse_only_kernel(const float *__restrict__ se, float *__restrict__ accum, int64_t B) {
    __shared__ float red[256];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < B; i += 256) s += se[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) accum[0] += red[0];
}

struct StepWs {
    float *wp[2], *pmax[2]; int *parg[2];
    // project-then-gather: token state is double buffered ([buffer][tower]) so batch k+1's
    // compaction can run while step k computes; the projected rows are per tower
    int *flags[2][2], *slot[2][2], *list[2][2], *count[2][2]; float *ptab[2];
    float *pooled[2]; int *argmax[2]; float *g_pooled[2];
    float *mult, *x, *s, *g, *gz;
    float *part_w[2], *part_b[2];
    size_t bytes;
};

static StepWs carve(void *ws, int64_t B, int T, int E, int L, int64_t V) {
    StepWs w;
    char *p = static_cast<char *>(ws);
    auto take = [&](size_t nbytes) { char *r = p; p += align256(nbytes); return r; };
    const size_t tiles128 = (size_t)(T + 2 + 127) / 128;
    const int ns = textcnn_wgrad_splits(B);
    for (int t = 0; t < 2; ++t) {
        w.wp[t] = reinterpret_cast<float *>(take(textcnn_wp_floats(E) * 4));
        w.pmax[t] = reinterpret_cast<float *>(take((size_t)B * tiles128 * NP * 4));
        w.parg[t] = reinterpret_cast<int *>(take((size_t)B * tiles128 * NP * 4));
        w.pooled[t] = reinterpret_cast<float *>(take((size_t)B * F_CONV * 4));
        w.argmax[t] = reinterpret_cast<int *>(take((size_t)B * F_CONV * 4));
        w.g_pooled[t] = reinterpret_cast<float *>(take((size_t)B * F_CONV * 4));
        w.part_w[t] = reinterpret_cast<float *>(take((size_t)ns * F_CONV * 3 * E * 4));
        w.part_b[t] = reinterpret_cast<float *>(take((size_t)ns * F_CONV * 4));
        for (int bf = 0; bf < 2; ++bf) {
            w.flags[bf][t] = reinterpret_cast<int *>(take((size_t)(V + 4) * 4));
            w.slot[bf][t] = reinterpret_cast<int *>(take((size_t)(V + 4) * 4));
            w.list[bf][t] = reinterpret_cast<int *>(take((size_t)proj_row_capacity(B, T, V) * 4));
            w.count[bf][t] = reinterpret_cast<int *>(take(256));
        }
        w.ptab[t] = reinterpret_cast<float *>(take(proj_ptab_floats(B, T, V) * 4));
    }
    w.mult = reinterpret_cast<float *>(take((size_t)B * 2 * L * 4));
    w.x = reinterpret_cast<float *>(take((size_t)B * 2 * L * 4));
    w.gz = reinterpret_cast<float *>(take((size_t)B * 2 * L * 4));
    w.s = reinterpret_cast<float *>(take((size_t)B * FM_K * 4));
    w.g = reinterpret_cast<float *>(take((size_t)B * 4));
    w.bytes = (size_t)(p - static_cast<char *>(ws));
    return w;
}

}  // namespace r4r

using namespace r4r;

extern "C" int r4r_deepconn_nparam(void) { return P_COUNT; }

extern "C" int r4r_deepconn_layout(int E, int L, int64_t *offsets, int64_t *sizes, int64_t *total) {
    R4R_REQUIRE(E > 0 && L > 0 && L <= MAX_L && offsets && sizes && total, "deepconn_layout: bad arguments");
    const Layout lay = make_layout(E, L);
    for (int i = 0; i < P_COUNT; ++i) { offsets[i] = lay.off[i]; sizes[i] = lay.size[i]; }
    *total = lay.total;
    return R4R_OK;
}

extern "C" size_t r4r_deepconn_ws_bytes(int64_t B, int T, int E, int L, int64_t V) {
    if (B < 0 || T <= 0 || E <= 0 || L <= 0 || V <= 0) return 0;
    return carve(nullptr, B, T, E, L, V).bytes;
}

// Byte offset of the [B, 2L] dropout-multiplier block inside the workspace (so a test can
// inject the very masks the device drew into the CPU oracle).
extern "C" size_t r4r_deepconn_ws_mult_offset(int64_t B, int T, int E, int L, int64_t V) {
    char base[1];
    const StepWs w = carve(base, B, T, E, L, V);
    return (size_t)(reinterpret_cast<char *>(w.mult) - base);
}

extern "C" size_t r4r_deepconn_ws_count_offset(int64_t B, int T, int E, int L, int64_t V, int tower, int buffer) {
    char base[1];
    const StepWs w = carve(base, B, T, E, L, V);
    return (size_t)(reinterpret_cast<char *>(w.count[buffer & 1][tower & 1]) - base);
}

// Token compaction of a batch into token-state buffer `token_buffer` (project-then-gather only;
// a no-op for configurations that run the direct conv).  Depends only on the indices: run it for
// batch k+1 on a side stream while step k computes, then pass tokens_ready = 1 to that step.
extern "C" int r4r_deepconn_tokens(const int64_t *user_idx, const int64_t *item_idx, void *ws, size_t ws_bytes,
                                   int64_t B, int T, int E, int L, int64_t V, int conv_algo, int token_buffer,
                                   int discard, void *stream) {
    R4R_REQUIRE(user_idx && item_idx && ws, "deepconn_tokens: null pointer");
    R4R_REQUIRE(token_buffer == 0 || token_buffer == 1, "deepconn_tokens: token_buffer must be 0 or 1");
    if (ws_bytes < r4r_deepconn_ws_bytes(B, T, E, L, V)) {
        set_error("deepconn_tokens: workspace %zu < %zu bytes", ws_bytes, r4r_deepconn_ws_bytes(B, T, E, L, V));
        return R4R_ERR_WORKSPACE;
    }
    if (B == 0 || textcnn_pick_algo(conv_algo, B, T, E, F_CONV) != R4R_CONV_PROJECT) return R4R_OK;
    const StepWs w = carve(ws, B, T, E, L, V);
    if (discard) {                                          // drop a prepared-but-unused token state
        for (int t = 0; t < 2; ++t) (void)hipMemsetAsync(w.count[token_buffer][t], 0, sizeof(int), as_stream(stream));
        return check_launch("deepconn_tokens(discard)");
    }
    ProjTower pt[2];
    const int64_t *idx[2] = {user_idx, item_idx};
    for (int t = 0; t < 2; ++t) {
        pt[t] = ProjTower{};
        pt[t].idx = idx[t];
        pt[t].flags = w.flags[token_buffer][t]; pt[t].slot = w.slot[token_buffer][t];
        pt[t].list = w.list[token_buffer][t]; pt[t].count = w.count[token_buffer][t];
    }
    return textcnn_proj_tokens_launch(V, pt, 2, B, T, /*zero_state=*/false, as_stream(stream));
}

extern "C" int r4r_deepconn_step(const float *table, int64_t V, const int64_t *user_idx, const int64_t *item_idx,
                                 const float *y, float *flat_p, float *flat_g,
                                 float *pred, float *se, float *sse_accum,
                                 void *ws, size_t ws_bytes,
                                 int64_t B, int T, int E, int L,
                                 float dropout_p, int training, uint64_t seed, uint64_t offset,
                                 float inv_denom, int conv_algo, int token_buffer, int tokens_ready,
                                 const int64_t *next_user_idx, const int64_t *next_item_idx,
                                 float *flat_m, float *flat_v, float lr, double beta1, double beta2, float eps,
                                 float weight_decay, int64_t adam_step,
                                 void *stream) {
    R4R_REQUIRE(table && user_idx && item_idx && flat_p && pred && ws, "deepconn_step: null pointer");
    R4R_REQUIRE(!flat_m == !flat_v, "deepconn_step: flat_m and flat_v go together");
    R4R_REQUIRE(!flat_m || (flat_g && adam_step >= 1), "deepconn_step: the fused optimiser needs gradients and "
                                                       "adam_step >= 1");
    R4R_REQUIRE(!next_user_idx == !next_item_idx, "deepconn_step: next_user_idx and next_item_idx go together");
    R4R_REQUIRE(!next_user_idx || flat_g, "deepconn_step: the next batch's tokens ride on the backward launches "
                                          "(training steps only)");
    R4R_REQUIRE(V > 0 && B >= 0 && T > 0, "deepconn_step: bad sizes");
    R4R_REQUIRE(E > 0 && E % 4 == 0, "deepconn_step: word_embed_size %d must be a positive multiple of 4", E);
    R4R_REQUIRE(L > 0 && L <= MAX_L, "deepconn_step: latent_size %d outside 1..%d", L, MAX_L);
    R4R_REQUIRE(!flat_g || (y && se), "deepconn_step: gradients need ratings y and the se buffer");
    R4R_REQUIRE(!y || se, "deepconn_step: se buffer required when y is given");
    R4R_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "deepconn_step: dropout %f outside [0,1)", (double)dropout_p);
    R4R_REQUIRE(B * (int64_t)((T + 2 + 127) / 128) < (1ll << 31), "deepconn_step: grid too large");
    R4R_REQUIRE(token_buffer == 0 || token_buffer == 1, "deepconn_step: token_buffer must be 0 or 1");
    if (ws_bytes < r4r_deepconn_ws_bytes(B, T, E, L, V)) {
        set_error("deepconn_step: workspace %zu < %zu bytes", ws_bytes, r4r_deepconn_ws_bytes(B, T, E, L, V));
        return R4R_ERR_WORKSPACE;
    }
    if (B == 0) return R4R_OK;
    hipStream_t st = as_stream(stream);
    const Layout lay = make_layout(E, L);
    const StepWs w = carve(ws, B, T, E, L, V);
    const float *P[P_COUNT];
    float *G[P_COUNT];
    for (int i = 0; i < P_COUNT; ++i) {
        P[i] = flat_p + lay.off[i];
        G[i] = flat_g ? flat_g + lay.off[i] : nullptr;
    }

    // 1+2: both towers, one grid -- either the direct gather-fused conv or project-then-gather
    const int64_t *idx[2] = {user_idx, item_idx};
    const int algo = textcnn_pick_algo(conv_algo, B, T, E, F_CONV);
    int tiles;
    if (algo == R4R_CONV_PROJECT) {
        ProjTower pt[2];
        for (int t = 0; t < 2; ++t) {
            pt[t].idx = idx[t];
            pt[t].conv_w = P[t ? P_ICW : P_UCW];
            pt[t].conv_b = P[t ? P_ICB : P_UCB];
            pt[t].flags = w.flags[token_buffer][t]; pt[t].slot = w.slot[token_buffer][t];
            pt[t].list = w.list[token_buffer][t]; pt[t].count = w.count[token_buffer][t];
            pt[t].ptab = w.ptab[t]; pt[t].pmax = w.pmax[t]; pt[t].parg = w.parg[t]; pt[t].wimg = w.wp[t];
        }
        if (!tokens_ready)
            if (int rc = textcnn_proj_tokens_launch(V, pt, 2, B, T, /*zero_state=*/false, st)) return rc;
        if (int rc = textcnn_proj_compute_launch(table, V, pt, 2, B, T, E, F_CONV, st)) return rc;
        tiles = proj_tiles(T);
    } else {
        FwdTower ft[2];
        for (int t = 0; t < 2; ++t) {
            ft[t].idx = idx[t];
            ft[t].conv_w = P[t ? P_ICW : P_UCW];
            ft[t].conv_b = P[t ? P_ICB : P_UCB];
            ft[t].wp = w.wp[t]; ft[t].pmax = w.pmax[t]; ft[t].parg = w.parg[t];
        }
        if (int rc = textcnn_fwd_launch(table, ft, 2, B, T, E, F_CONV, st)) return rc;
        tiles = textcnn_tiles(T);
    }

    // 3: head forward (+ its backward)
    HeadArgs h;
    for (int t = 0; t < 2; ++t) {
        h.pmax[t] = w.pmax[t]; h.parg[t] = w.parg[t];
        h.fc_w[t] = P[t ? P_IFW : P_UFW]; h.fc_b[t] = P[t ? P_IFB : P_UFB];
        h.pooled[t] = w.pooled[t]; h.argmax[t] = w.argmax[t]; h.g_pooled[t] = w.g_pooled[t];
    }
    h.V = P[P_FMV]; h.lin_w = P[P_FMLW]; h.lin_b = P[P_FMLB]; h.gbias = P[P_GB];
    h.y = y; h.mult = w.mult; h.x = w.x; h.s = w.s; h.g = w.g; h.gz = w.gz;
    h.pred = pred; h.se = se;
    h.B = B; h.L = L; h.tiles = tiles; h.training = training; h.want_grad = flat_g != nullptr;
    h.p_drop = dropout_p; h.inv_denom = inv_denom; h.seed = seed; h.offset = offset;
    if (L <= 16) deepconn_head_wg_kernel<16><<<(unsigned)B, 256, 0, st>>>(h);
    else if (L <= 32) deepconn_head_wg_kernel<32><<<(unsigned)B, 256, 0, st>>>(h);
    else if (L <= 64) deepconn_head_wg_kernel<64><<<(unsigned)B, 256, 0, st>>>(h);     // (hyper_params.py:63 has no bound)
    else deepconn_head_wg_kernel<128><<<(unsigned)B, 256, 0, st>>>(h);    // latent_size 65 .. 128: 115 KB of the CU's 160 KB of LDS

    if (!flat_g) {
        if (y && sse_accum) sse_only_kernel<<<1, 256, 0, st>>>(se, sse_accum, B);
        return check_launch("deepconn_step(forward)");
    }

    // 4: head parameter gradients (+ running sum of SE)
    HeadGradArgs hg;
    for (int t = 0; t < 2; ++t) {
        hg.pooled[t] = w.pooled[t];
        hg.g_fc_w[t] = G[t ? P_IFW : P_UFW]; hg.g_fc_b[t] = G[t ? P_IFB : P_UFB];
    }
    hg.x = w.x; hg.s = w.s; hg.g = w.g; hg.gz = w.gz; hg.V = P[P_FMV]; hg.se = se;
    hg.g_V = G[P_FMV]; hg.g_lin_w = G[P_FMLW]; hg.g_lin_b = G[P_FMLB]; hg.g_gb = G[P_GB];
    hg.sse_accum = sse_accum; hg.B = B; hg.L = L;
    const int nout = 2 * (L * F_CONV + L) + 2 * L * FM_K + 2 * L + 3;
    const int hg_blocks = (nout + HG_COLS - 1) / HG_COLS;

    // 5: conv weight gradients of both towers + the head gradients, one launch
    WgradTower wt[2];
    WgradArgs wa;
    for (int t = 0; t < 2; ++t) {
        wt[t].idx = idx[t]; wt[t].g_pooled = w.g_pooled[t]; wt[t].argmax = w.argmax[t];
        wt[t].part_w = w.part_w[t]; wt[t].part_b = w.part_b[t];
        wt[t].d_w = G[t ? P_ICW : P_UCW]; wt[t].d_b = G[t ? P_ICB : P_UCB];
    }
    for (int k = 0; k < MAX_TOWERS; ++k) wa.t[k] = wt[k < 2 ? k : 0];
    wa.table = table; wa.N = B; wa.T = T; wa.E = E; wa.F = F_CONV;
    wa.table_bytes = (int64_t)V * E * 4;                   // (the wide wgrad reads the rows through a buffer resource)
    wa.nsplit = textcnn_wgrad_splits(B);
    wa.per_split = (int)cdiv(B, wa.nsplit);
    // token state of the next batch (same B, T) into the OTHER token buffer, if asked for and
    // if that batch will run project-then-gather
    const bool prefetch = next_user_idx && algo == R4R_CONV_PROJECT;
    TokenArgs nx{};
    if (prefetch) {
        ProjTower nt[2];
        const int64_t *nidx[2] = {next_user_idx, next_item_idx};
        const int ob = token_buffer ^ 1;
        for (int t = 0; t < 2; ++t) {
            nt[t] = ProjTower{};
            nt[t].idx = nidx[t];
            nt[t].flags = w.flags[ob][t]; nt[t].slot = w.slot[ob][t];
            nt[t].list = w.list[ob][t]; nt[t].count = w.count[ob][t];
        }
        nx = make_token_args(V, nt, 2, B, T);
    }
    {
        ScopedTiming tm(R4R_TIMING_TEXTCNN_WGRAD, st);
        deepconn_backward_kernel<<<dim3(F_CONV, wa.nsplit, prefetch ? 4 : 3), WG_THREADS, 0, st>>>(DcBackwardArgs{wa, hg, hg_blocks, nx});
    }
    // 6: wgrad partial reduce -> flat gradient buffer (+ compaction of the next batch's tokens)
    const int red_blocks = (F_CONV * 3 * E + F_CONV + RED_THREADS - 1) / RED_THREADS;
    const int comp_blocks = prefetch ? (int)cdiv((V + 3) / 4, RED_THREADS * compact_groups(V)) : 0;
    FusedAdam opt{};
    int opt_blocks = 0;
    if (flat_m) {
        opt.on = 1;
        opt.p = flat_p; opt.m = flat_m; opt.v = flat_v; opt.g = flat_g;
        opt.lo[0] = lay.off[P_UFW]; opt.hi[0] = lay.off[P_ICW];      // user fc weight + bias
        opt.lo[1] = lay.off[P_IFW]; opt.hi[1] = lay.total;           // item fc, FM, global bias
        opt.s = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
        const int64_t longest = opt.hi[0] - opt.lo[0] > opt.hi[1] - opt.lo[1] ? opt.hi[0] - opt.lo[0] : opt.hi[1] - opt.lo[1];
        opt_blocks = (int)cdiv(longest, RED_THREADS);
    }
    deepconn_reduce_kernel<<<dim3(red_blocks + comp_blocks + opt_blocks, 2), RED_THREADS, 0, st>>>(
        wa, red_blocks, comp_blocks, nx, opt);
    return check_launch("deepconn_step");
}

// accum[0] += sum of se[0 .. n): one workgroup, strided per-thread sums and a fixed tree (deterministic) -- the
// running train metric of a step whose launches do not carry it (main.py:57: float(torch.sum(loss)), kept on the device)
extern "C" int r4r_sse_accumulate(const float *se, int64_t n, float *accum, void *stream) {
    R4R_REQUIRE(accum && (se || n == 0) && n >= 0, "sse_accumulate: bad arguments");
    if (n == 0) return R4R_OK;
    sse_only_kernel<<<1, 256, 0, as_stream(stream)>>>(se, accum, n);
    return check_launch("sse_accumulate");
}
