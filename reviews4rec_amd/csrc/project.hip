// "Project-then-gather" TextCNN forward for gfx950.
//
// Same function as textcnn.hip's direct kernel -- relu/max-pool of
//   Y[n,f,p] = b[f] + sum_j sum_e table[idx[n,p+j-2], e] * W[f,j,e]
// (common_pytorch_models.py:14-17,29-31 fed by DeepCoNN.py:53-54) -- restructured
// around two facts of the reference: the word table is FROZEN
// (Embedding.from_pretrained, DeepCoNN.py:15) and tokens repeat heavily inside a
// batch (Zipf vocabulary + zero padding, data.py:198-199).  Re-associating the sum,
//   Y[n,f,p] = b[f] + sum_j  Q_j[ idx[n,p+j-2], f ],     Q_j = table . W[:,j,:]^T,
// so the matrix work only has to be done once per DISTINCT token of the batch:
//   1. mark + compact the distinct tokens of each tower's documents
//   2. projection GEMM (fp32 MFMA): Q[u, j*100+f] for the U distinct tokens  -- flops
//      proportional to U (<= positions; ~8x fewer at cfg3 B=128, ~20x at B=1024)
//   3. gather-add-max: stream the positions of every document, gather the three
//      400-byte tap rows, slide-add them, keep (max, first argmax) per filter.
//      1.2 KB per position -- the HBM/L2-bound embedding gather of the north star.
// Step 3 writes the same per-128-position partials the direct kernel writes, so the
// pool-finish / head / wgrad kernels are shared.  Only the summation ORDER differs
// from the direct kernel (per-tap dot products first, then 3 adds): fp32 rounding
// level, covered by the same parity tests.
#include <stdlib.h>

#include <type_traits>

#include "textcnn.h"
#include "tokens_device.h"
#include "trace_device.h"

namespace r4r {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PF = 100;            // filters (rows of 100 floats = 400 B per tap)
constexpr int PROW = 3 * PF;       // projected row: 3 taps x 100 filters
#ifndef R4R_PSTR
#define R4R_PSTR 304
#endif
constexpr int PSTR = R4R_PSTR;     // its stride in the projected-row table: 1,216 B = 19 whole 64-byte pieces (at 1,200 B three
                                   // of four 64-byte store pieces straddled two requests, and a gathered row touched 10.4 lines instead of 10)
static_assert(PSTR >= PROW && PSTR % 4 == 0, "projected-row stride");
constexpr int PN = 304;            // GEMM N: 300 padded to 19 tiles of 16
constexpr int PNT = PN / 16;       // 19
constexpr int PEC = 16;            // K chunk
constexpr int PS = PEC + 8;        // LDS row stride (floats), == 8 mod 16: conflict-free b128 reads
constexpr int GM = 2;              // 16-row tiles per wave
constexpr int PM = 64 * GM;        // GEMM rows per workgroup (4 waves x GM row tiles of 16)
constexpr int PNH = 10;                                // column tiles of the wider half (10 + 9 = 19)
constexpr int PNH_COLS = PNH * 16;                     // 160
constexpr int SEG = 128;           // positions per partial (matches the direct kernel's NW=4 tile)

// Experiment switches of the projection GEMM (make variant EXTRA="-D..."; defaults = the shipped form)

// Staging row of thread-quad s (= tid >> 2).  A ds_write_b128 is served in groups of 8 consecutive lanes = two
// quads over 32 banks; rows s and s + 1 at the 24-float stride overlap in 8 banks (a 2-way conflict: 27 % of the
// kernel's LDS cycles, profiles/r02k_bench_pmc_summary.json), rows s and s + 2 are 48 floats apart = 16 banks: none.
__device__ __forceinline__ int stage_row(int s) {
    return (s & ~3) | ((s & 1) << 1) | ((s >> 1) & 1);
}

__device__ __forceinline__ void store_row4(float *dst, const f32x4 v) {
    *reinterpret_cast<f32x4 *>(dst) = v;
}

struct ProjArgs {
    ProjTower t[MAX_TOWERS];
    const float *table;
    int64_t N, V;
    int T, E, F, nchunk, tiles, cap;
    int ntower, balanced;          // 1: the GEMM may use its 7-row-tile form; 0: tile form (R4R_GEMM=tile); 2: tile form, whole tiles only (R4R_GEMM=whole)
};

#ifdef R4R_TRACE
// Timeline instrumentation (make trace): 8 words per workgroup of the projection GEMM --
// start, operands-staged, loop-done, end (s_memrealtime, 100 MHz), HW_ID, XCC_ID, active flag.
__device__ unsigned long long *g_trace = nullptr;
extern "C" int r4r_debug_trace(void *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#define TRACE_STAMP(k)                                                                                  \
    if (g_trace && threadIdx.x == 0)                                                                    \
        g_trace[((size_t)blockIdx.x) * 8 + (k)] = wall_clock64();
// the LAST wave's time (the A-resident form's waves run free in their later passes: wave 0 is not the workgroup)
#define TRACE_STAMP_LAST(k)                                                                             \
    if (g_trace && (threadIdx.x & 63) == 0)                                                             \
        atomicMax(g_trace + ((size_t)blockIdx.x) * 8 + (k), (unsigned long long)wall_clock64());
// where the workgroup runs (HW_ID, XCC_ID) + its active flag; shader-cycle spans (vs the 100 MHz stamps: the clock)
#define TRACE_WHERE(xcc_mask)                                                                           \
    if (g_trace && threadIdx.x == 0) {                                                                  \
        unsigned long long *tr = g_trace + ((size_t)blockIdx.x) * 8;                                    \
        tr[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);                                              \
        tr[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & (xcc_mask);                                \
        tr[6] = 1;                                                                                      \
    }
#define TRACE_CLK(name) const unsigned long long name = __builtin_readcyclecounter();
#define TRACE_CLK_SPAN(word, expr)                                                                      \
    if (g_trace && threadIdx.x == 0) g_trace[((size_t)blockIdx.x) * 8 + (word)] = (expr);
// (word 5 of the trace record: XCC id in the low byte, the end of the staging pass's prologue above it)
#define PRO_STAMP if (g_trace && threadIdx.x == 0) g_trace[((size_t)blockIdx.x) * 8 + 5] |= (wall_clock64() << 8);
#else
#define TRACE_STAMP(k)
#define TRACE_STAMP_LAST(k)
#define TRACE_WHERE(xcc_mask)
#define TRACE_CLK(name)
#define TRACE_CLK_SPAN(word, expr)
#define PRO_STAMP
#endif

// ---- 0. zero the token-state (callers whose workspace is not persistently zeroed).  A kernel
// rather than hipMemsetAsync: memset nodes misbehaved under hipGraph replay (memory fault).
__global__ void proj_zero_kernel(ProjArgs a) {
    const ProjTower &tw = a.t[blockIdx.y];
    const int64_t n = ((a.V + 3) / 4) * 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        tw.flags[i] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) tw.count[0] = 0;
}

// ---- 1a. mark the tokens each tower's documents use (body: tokens_device.h)
__global__ void proj_mark_kernel(TokenArgs a) { token_mark_block(a, blockIdx.x, gridDim.x, blockDim.x); }

// ---- 1b. compact: slot[v] = dense row id of token v (or -1), list[row] = v, count
// (body: tokens_device.h).  grid = (ceil(V / 4096), ntower).
__global__ __launch_bounds__(1024) void proj_compact_kernel(TokenArgs a) {
    token_compact_block<16>(a.t[blockIdx.y], a.V, blockIdx.x);
}

// ---- 2. projection GEMM: Q[row, j*100+f] = table[list[row], :] . W[f, j, :].
// The B operand is staged straight from the conv weight [F][3][E]: LDS row n = j*100+f takes
// the 16 contiguous floats W[f][j][c*16 .. c*16+15] (four float4 per row).
// persistent grid (one workgroup per CU at most); ONE workgroup of 8 waves per 128-row tile: wave w owns rows
// [32 (w & 3), +32) x column half (w >> 2) of the 304 columns (2 x (10 or 9) accumulators of
// 16x16), so the A tile is staged once for both halves and a CU holds one workgroup whose two
// waves per SIMD cover each other's stalls.  K runs in chunks of 16 floats through two LDS
// buffers, two operand register sets and two staging register sets.
//
// (What bounds it -- the clock under load, what a memory instruction costs the MFMA stream, prologue + epilogue --
// and the decompositions that were built and measured slower: DESIGN.md 4.2 and Appendix A.)
constexpr int GEMM_THREADS = 512;
constexpr int WB_ROWS = 320;                              // B rows staged: 304 padded to 5 x 64
constexpr int GEMM_BUF = (PM + WB_ROWS) * PS;               // floats per LDS buffer
constexpr int GEMM_LDS_BYTES = 2 * GEMM_BUF * 4;              // 86,016 B

// GMT = row tiles of 16 per wave.  2: the full tile -- wave w owns rows 32 (w & 3) .. of the 128 and the column
// half (w >> 2); `colbase` is 0 and the workgroup stages all 320 B rows (NB = 3 | 2 of them per thread).  1: a
// COLUMN PART of a tile (the last, partial round of the tile list, below) -- wave w owns rows 16 w ..; all eight
// waves share the NTILE column tiles from `colbase` on, and only those B rows are staged (NB rounds of 128).
template <int NTILE, int NB, int GMT = GM>
__device__ __forceinline__ void proj_gemm_body(const ProjArgs &a, float *lds, int tower, int row0, int colbase = 0) {
    const ProjTower &tw = a.t[tower];
    const int count = tw.count[0];
    TRACE_STAMP(0)
    TRACE_WHERE(~0u)
    const float *__restrict__ table = a.table;
    const int E = a.E, nchunk = a.nchunk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, q = lane >> 4;
    const int col0 = GMT == GM ? (wave >> 2) * PNH_COLS : 0;   // first column inside the staged B rows
    const int rowgrp = GMT == GM ? (wave & 3) : wave;

    // staging role: float4 column c4 of A row (tid >> 2) and of B rows (tid >> 2) + 128 k
    // (k < NB: waves 0..3 stage three, rows 0..319; waves 4..7 two).
    const int c4 = tid & 3, srow = stage_row(tid >> 2);
    const float *aptr = table + (long)tw.list[min(row0 + srow, count - 1)] * E;
    const float *bptr[NB];
    const float *__restrict__ conv_w = tw.conv_w;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int n = colbase + srow + 128 * k;             // n = j * 100 + f
        const int j = n / PF, f = n - j * PF;
        bptr[k] = conv_w + (n < PROW ? ((long)f * 3 + j) * E : 0);
    }
    // Every load is unconditional (hipcc turns a predicated load into a branch and a full counter
    // wait): rows past `count` re-read the last valid row and padding columns re-read W[0][0] --
    // their results are never stored -- and the K tail (E % 16) re-reads the last float4 of the
    // row, with the B side zeroed before it reaches LDS so the products vanish.  (A second set of
    // staging registers, giving the loads two chunks to arrive, measured 1 % slower.)
    f32x4 ar, br[NB];
    auto issue_loads = [&](int c) {                          // any c: past the end it re-reads the tail
        const int e = min(c * PEC + c4 * 4, E - 4);
        ar = *reinterpret_cast<const f32x4 *>(aptr + e);
#pragma unroll
        for (int k = 0; k < NB; ++k) br[k] = *reinterpret_cast<const f32x4 *>(bptr[k] + e);
    };
    auto write_lds = [&](float *buf, int c) {
        const float keep = (c * PEC + c4 * 4 < E) ? 1.f : 0.f;  // zero the K tail of B (branch-free)
        *reinterpret_cast<f32x4 *>(buf + srow * PS + c4 * 4) = ar;
        float *Bl = buf + PM * PS;
#pragma unroll
        for (int k = 0; k < NB; ++k)
            *reinterpret_cast<f32x4 *>(Bl + (srow + 128 * k) * PS + c4 * 4) = br[k] * keep;
    };
    f32x4 acc[GMT][NTILE];
#pragma unroll
    for (int mi = 0; mi < GMT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NTILE; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int aoffl = (rowgrp * 16 * GMT + lrow) * PS + q * 4, boffl = (PM + col0 + lrow) * PS + q * 4;
    auto read_ops = [&](const float *buf, f32x4 (&av)[GMT], f32x4 (&b)[NTILE]) {
#pragma unroll
        for (int mi = 0; mi < GMT; ++mi) av[mi] = *reinterpret_cast<const f32x4 *>(buf + aoffl + mi * 16 * PS);
#pragma unroll
        for (int ni = 0; ni < NTILE; ++ni) b[ni] = *reinterpret_cast<const f32x4 *>(buf + boffl + ni * 16 * PS);
    };
    auto mfma = [&](const f32x4 (&av)[GMT], const f32x4 (&b)[NTILE], int kk) {
#pragma unroll
        for (int mi = 0; mi < GMT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NTILE; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi][kk], b[ni][kk], acc[mi][ni], 0, 0, 0);
    };
    // Operand registers are double-buffered too: the staging of chunk c+2 (registers -> LDS),
    // the global loads of chunk c+3 and the ds_reads of chunk c+1 are all issued among the
    // MFMAs of chunk c, so between two chunks the matrix pipe only waits for the barrier.
    // The body is branch-free (one scheduling region): past the last chunk it stages and reads
    // data nobody consumes.  A 16x16x4 fp32 MFMA occupies the pipe for 32 cycles, which leaves
    // room for one memory instruction behind every few of them; left to itself hipcc emits
    // the reads, the stores and the loads as three bursts during which the pipe idles.
    // chunk c: operands `cur` are in registers; LDS buffer (c+1)&1 holds chunk c+1 once the
    // barrier is passed; the staging registers hold chunk c+2 (loaded a whole chunk ago).
    auto step = [&](int c, const f32x4 (&cav)[GMT], const f32x4 (&cb)[NTILE], f32x4 (&nav)[GMT], f32x4 (&nb)[NTILE]) {
        __syncthreads();                                    // chunk c+1 visible; buffer c&1 fully read
        write_lds(lds + (c & 1) * GEMM_BUF, c + 2);
        issue_loads(c + 3);
        mfma(cav, cb, 0);
        mfma(cav, cb, 1);
        read_ops(lds + ((c + 1) & 1) * GEMM_BUF, nav, nb);
        mfma(cav, cb, 2);
        mfma(cav, cb, 3);
        // schedule: (1 LDS write, 3 MFMA) x NSTG | (1 load, 3 MFMA) x NSTG | (1 LDS read, 3 MFMA) x NREAD, rest
        // (PER MFMAs behind every memory instruction: 3 for the full tile's 80 | 72 MFMAs per chunk, fewer for
        // the column parts' 40 .. 16)
        constexpr int NREAD = GMT + NTILE, NSTG = 1 + NB, NM = 4 * GMT * NTILE;
        constexpr int PER = NM / (NREAD + 2 * NSTG) >= 3 ? 3 : (NM / (NREAD + 2 * NSTG) >= 1 ? NM / (NREAD + 2 * NSTG) : 1);
#pragma unroll
        for (int i = 0; i < NSTG; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
        }
#pragma unroll
        for (int i = 0; i < NSTG; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
        }
#pragma unroll
        for (int i = 0; i < NREAD; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
        }
        if constexpr (NM > PER * (NREAD + 2 * NSTG)) __builtin_amdgcn_sched_group_barrier(0x008, NM - PER * (NREAD + 2 * NSTG), 0);
    };
    f32x4 av0[GMT], b0[NTILE], av1[GMT], b1[NTILE];
    issue_loads(0);
    write_lds(lds, 0);
    issue_loads(1);
    __syncthreads();
    read_ops(lds, av0, b0);
    write_lds(lds + GEMM_BUF, 1);
    issue_loads(2);
    TRACE_STAMP(1)
    TRACE_CLK(clk0)
    // (the odd last chunk is peeled: with an exit in the middle of the loop body hipcc's counter
    // analysis falls back to vmcnt(0) at the loop head)
    int c = 0;
    for (; c + 1 < nchunk; c += 2) {
        step(c, av0, b0, av1, b1);
        step(c + 1, av1, b1, av0, b0);
    }
    if (c < nchunk) step(c, av0, b0, av1, b1);
    TRACE_CLK_SPAN(7, __builtin_readcyclecounter() - clk0)      // shader cycles of the loop
    __syncthreads();                                        // all operand reads done: LDS is free
    TRACE_STAMP(2)
    // Epilogue.  C layout of the MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg -- storing
    // that directly is 80 scattered 4-byte stores per lane.  Instead each wave transposes one
    // 16-row tile at a time through its own LDS slab and writes whole row segments as float4
    // (an LDS queue is in-order per wave, so no barrier is needed inside a wave).
    constexpr int TS = NTILE * 16 + 4;
    float *slab = lds + wave * (16 * (PNH_COLS + 4));
    constexpr int NV = NTILE * 4;
#pragma unroll
    for (int mi = 0; mi < GMT; ++mi) {
#pragma unroll
        for (int ni = 0; ni < NTILE; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[(q * 4 + r) * TS + ni * 16 + lrow] = acc[mi][ni][r];
        for (int i = lane; i < 16 * NV; i += 64) {
            const int rr = i / NV, cv = i - rr * NV;
            const int row = row0 + rowgrp * 16 * GMT + mi * 16 + rr;
            const int col = colbase + col0 + cv * 4;
            if (row < count && col < PROW)
                *reinterpret_cast<f32x4 *>(tw.ptab + (size_t)row * PSTR + col) =
                    *reinterpret_cast<const f32x4 *>(slab + rr * TS + cv * 4);
        }
    }
    TRACE_STAMP(3)
}

// ---- 2b. the balanced form of the projection GEMM.
// The tile form above gives every workgroup 8 row tiles of 16 rows (128 rows x 304 columns = 152
// MFMA tiles, 38 per SIMD) and leaves the CUs without a tile idle: at the headline batch 231 tiles on
// 256 CUs, i.e. the launch takes the time of 8 row tiles where 7.2 per CU would do.  Row tiles cannot
// be split, but a row tile's 19 COLUMN tiles can: here every workgroup owns 7 private row tiles and
// the row tiles left over (total - 7 per workgroup) are each SHARED by >= 3 workgroups, which stage
// its 16 rows next to their own 112 (the LDS image is the tile form's: 128 A rows + 320 B rows per
// K chunk) and compute <= 7 of its column tiles each.  Inside the workgroup the 7 x 19 private tiles
// are split by COLUMNS over the four SIMDs -- 5 / 5 / 5 / 4 column tiles, the two waves of a SIMD
// sharing their 7 x NC block evenly -- so SIMDs 0-2 run 35 MFMA tiles per K step and SIMD 3 runs 28 +
// the shared row tile's <= 7 (computed unconditionally, stored only where assigned: no branch in the
// loop): 35 instead of 38 per SIMD, and every CU busy.  Same K order per output element as the tile form: identical bits.
// The plan is a pure function of the towers' distinct-token counts (read by every workgroup); when
// it does not apply -- fewer than 7 or more than 7 1/3 row tiles per workgroup -- the tile form runs.
constexpr int G7_WGS = 256;        // persistent workgroups = CUs of an MI355X (one workgroup per CU by LDS)
constexpr int G7_ROWS = 7;         // private row tiles per workgroup

struct Gemm7Plan { int tower, row0, sh_row0, sh_c0, sh_n; };

__device__ __forceinline__ bool gemm7_plan(const ProjArgs &a, int wg, int nwg, Gemm7Plan &p) {
    int rt[MAX_TOWERS], g[MAX_TOWERS];
    int U = 0;
    _Pragma("unroll") for (int t = 0; t < MAX_TOWERS; ++t) { rt[t] = t < a.ntower ? (a.t[t].count[0] + 15) >> 4 : 0; U += rt[t]; g[t] = 0; }
    const int cap = nwg < G7_WGS ? nwg : G7_WGS;
    int G = U / G7_ROWS;
    if (G > cap) G = cap;
    if (G < a.ntower) return false;
    _Pragma("unroll") for (int t = 0; t < MAX_TOWERS; ++t) if (t < a.ntower) {
        g[t] = rt[t] * G / U;                               // (products < 2^24: 32-bit arithmetic)
        const int left = rt[t] - G7_ROWS * g[t];           // row tiles nobody owns: shared, >= 3 workgroups each
        if (g[t] < 1 || left < 0 || 3 * left > g[t]) return false;
    }
    p.tower = -1;
    int base = 0;
    _Pragma("unroll") for (int t = 0; t < MAX_TOWERS; ++t) if (t < a.ntower) {
        if (wg >= base && wg < base + g[t]) {
            const int wl = wg - base, left = rt[t] - G7_ROWS * g[t];
            p.tower = t;
            p.row0 = wl * G7_ROWS * 16;
            p.sh_row0 = -1; p.sh_c0 = 0; p.sh_n = 0;
            if (left > 0) {
                const int lt = wl * left / g[t];                                   // the shared row tile of this workgroup
                const int w_lo = (lt * g[t] + left - 1) / left;                     // its sharers: [w_lo, w_hi)
                const int w_hi = ((lt + 1) * g[t] + left - 1) / left;
                const int n = w_hi - w_lo, k = wl - w_lo;
                p.sh_row0 = (G7_ROWS * g[t] + lt) * 16;
                p.sh_c0 = PNT * k / n;
                p.sh_n = PNT * (k + 1) / n - p.sh_c0;                               // <= ceil(19 / 3) = 7 (10 with two sharers)
            }
        }
        base += g[t];
    }
    return true;                                            // p.tower < 0: nothing to do for this workgroup
}

// table operand x weight operand -> accumulator (R4R_EPI >= 1: roles swapped, see the tile form)
#define MFMA4(tab, wgt, c) __builtin_amdgcn_mfma_f32_16x16x4f32(tab, wgt, c, 0, 0, 0)

// One wave of the balanced form.  SIMD = wave & 3 owns NC column tiles (5, 5, 5, 4) of the 7 private row
// tiles; its two waves split that 7 x NC block CHECKERBOARD-wise so that both carry the same load (18 / 17
// tiles at NC = 5) -- a 4 + 3 row split leaves the lighter wave waiting at the chunk barrier while the
// heavier one finishes alone, with nobody to fill the issue holes its memory instructions tear (measured:
// 81 % pipe utilisation in the loop against the tile form's 92 %):
//     HALF 0: row tiles 0..3 x the first CA columns  +  row tiles 4..6 x the rest     (CA = ceil(NC / 2))
//     HALF 1: row tiles 0..3 x the rest              +  row tiles 4..6 x the first CA
// EX: column tiles of the shared row tile this wave adds (SIMD 3 only).  NB: B rows this thread stages
// per chunk (waves 0-3 stage three and are HALF 1, waves 4-7 two and are HALF 0).
template <int NC, int HALF, int EX, int NB>
__device__ __forceinline__ void proj_gemm7_body(const ProjArgs &a, float *lds, const Gemm7Plan &p) {
    const ProjTower &tw = a.t[p.tower];
    const int count = tw.count[0];
    TRACE_STAMP(0)
    TRACE_WHERE(~0u)
    const float *__restrict__ table = a.table;
    const int E = a.E, nchunk = a.nchunk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, q = lane >> 4, simd = wave & 3;
    constexpr int NE = EX > 0 ? EX : 1;
    constexpr int CA = (NC + 1) / 2;
    constexpr int CT = HALF ? NC - CA : CA, CT0 = HALF ? CA : 0;      // column tiles of row tiles 0..3
    constexpr int CB = HALF ? CA : NC - CA, CB0 = HALF ? 0 : CA;      // column tiles of row tiles 4..6
    constexpr int RT = 4, RB = G7_ROWS - RT;

    // staging role, as in the tile form: float4 column c4 of A row (tid >> 2) -- LDS rows 0..111 are the
    // private rows, 112..127 the shared row tile -- and of B rows (tid >> 2) + 128 k
    const int c4 = tid & 3, srow = stage_row(tid >> 2);
    int grow = srow < G7_ROWS * 16 ? p.row0 + srow : (p.sh_row0 < 0 ? 0 : p.sh_row0 + srow - G7_ROWS * 16);
    grow = grow < count ? grow : count - 1;
    const float *aptr = table + (long)tw.list[grow] * E;
    const float *bptr[NB];
    const float *__restrict__ conv_w = tw.conv_w;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int n = srow + 128 * k;                       // n = j * 100 + f
        const int j = n / PF, f = n - j * PF;
        bptr[k] = conv_w + (n < PROW ? ((long)f * 3 + j) * E : 0);
    }
    f32x4 ar, br[NB];
    auto issue_loads = [&](int c) {
        const int e = min(c * PEC + c4 * 4, E - 4);
        ar = *reinterpret_cast<const f32x4 *>(aptr + e);
#pragma unroll
        for (int k = 0; k < NB; ++k) br[k] = *reinterpret_cast<const f32x4 *>(bptr[k] + e);
    };
    auto write_lds = [&](float *buf, int c) {
        const float keep = (c * PEC + c4 * 4 < E) ? 1.f : 0.f;
        *reinterpret_cast<f32x4 *>(buf + srow * PS + c4 * 4) = ar;
        float *Bl = buf + PM * PS;
#pragma unroll
        for (int k = 0; k < NB; ++k)
            *reinterpret_cast<f32x4 *>(Bl + (srow + 128 * k) * PS + c4 * 4) = br[k] * keep;
    };
    f32x4 acct[RT][CT], accb[RB][CB], ex[NE];
#pragma unroll
    for (int mi = 0; mi < RT; ++mi)
#pragma unroll
        for (int ni = 0; ni < CT; ++ni) acct[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mi = 0; mi < RB; ++mi)
#pragma unroll
        for (int ni = 0; ni < CB; ++ni) accb[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NE; ++j) ex[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int EOFF = HALF ? 4 : 0;                      // first shared-tile unit of this wave (SIMD 3 only)
    const int cbase = simd * 5;                             // first column tile (SIMD 3: 15..18)
    const int aoffl = lrow * PS + q * 4, boffl = (PM + cbase * 16 + lrow) * PS + q * 4;
    const int saoffl = (G7_ROWS * 16 + lrow) * PS + q * 4;
    int ecol[NE], eboffl[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        ecol[j] = min(p.sh_c0 + EOFF + j, PNT - 1);         // beyond the assignment: computed, never stored
        eboffl[j] = (PM + ecol[j] * 16 + lrow) * PS + q * 4;
    }
    struct Ops { f32x4 av[G7_ROWS], b[NC], sa, eb[NE]; };
    auto read_ops = [&](const float *buf, Ops &o) {
#pragma unroll
        for (int mi = 0; mi < G7_ROWS; ++mi) o.av[mi] = *reinterpret_cast<const f32x4 *>(buf + aoffl + mi * 16 * PS);
#pragma unroll
        for (int ni = 0; ni < NC; ++ni) o.b[ni] = *reinterpret_cast<const f32x4 *>(buf + boffl + ni * 16 * PS);
        if (EX > 0) {
            o.sa = *reinterpret_cast<const f32x4 *>(buf + saoffl);
#pragma unroll
            for (int j = 0; j < NE; ++j) o.eb[j] = *reinterpret_cast<const f32x4 *>(buf + eboffl[j]);
        }
    };
    auto mfma = [&](const Ops &o, int kk) {
#pragma unroll
        for (int mi = 0; mi < RT; ++mi)
#pragma unroll
            for (int ni = 0; ni < CT; ++ni)
                acct[mi][ni] = MFMA4(o.av[mi][kk], o.b[CT0 + ni][kk], acct[mi][ni]);
#pragma unroll
        for (int mi = 0; mi < RB; ++mi)
#pragma unroll
            for (int ni = 0; ni < CB; ++ni)
                accb[mi][ni] = MFMA4(o.av[RT + mi][kk], o.b[CB0 + ni][kk], accb[mi][ni]);
        if (EX > 0) {
#pragma unroll
            for (int j = 0; j < NE; ++j)
                ex[j] = MFMA4(o.sa[kk], o.eb[j][kk], ex[j]);
        }
    };
    // the tile form's software pipeline (see there): LDS and operand registers double-buffered, staging
    // registers a chunk ahead, one memory instruction behind every three MFMAs
    auto step = [&](int c, const Ops &cur, Ops &nxt) {
        __syncthreads();
        write_lds(lds + (c & 1) * GEMM_BUF, c + 2);
        issue_loads(c + 3);
        mfma(cur, 0);
        mfma(cur, 1);
        read_ops(lds + ((c + 1) & 1) * GEMM_BUF, nxt);
        mfma(cur, 2);
        mfma(cur, 3);
        constexpr int NREAD = G7_ROWS + NC + (EX > 0 ? 1 + EX : 0), NSTG = 1 + NB;
        constexpr int NM = 4 * (RT * CT + RB * CB + EX);
        constexpr int GAP = NM / (NREAD + 2 * NSTG) >= 3 ? 3 : 2;     // MFMAs between two memory instructions
        static_assert(NM >= GAP * (NREAD + 2 * NSTG), "more memory instructions than MFMA slots");
#pragma unroll
        for (int i = 0; i < NSTG; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, GAP, 0);
        }
#pragma unroll
        for (int i = 0; i < NSTG; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, GAP, 0);
        }
#pragma unroll
        for (int i = 0; i < NREAD; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, GAP, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NM - GAP * (NREAD + 2 * NSTG), 0);
    };
    Ops o0, o1;
    issue_loads(0);
    write_lds(lds, 0);
    issue_loads(1);
    __syncthreads();
    read_ops(lds, o0);
    write_lds(lds + GEMM_BUF, 1);
    issue_loads(2);
    TRACE_STAMP(1)
    TRACE_CLK(clk0)
    int c = 0;
    for (; c + 1 < nchunk; c += 2) {
        step(c, o0, o1);
        step(c + 1, o1, o0);
    }
    if (c < nchunk) step(c, o0, o1);
    TRACE_CLK_SPAN(7, __builtin_readcyclecounter() - clk0)
    __syncthreads();                                        // all operand reads done: LDS is free
    TRACE_STAMP(2)
    // epilogue: per wave, one 16-row tile at a time through its own LDS slab, whole row segments as float4
    constexpr int CMAX = CT > CB ? CT : CB;
    constexpr int TS = CMAX * 16 + 4;
    float *slab = lds + wave * (16 * (PNH_COLS + 4));
    auto store_tile = [&](const f32x4 *row_acc, int ntile, int row_first, int col_tile0) {
        const int nv = ntile * 4;
        for (int ni = 0; ni < ntile; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[(q * 4 + r) * TS + ni * 16 + lrow] = row_acc[ni][r];
        for (int i = lane; i < 16 * nv; i += 64) {
            const int rr = i / nv, cv = i - rr * nv;
            const int row = row_first + rr, col = col_tile0 * 16 + cv * 4;
            if (row < count && col < PROW)
                *reinterpret_cast<f32x4 *>(tw.ptab + (size_t)row * PSTR + col) =
                    *reinterpret_cast<const f32x4 *>(slab + rr * TS + cv * 4);
        }
    };
#pragma unroll
    for (int mi = 0; mi < RT; ++mi) store_tile(acct[mi], CT, p.row0 + mi * 16, cbase + CT0);
#pragma unroll
    for (int mi = 0; mi < RB; ++mi) store_tile(accb[mi], CB, p.row0 + (RT + mi) * 16, cbase + CB0);
    if (EX > 0) {
#pragma unroll
        for (int j = 0; j < NE; ++j)
            if (EOFF + j < p.sh_n && p.sh_row0 >= 0) store_tile(&ex[j], 1, p.sh_row0, ecol[j]);
    }
    TRACE_STAMP(3)
}

// ---- 2c. the A-resident form of the balanced decomposition (form 3).
// In the forms above every workgroup finishes its K loop at the same moment and 35.5 MB of projected rows leave in one
// burst, with no MFMA issued meanwhile.  Output can only leave earlier if it is COMPLETE earlier, i.e. if a workgroup
// runs all of K over part of its tile, then all of K again over the rest -- without staging its operands twice.
// Here the workgroup's A rows (P private row tiles + the shared one, <= 128 rows x all of K, swizzled 64-byte chunk
// rows: 155,648 B of the CU's 160 KB at E = 300) stay RESIDENT in LDS once staged, and the B operand never passes
// through LDS: every wave owns whole COLUMN tiles (3 or 2 of the 19: waves w and w + 4 of a SIMD share its 5, SIMD
// 3 has 4 and adds the shared row tile's units like the balanced form) and loads its W fragments straight from the
// conv weight in the MFMA operand layout (16 filters x 64 B per instruction -- the shape the staging loads had).
// The wave then sweeps K once per PASS:
//   pass 1: EVERY wave all P row tiles x 2 of its column tiles (2 P tiles, 4 P per SIMD: perfectly balanced) --
//           the K loop of the forms above minus the B staging: per chunk and thread one table-row piece global ->
//           register -> LDS (all 128 rows), P operand reads, 2 weight-fragment loads, 8 P MFMAs; every chunk has its
//           own LDS region, so the barriers only publish data (no buffer is ever reused);
//   pass 2: no staging and NO barrier -- the waves run free over the resident chunks: the three-column waves finish
//           their third column (P tiles), SIMD 3's waves compute the shared row tile's units (4 | 3), the two-column
//           waves store their results and leave.  Pass 1's results are stored straight from the accumulators (float4
//           per tile: the operand roles are swapped, see MFMA4S), ONE TILE PER K STEP of pass 2: a wave's loads wait
//           behind its older stores (one in-order counter), so a burst of stores ahead of the loop stalled every wave
//           until its last store was acknowledged, while one store per step has a K step's time to land.
// Only pass 2's share of the output (a fifth at P = 7) is still written in the final burst.  Same K order per output
// element as the other forms: identical bits.  P = 4 .. 7 (the balanced form's plan generalised: the smallest P whose
// plan applies puts the most CUs to work -- cfg4's 1,234 row tiles run as 246 workgroups of 5 instead of 176 of 7).
// (Measured alternatives: DESIGN.md Appendix A, docs/DESIGN_rounds1-4.md 4.1b.)
//
// weight operand x table operand: the lane ends up with 4 consecutive COLUMNS of one table row (same fma chain as
// the other forms' table x weight order: identical bits), which it stores as one float4
#define MFMA4S(tab, wgt, c) __builtin_amdgcn_mfma_f32_16x16x4f32(wgt, tab, c, 0, 0, 0)
constexpr int AR_ROWS = 128;                       // LDS rows per K chunk
constexpr int AR_CHUNK = AR_ROWS * PEC;            // floats per chunk region
constexpr int AR_MAX_CHUNKS = 20;                  // 20 x 8 KB = 160 KB: E <= 320
constexpr int AR_PMIN = 4, AR_PMAX = 7;  // private row tiles per workgroup

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct AresPlan { int tower, P, row0, sh_row0, sh_c0, sh_n; };
struct AresFit { int rt[MAX_TOWERS], g[MAX_TOWERS], U, P; };

// gemm7_plan for P private row tiles per workgroup on `cap` workgroups: G = min(U / P, cap) of them split over the
// towers in proportion to their row tiles (workgroups the flooring leaves over go where the most row tiles are left),
// the left-over row tiles of a tower each shared column-wise by >= SH of its workgroups (SH = 3: at most 7 column
// units per sharer, all on SIMD 3; SH = 2: up to 10, the three units past the seventh on the two-column waves of
// SIMDs 0 - 2).  A pure function of the towers' distinct-token counts, evaluated by every workgroup.
// (loops over the towers are unrolled to MAX_TOWERS with a guard: a run-time trip count made hipcc index rt[] / g[] dynamically,
// i.e. keep them in scratch memory -- the plan is evaluated by every workgroup before its first load)
__device__ __forceinline__ bool ares_fit(const ProjArgs &a, AresFit &f, int P, int cap, int SH) {
    int G = f.U / P;
    if (G > cap) G = cap;
    if (G < a.ntower) return false;
    int used = 0;
    _Pragma("unroll") for (int t = 0; t < MAX_TOWERS; ++t) if (t < a.ntower) {
        f.g[t] = f.rt[t] * G / f.U;                         // (products < 2^24: 32-bit arithmetic)
        if (f.g[t] < 1) return false;
        used += f.g[t];
    }
    for (; used < G; ++used) {                              // (fewer than ntower of them)
        int best = -1, most = 0;
        _Pragma("unroll") for (int t = 0; t < MAX_TOWERS; ++t) if (t < a.ntower) {
            const int left = f.rt[t] - P * (f.g[t] + 1);
            if (left >= 0 && f.rt[t] - P * f.g[t] > most) { most = f.rt[t] - P * f.g[t]; best = t; }
        }
        if (best < 0) break;
        _Pragma("unroll") for (int t = 0; t < MAX_TOWERS; ++t) f.g[t] += (t == best && t < a.ntower);
    }
    _Pragma("unroll") for (int t = 0; t < MAX_TOWERS; ++t) if (t < a.ntower) {
        const int left = f.rt[t] - P * f.g[t];              // row tiles nobody owns: shared, >= SH workgroups each
        if (left < 0 || SH * left > f.g[t]) return false;
    }
    f.P = P;
    return true;
}

// The plan of a launch on `nwg` persistent workgroups: the smallest P that fits -- the most workgroups at work; past
// 7 1/3 row tiles per workgroup, 7 private ones and half a shared one (30,720 rows on 256 workgroups).
__device__ __forceinline__ bool ares_plan_fit(const ProjArgs &a, int nwg, AresFit &f) {
    f.U = 0;
    for (int t = 0; t < MAX_TOWERS; ++t) f.rt[t] = 0;
    int tiles128 = 0;
    _Pragma("unroll") for (int t = 0; t < MAX_TOWERS; ++t) if (t < a.ntower) {
        const int cnt = a.t[t].count[0];
        f.rt[t] = (cnt + 15) >> 4;
        f.U += f.rt[t];
        tiles128 += (cnt + PM - 1) / PM;
    }
    if (f.U == 0 || f.U >= (1 << 15)) return false;
    const int grid = nwg < G7_WGS ? nwg : G7_WGS;
    // (a launch whose 128-row tiles fill at most half of the grid is faster in the tile form's column parts:
    // B = 32 / 48: 35 / 36 us against 40 with 4 row tiles per workgroup here)
    if (tiles128 * 2 <= grid) return false;
    for (int P = AR_PMIN; P <= AR_PMAX; ++P)
        if (ares_fit(a, f, P, grid, 3)) return true;
    return ares_fit(a, f, AR_PMAX, grid, 2);
}

// workgroup wg's share of the plan (p.tower < 0: nothing)
__device__ __forceinline__ void ares_assign(const ProjArgs &a, const AresFit &f, int wg, AresPlan &p) {
    const int P = f.P;
    p.tower = -1;
    p.P = P;
    int base = 0;
    _Pragma("unroll") for (int t = 0; t < MAX_TOWERS; ++t) if (t < a.ntower) {
        if (wg >= base && wg < base + f.g[t]) {
            const int wl = wg - base, left = f.rt[t] - P * f.g[t];
            p.tower = t;
            p.row0 = wl * P * 16;
            p.sh_row0 = -1; p.sh_c0 = 0; p.sh_n = 0;
            if (left > 0) {
                const int lt = wl * left / f.g[t];                                 // the shared row tile of this workgroup
                const int w_lo = (lt * f.g[t] + left - 1) / left;                   // its sharers: [w_lo, w_hi)
                const int w_hi = ((lt + 1) * f.g[t] + left - 1) / left;
                const int n = w_hi - w_lo, k = wl - w_lo;
                p.sh_row0 = (P * f.g[t] + lt) * 16;
                p.sh_c0 = PNT * k / n;
                p.sh_n = PNT * (k + 1) / n - p.sh_c0;                               // <= ceil(19 / 3) = 7 (10 with two sharers)
            }
        }
        base += f.g[t];
    }
}

// 16-byte slot of logical float4 column q in a 64-byte chunk row: q ^ g((row >> 2) & 3), g = [0, 3, 2, 1].  A
// ds_read_b128 lane group holds rows {a, a+4, a+8, a+12} x two q values per residue a = row & 3 (bank slot =
// 4 a + physical column): this g makes the four physical columns distinct; a ds_write_b128 lane group (two
// consecutive rows x 4 columns, 32 banks) is conflict-free by construction.
__device__ __forceinline__ int ar_col(int row, int q) { return q ^ ((4 - ((row >> 2) & 3)) & 3); }


struct AresCtx {
    const float *aptr;             // this thread's table row (staging)
    float *st_dst;                 // ... and where its piece lands inside a chunk region
    const float *lds;
    int E, nchunk, c4, q, a_off;
};

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int NR>
struct AresA { f32x4 a[NR]; };
template <int NCW>
struct AresB { f32x4 b[NCW]; };

// One pass: resident row tiles rt0 .. rt0 + NR - 1 x the NCW column tiles whose weight rows `bptr` names.  S > 0: this
// pass also brings the A rows in, S chunks per barrier (pass 1).  `store_prev(k)`, k = 0 .. NPS - 1 (compile-time),
// stores one tile of the PREVIOUS pass's results; one is issued per K step of this pass.  Results stay in acc.
template <int NR, int NCW, int S, int NPS, typename F>
__device__ __forceinline__ void ares_pass(const AresCtx &x, int rt0, const float *const (&bptr)[NCW],
                                          f32x4 (&acc)[NR][NCW], F store_prev, const AresB<NCW> *pre = nullptr) {
    constexpr int SS = S > 0 ? S : 1;
    // Operand registers: the table fragments (LDS) are double-buffered; the weight fragments come from L2 and are
    // requested DB chunks ahead into a ring of DB + 1 sets.  (A timing ablation with the free pass's weight loads
    // aimed at one hot line took pass 2 from 11.7 to 8.8 us; requesting them 3 chunks ahead instead of 1 did NOT
    // -- 12.3 us: what those loads cost is their 16 lines per instruction in the address path, not their latency.)
    constexpr int DB = 1, RB = DB + 1;                     // (3 chunks ahead measured no faster in either pass)
    const int E = x.E, nchunk = x.nchunk;
    f32x4 ar[SS];
    auto ld_a = [&](int s, f32x4 (&r)[SS]) {                 // super-chunk s = chunks s S .. s S + S - 1 (past the end: the last chunk again)
#pragma unroll
        for (int k = 0; k < SS; ++k) {
            const int c = min(s * SS + k, nchunk - 1);
            r[k] = *reinterpret_cast<const f32x4 *>(x.aptr + min(c * PEC + x.c4 * 4, E - 4));
        }
    };
    auto st_a = [&](int s, const f32x4 (&r)[SS]) {           // the K tail (E % 16) is zeroed on this side
#pragma unroll
        for (int k = 0; k < SS; ++k) {
            const int c = min(s * SS + k, nchunk - 1);
            const bool keep = c * PEC + x.c4 * 4 < E;
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = keep ? r[k][i] : 0.f;
            *reinterpret_cast<f32x4 *>(x.st_dst + c * AR_CHUNK) = v;
        }
    };
    auto req_b = [&](int c, AresB<NCW> &o) {                 // weight fragments: global -> registers
        const int e = min(min(c, nchunk - 1) * PEC + x.q * 4, E - 4);
#pragma unroll
        for (int j = 0; j < NCW; ++j) o.b[j] = *reinterpret_cast<const f32x4 *>(bptr[j] + e);
    };
    auto req_a = [&](int c, AresA<NR> &o) {                  // table fragments: resident LDS -> registers
        const float *ab = x.lds + min(c, nchunk - 1) * AR_CHUNK + x.a_off;
#pragma unroll
        for (int i = 0; i < NR; ++i) o.a[i] = *reinterpret_cast<const f32x4 *>(ab + (rt0 + i) * 16 * PEC);
    };
    auto mfma = [&](const AresA<NR> &oa, const AresB<NCW> &ob) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < NR; ++i)
#pragma unroll
                for (int j = 0; j < NCW; ++j) acc[i][j] = MFMA4S(oa.a[i][kk], ob.b[j][kk], acc[i][j]);
    };
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int j = 0; j < NCW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    AresA<NR> oa[2];
    AresB<NCW> ob[RB];
    if (S > 0) {
        ld_a(0, ar);
        static_for<0, DB>([&](auto kc) { req_b(decltype(kc)::value, ob[decltype(kc)::value]); });   // weight fragments of the first chunk(s): in flight under the staging
        st_a(0, ar);
        ld_a(1, ar);
        __syncthreads();                                     // super-chunk 0 in LDS
        req_a(0, oa[0]);
        st_a(1, ar);
        ld_a(2, ar);
        PRO_STAMP
    } else {
        // (`pre`: chunk 0's weight fragments were requested long ago -- under the previous pass -- by the caller:
        // the free pass then starts on its MFMAs instead of an L2 round trip with an idle matrix pipe)
        static_for<0, DB>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if (k == 0 && pre) ob[0] = *pre;
            else req_b(k, ob[k]);
        });
        req_a(0, oa[0]);
    }
    // chunk c (ring position r = c mod lcm(2, RB), compile-time): its operands are in oa[r & 1] / ob[r % RB].  HEAD
    // (first chunk of super-chunk s, staging pass): behind the barrier super-chunk s + 1 is visible, the staging
    // registers (super-chunk s + 2, requested a super-step ago) go to LDS and super-chunk s + 3 is requested.  One
    // memory instruction behind every few MFMAs, like the forms above.
    auto step = [&](int c, auto rc, auto headc, auto storec) {
        constexpr int r = decltype(rc)::value;
        constexpr bool head = decltype(headc)::value;
        constexpr int store_k = decltype(storec)::value;        // >= 0: this step also issues store_prev(store_k)
        // (the table-row loads go out BEHIND the weight-fragment loads: the next step waits for those with a
        // counted vmcnt, which an older, slower load -- table rows come from the Infinity Cache or HBM -- would hold up)
        if (S > 0 && head) __syncthreads();
        req_b(c + DB, ob[(r + DB) % RB]);
        req_a(c + 1, oa[(r + 1) & 1]);
        if constexpr (store_k >= 0) store_prev(storec);
        if (S > 0 && head) {
            st_a(c / SS + 2, ar);
            ld_a(c / SS + 3, ar);
        }
        mfma(oa[r & 1], ob[r % RB]);
        constexpr int NLD = NCW + ((S > 0 && head) ? SS : 0), NWR = (S > 0 && head) ? SS : 0;
        constexpr int NM = 4 * NR * NCW;
        constexpr int NMEM_MAX = NR + NCW + 2 * SS + 1;
        constexpr int GAP = NM / NMEM_MAX >= 4 ? 4 : (NM / NMEM_MAX >= 1 ? NM / NMEM_MAX : 1);
#pragma unroll
        for (int i = 0; i < NCW; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, GAP, 0);
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, GAP, 0);
        }
        if constexpr (store_k >= 0) {
            __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, GAP, 0);
        }
        if (S > 0 && head) {
#pragma unroll
            for (int i = 0; i < SS; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, GAP, 0);
            }
#pragma unroll
            for (int i = 0; i < SS; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, GAP, 0);
            }
        }
        constexpr int REST = NM - GAP * (NR + NLD + NWR + (store_k >= 0 ? 1 : 0));
        if constexpr (REST > 0) __builtin_amdgcn_sched_group_barrier(0x008, REST, 0);
    };
    // U chunks per loop iteration: whole super-chunks and whole turns of both operand rings
    using NoSt = std::integral_constant<int, -1>;
    static_assert(SS == 1 || SS == 2 || SS == 4, "chunks per barrier");
    static_assert(RB == 2 || RB == 4, "weight-fragment ring of 2 or 4 sets");
    static_assert(NPS == 0 || S == 0, "the staging pass has no previous pass to store");
    constexpr int U = (SS == 4 || RB == 4) ? 4 : 2;
    int c = 0;
    if constexpr (NPS > 0) {
        // the first NPS steps, peeled: step k carries store k of the previous pass (the accumulator index must be a
        // compile-time constant).  Launches with fewer chunks than stores (small E) issue them all up front.
        constexpr int NPE = (NPS + U - 1) / U * U;           // whole turns: the loop below starts at ring position 0
        if (nchunk >= NPE) {
            static_for<0, NPE>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                step(k, std::integral_constant<int, k % U>{}, std::false_type{}, std::integral_constant<int, (k < NPS ? k : -1)>{});
            });
            c = NPE;
        } else {
            static_for<0, NPS>([&](auto kc) { store_prev(kc); });
        }
    }
    for (; c + U <= nchunk; c += U)
        static_for<0, U>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            step(c + k, kc, std::bool_constant<k % SS == 0>{}, NoSt{});
        });
    // the last, partial turn (peeled: an exit inside the loop body makes hipcc wait vmcnt(0) at its head)
    static_for<0, U - 1>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if (c + k < nchunk) step(c + k, kc, std::bool_constant<k % SS == 0>{}, NoSt{});
    });
}

// P: private row tiles of the workgroup; NCW: column tiles of this wave (from ct0); NEX: units of the shared row tile
// it adds in pass 2 (SIMD 3: 4 | 3, from eoff; the two-column waves of SIMDs 0 - 2: one of units 7 .. 9)
template <int P, int NCW, int NEX>
__device__ __forceinline__ void proj_gemm_ares_body(const ProjArgs &a, float *lds, const AresPlan &p, int ct0, int eoff) {
    constexpr int NE = NEX > 0 ? NEX : 1;
    const ProjTower &tw = a.t[p.tower];
    const int count = tw.count[0];
    TRACE_STAMP(0)
    TRACE_WHERE(0xf)
    const int tid = threadIdx.x, lane = tid & 63, lrow = lane & 15;
    AresCtx x;
    x.lds = lds; x.E = a.E; x.nchunk = a.nchunk;
    x.q = lane >> 4;
    x.a_off = lrow * PEC + ar_col(lrow, x.q) * 4;
    // staging role: float4 column c4 of LDS row tid >> 2 (rows 0 .. 16 P - 1 private, the next 16 the shared row tile;
    // threads beyond them re-stage its last row: unconditional loads, nothing reads what they write)
    x.c4 = tid & 3;
    const int srow = tid >> 2;
    int grow = srow < P * 16 ? p.row0 + srow : (p.sh_row0 < 0 ? 0 : p.sh_row0 + min(srow - P * 16, 15));
    grow = grow < count ? grow : count - 1;
    x.aptr = a.table + (long)tw.list[grow] * a.E;
    x.st_dst = lds + srow * PEC + ar_col(srow, x.c4) * 4;
    // weight fragments: lane (lrow, q) of column tile ct holds W[f][j][16 c + 4 q ..], n = 16 ct + lrow = 100 j + f
    const float *__restrict__ conv_w = tw.conv_w;
    auto wrow = [&](int ct) {
        const int n = ct * 16 + lrow;
        const int j = n / PF, f = n - j * PF;
        return conv_w + (n < PROW ? ((long)f * 3 + j) * a.E : 0);       // padding columns re-read W[0][0]: never stored
    };
    // Stores go through a buffer descriptor over this tower's projected rows (wave-uniform words): branch-free (a
    // branch would end the K step's scheduling region) -- rows past `count`, padding columns and unassigned shared
    // units get an offset outside the descriptor, which the hardware drops.  32-bit byte offsets: the launcher keeps
    // form 3 to outputs under 2 GB.
    const unsigned long long pbase = reinterpret_cast<unsigned long long>(tw.ptab);
    const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pbase), phi = __builtin_amdgcn_readfirstlane((unsigned)(pbase >> 32));
    float *pt = reinterpret_cast<float *>(((unsigned long long)phi << 32) | plo);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(pt, 0, __builtin_amdgcn_readfirstlane(count) * (PSTR * 4), 0x00020000);
    auto store_tile = [&](const f32x4 &v, int row_first, int ct, bool ok = true) {
        const int row = row_first + lrow, col = ct * 16 + x.q * 4;
        const int off = (ok && row < count && col < PROW) ? (row * PSTR + col) * 4 : 0x7ffffff0;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, off, 0, 0);
    };
    // pass 2's first weight fragments: requested before pass 1
    [[maybe_unused]] auto first_frag = [&](const float *wr) {
        return *reinterpret_cast<const f32x4 *>(wr + min(x.q * 4, a.E - 4));
    };
    [[maybe_unused]] AresB<1> pre3;
    [[maybe_unused]] AresB<NE> prex;
    if constexpr (NCW == 3) pre3.b[0] = first_frag(wrow(ct0 + 2));
    else if constexpr (NEX > 0) {
#pragma unroll
        for (int j = 0; j < NE; ++j) prex.b[j] = first_frag(wrow(min(p.sh_c0 + eoff + j, PNT - 1)));
    }
    // ---- pass 1: all P row tiles x the wave's first two column tiles; stages the A rows
    const float *b1[2] = {wrow(ct0), wrow(ct0 + 1)};
    f32x4 a1[P][2];
    TRACE_CLK(clk0)
    ares_pass<P, 2, 1, 0>(x, 0, b1, a1, [](auto) {});      // (one chunk staged per barrier: 2 or 4 measured the same)
    TRACE_STAMP(1)
    TRACE_CLK(clk1)
    TRACE_CLK_SPAN(7, clk1 - clk0)                           // shader cycles of pass 1
    __syncthreads();                                         // every chunk of every row is in LDS: the waves run free from here
    auto store1 = [&](auto kc) {                             // store k of pass 1's results: tile (k / 2, k % 2)
        constexpr int k = decltype(kc)::value;
        store_tile(a1[k / 2][k % 2], p.row0 + (k / 2) * 16, ct0 + k % 2);
    };
    // ---- pass 2
    if constexpr (NCW == 3) {                                // the third column tile
        const float *b2[1] = {wrow(ct0 + 2)};
        f32x4 a2[P][1];
        ares_pass<P, 1, 0, P * 2>(x, 0, b2, a2, store1, &pre3);
        TRACE_STAMP_LAST(2)
#pragma unroll
        for (int i = 0; i < P; ++i) store_tile(a2[i][0], p.row0 + i * 16, ct0 + 2);
    } else if constexpr (NEX > 0) {                          // the shared row tile (resident row tile P) x this wave's units
        const float *be[NE];
        int ecol[NE];
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            ecol[j] = min(p.sh_c0 + eoff + j, PNT - 1);      // beyond the assignment: computed, never stored
            be[j] = wrow(ecol[j]);
        }
        f32x4 ax[1][NE];
        ares_pass<1, NE, 0, P * 2>(x, P, be, ax, store1, &prex);
        TRACE_STAMP_LAST(2)
#pragma unroll
        for (int j = 0; j < NE; ++j) store_tile(ax[0][j], p.sh_row0, ecol[j], eoff + j < p.sh_n && p.sh_row0 >= 0);
    } else {
        static_for<0, P * 2>([&](auto kc) { store1(kc); });  // nothing left to compute: store and leave
        TRACE_STAMP_LAST(2)
    }
    TRACE_CLK_SPAN(6, ((__builtin_readcyclecounter() - clk1) << 1) | 1)     // shader cycles of wave 0's pass 2 (bit 0 = active)
    TRACE_STAMP_LAST(3)
}

// the four wave roles of the A-resident form: waves w / w + 4 take 3 / 2 column tiles of SIMD w & 3's five
template <int P>
__device__ __forceinline__ void proj_gemm_ares_wg(const ProjArgs &a, float *lds, const AresPlan &p) {
    const int wave = threadIdx.x >> 6, simd = wave & 3;
    if (wave < 4) {
        if (simd == 3) proj_gemm_ares_body<P, 2, 4>(a, lds, p, 15, 0);
        else proj_gemm_ares_body<P, 3, 0>(a, lds, p, simd * 5, 0);
    } else {
        if (simd == 3) { proj_gemm_ares_body<P, 2, 3>(a, lds, p, 17, 4); return; }
        if constexpr (P == AR_PMAX)                          // units 7 .. 9 of a row tile shared by two workgroups only
            if (p.sh_n > 7 + simd) { proj_gemm_ares_body<P, 2, 1>(a, lds, p, simd * 5 + 3, 7 + simd); return; }
        proj_gemm_ares_body<P, 2, 0>(a, lds, p, simd * 5 + 3, 0);
    }
}

__global__ __launch_bounds__(GEMM_THREADS) void proj_gemm_kernel(ProjArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lds = reinterpret_cast<float *>(smem);
    if (a.balanced == 3) {
        // One round only.  (Several rounds of tiles per workgroup -- 45 k rows as 474 tiles of 6 in two rounds -- were
        // built and measured: 104 us against the tile form's 88 for unpadded documents, 155 against 129 for uniform
        // words, and the round loop around the tile bodies cost the headline launch 6 us in register allocation.)
        AresFit f;
        AresPlan p;
        if (ares_plan_fit(a, (int)gridDim.x, f)) {
            ares_assign(a, f, (int)blockIdx.x, p);
            if (p.tower < 0) return;                         // uniform: this workgroup has no rows
            switch (p.P) {
                case 4: proj_gemm_ares_wg<4>(a, lds, p); break;
                case 5: proj_gemm_ares_wg<5>(a, lds, p); break;
                case 6: proj_gemm_ares_wg<6>(a, lds, p); break;
                default: proj_gemm_ares_wg<7>(a, lds, p); break;
            }
            return;
        }
    }
    // (towers unrolled to MAX_TOWERS and first[] searched by selects: run-time indices would put the array in scratch memory)
    int first[MAX_TOWERS + 1];
    first[0] = 0;
#pragma unroll
    for (int t = 0; t < MAX_TOWERS; ++t) first[t + 1] = first[t] + (t < a.ntower ? (a.t[t].count[0] + PM - 1) / PM : 0);
    auto tower_of = [&](int tile, int &row0) {
        int t = 0, f0 = 0;
#pragma unroll
        for (int k = 1; k < MAX_TOWERS; ++k)
            if (k < a.ntower && tile >= first[k]) { t = k; f0 = first[k]; }
        row0 = (tile - f0) * PM;
        return t;
    };
    // (a launch whose tiles fill at most half of the grid is faster in column parts -- below -- than in the
    // balanced form, which keeps 112 rows per workgroup whatever the count)
    if (a.balanced == 1 && first[MAX_TOWERS] * 2 > (int)gridDim.x) {
        Gemm7Plan p;
        if (gemm7_plan(a, (int)blockIdx.x, (int)gridDim.x, p)) {
            if (p.tower < 0) return;                         // uniform: this workgroup has no rows
            const int wave = threadIdx.x >> 6;
            if (wave < 4) {                                  // HALF 1 (17 tiles), three B rows to stage
                if ((wave & 3) == 3) proj_gemm7_body<4, 1, 3, 3>(a, lds, p);
                else proj_gemm7_body<5, 1, 0, 3>(a, lds, p);
            } else {                                         // HALF 0 (18 tiles), two B rows to stage
                if ((wave & 3) == 3) proj_gemm7_body<4, 0, 4, 2>(a, lds, p);
                else proj_gemm7_body<5, 0, 0, 2>(a, lds, p);
            }
            return;
        }
    }
    // tile form, persistent: the 128-row tiles of all towers form one list; workgroup w takes tiles
    // w, w + gridDim.x, ... (a grid of one workgroup per CU: nothing is launched only to exit)
    // The LAST round of the list is usually partial -- 536 tiles on 256 workgroups are two full rounds and 24
    // tiles, for which 232 workgroups used to wait a whole tile time.  When it fills at most half (a quarter) of
    // the grid, each of its tiles is cut into 2 (4) COLUMN PARTS (10 + 9, or 5 + 5 + 5 + 4 column tiles) that as
    // many workgroups compute side by side, every wave on its own 16 rows: same per-element summation order,
    // same bits, and the round costs 0.55 (0.3) of a tile time.
    const int total = first[MAX_TOWERS], G = (int)gridDim.x;
    const int full = total / G * G, tail = total - full;
    const int parts = a.balanced == 2 ? 1 : (tail * 4 <= G ? 4 : (tail * 2 <= G ? 2 : 1));   // (2: whole tiles only -- A/B runs, tests)
    for (int tile = blockIdx.x; tile < (parts == 1 ? total : full); tile += gridDim.x) {
        int row0;
        const int t = tower_of(tile, row0);
        if (threadIdx.x >> 8) proj_gemm_body<PNT - PNH, 2>(a, lds, t, row0);   // waves 4..7: columns 160..303
        else proj_gemm_body<PNH, 3>(a, lds, t, row0);                          // waves 0..3: columns 0..159
        __syncthreads();                                    // every wave is done with the LDS buffers (operand reads, epilogue slabs)
    }
    if (parts > 1 && (int)blockIdx.x < tail * parts) {
        const int tile = full + (int)blockIdx.x / parts, part = (int)blockIdx.x % parts;
        int row0;
        const int t = tower_of(tile, row0);
        if (parts == 2) {
            if (part == 0) proj_gemm_body<PNH, 2, 1>(a, lds, t, row0, 0);
            else proj_gemm_body<PNT - PNH, 2, 1>(a, lds, t, row0, PNH_COLS);
        } else {
            if (part < 3) proj_gemm_body<5, 1, 1>(a, lds, t, row0, part * 80);
            else proj_gemm_body<4, 1, 1>(a, lds, t, row0, 240);
        }
    }
}

// ---- 2d. the weight-resident form (form 4) for narrow tables (E <= 128: NARRE / TransNet at E = 64).
//
// At E = 64 the GEMM has four K chunks: 0.76 GFLOP against 24 MB of output at cfg4 -- the A-resident form above is
// then all prologue (stage the rows, barrier per chunk) and epilogue (its stores come in one burst), 19.4 us for 6 us
// of MFMA work.  Here the roles are turned round: the WEIGHTS of the workgroup's tower -- all 304 x E of them, 78 KB
// at E = 64 -- go to LDS once (every workgroup reads the same lines: L2 hits), and after ONE barrier the eight waves
// run free over units of (16-row tile) x (four column tiles): the table fragments of a unit are 4 nchunk floats per
// lane straight from global memory into registers (requested one unit ahead: the loads of unit k + 1 are older than
// the stores of unit k in the wave's in-order counter, so waiting for them does not wait for those stores), the
// weight fragments come from LDS (rows padded by one 16-byte slot: an odd slot stride spreads the 16 lanes of a
// ds_read_b128 group over all 64 banks), 16 nchunk MFMAs per unit on four independent accumulators, four float4
// stores.  Units go round-robin over the waves of the tower's workgroups (split over the towers in proportion to
// their row tiles), so the tail is one unit, and a launch stores steadily from its first microsecond.  Same operand
// arrangement and K order per output element as the other forms: identical bits.
constexpr int WR_THREADS = 512, WR_WAVES = WR_THREADS / 64;
constexpr int WR_MAX_CHUNKS = 8;                       // 304 rows x (128 + 4) floats = 160,512 B
constexpr int WR_NQ = (PNT + 3) / 4;                   // 5 column quads (the last one: 3 tiles)
__host__ __device__ constexpr int wr_stride(int nch) { return nch * PEC + 4; }
// LDS rows of the weight image: the 300 real ones only.  Rows 300 .. 303 (N padded to 19 tiles of 16) are read by lanes
// whose outputs are never stored -- they read row 299 instead -- and without them the image of a table of E <= 64 is
// 81,600 B: TWO workgroups per CU (2 x 81,920 allocated = the CU's 160 KB), four waves per SIMD instead of two.
#ifndef R4R_WRES_ROWS
#define R4R_WRES_ROWS PROW
#endif
constexpr int WR_ROWS = R4R_WRES_ROWS;
__host__ __device__ constexpr int wr_lds_bytes(int nch) { return WR_ROWS * wr_stride(nch) * 4; }
__host__ __device__ constexpr int wr_wgs_per_cu(int nch) { return 2 * wr_lds_bytes(nch) <= 160 * 1024 - 2 * 256 ? 2 : 1; }

template <int NCH>
__global__ __launch_bounds__(WR_THREADS) void proj_gemm_wres_kernel(ProjArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lds = reinterpret_cast<float *>(smem);
    constexpr int STR = wr_stride(NCH), SLOTS = NCH * 4;
    // the towers' workgroups: in proportion to their row tiles, every tower with rows at least one
    int rt[MAX_TOWERS], g[MAX_TOWERS], U = 0, live = 0;
    // the towers' row counts in ONE round trip: lane t reads tower t's (as scalar loads, one per tower and each behind a
    // branch, they were a chain of dependent round trips in front of everything else)
    static_assert(MAX_TOWERS == 4, "the lane -> tower select below");
    const int ln = threadIdx.x & 63;
    const int *cp = ln == 0 ? a.t[0].count : (ln == 1 ? a.t[1].count : (ln == 2 ? a.t[2].count : a.t[3].count));
    const int cl = ln < a.ntower ? cp[0] : 0;
    for (int t = 0; t < MAX_TOWERS; ++t) {
        rt[t] = (__builtin_amdgcn_readlane(cl, t) + 15) >> 4;
        U += rt[t];
        live += rt[t] > 0;
    }
    if (U == 0) return;
    const int G = (int)gridDim.x;                            // (the launcher gives every tower a workgroup: G >= ntower)
    int used = 0, big = 0;
    for (int t = 0; t < MAX_TOWERS; ++t) {
        g[t] = rt[t] > 0 ? max(1, (int)((unsigned)rt[t] * (unsigned)(G - live) / (unsigned)U)) : 0;   // (row tiles < 2^20, workgroups < 2^10: 32 bits)
        used += g[t];
        if (rt[t] > rt[big]) big = t;
    }
    for (int t = 0; t < MAX_TOWERS; ++t) g[t] += (t == big) ? G - used : 0;     // (>= 0: the floors sum to <= G - live + live)
    int tower = -1, wl = 0, base = 0;
    for (int t = 0; t < MAX_TOWERS; ++t) {
        if ((int)blockIdx.x >= base && (int)blockIdx.x < base + g[t]) { tower = t; wl = (int)blockIdx.x - base; }
        base += g[t];
    }
    if (tower < 0) return;
    const ProjTower &tw = a.t[tower];
    const int count = tw.count[0], E = a.E;
    const int tid = threadIdx.x, lane = tid & 63, lrow = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // (a scalar: the unit loop below branches on scalars)
    // (32-bit byte offsets into this tower's projected rows: the launcher keeps the form to outputs under 2 GB)
    const unsigned long long pbase = reinterpret_cast<unsigned long long>(tw.ptab);
    const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pbase), phi = __builtin_amdgcn_readfirstlane((unsigned)(pbase >> 32));
    float *pt = reinterpret_cast<float *>(((unsigned long long)phi << 32) | plo);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(pt, 0, __builtin_amdgcn_readfirstlane(count) * (PSTR * 4), 0x00020000);
    const int units = rt[tower] * WR_NQ, stride = g[tower] * WR_WAVES;
    // The table fragments of a unit's row tile take two dependent reads -- the token of the lane's row, then the row --
    // so they run NB units ahead through a ring of NB register sets, the tokens 2 NB units ahead (NB = 1 .. 5 measured:
    // no difference at cfg5, where the rows come from HBM -- the launch is bound by its stores --, and a longer
    // prologue at cfg4; NB = 1).  Rows past the end read row count - 1 and are never stored; units past the end
    // re-read the last unit's rows and are not computed.
    constexpr int NB = 1;
    auto load_tok = [&](int u) { return tw.list[min((min(u, units - 1) / WR_NQ) * 16 + lrow, count - 1)]; };
    auto load_a = [&](int tok, f32x4 (&o)[NCH]) {
        const float *ap = a.table + (long)tok * E;
#pragma unroll
        for (int c = 0; c < NCH; ++c) o[c] = *reinterpret_cast<const f32x4 *>(ap + min(c * PEC + q * 4, E - 4));
    };
    // (the K tail, E % 16, lies in the last chunk only: zeroed on this side too when the fragment is used)
    const bool tail_keep = (NCH - 1) * PEC + q * 4 < E;
    int u = wl * WR_WAVES + wave;
    f32x4 ring[NB][NCH];
    int tokr[NB];
    // ---- prologue, ordered by what depends on what: the first units' tokens are requested FIRST, then the weights
    // (one round trip to L2: every load of the fill is issued before the first LDS write), then -- the tokens are
    // older in the wave's in-order counter: waiting for them does not wait for the weights -- the first units' rows;
    // only then are the weights written to LDS (row n = 100 j + f holds W[f][j][:], zero past E and past row 299).
    const bool has_units = u < units;
    if (has_units) {
#pragma unroll
        for (int j = 0; j < NB; ++j) tokr[j] = load_tok(u + j * stride);
    }
    const float *__restrict__ conv_w = tw.conv_w;
    constexpr int NFILL = (PN * SLOTS + WR_THREADS - 1) / WR_THREADS;
    f32x4 wv[NFILL];
    bool wreal[NFILL];
#pragma unroll
    for (int k = 0; k < NFILL; ++k) {
        const int i = tid + k * WR_THREADS;
        const int n = i / SLOTS, sl = i - n * SLOTS;
        const int j = n / PF, f = n - j * PF;
        wreal[k] = n < PROW && sl * 4 < E;
        wv[k] = *reinterpret_cast<const f32x4 *>(conv_w + (wreal[k] ? ((long)f * 3 + j) * E + sl * 4 : 0));
    }
    // Entry state of the unit loop: the queue the loop has on its back edge -- per ring set: table-fragment loads, a
    // token load, four stores (here to an offset outside the descriptor: dropped by the hardware).  hipcc merges the
    // entry and back-edge states at the loop head; with nothing behind the loads on the entry side it waited
    // vmcnt(0) in every iteration, i.e. for the previous unit's stores to be acknowledged, where the loads alone are
    // what the data needs.
    if (has_units) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            load_a(tokr[j], ring[j]);
            tokr[j] = load_tok(u + (NB + j) * stride);
        }
    }
#pragma unroll
    for (int k = 0; k < NFILL; ++k) {
        const int i = tid + k * WR_THREADS;
        const int n = i / SLOTS, sl = i - n * SLOTS;
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = wreal[k] ? wv[k][c] : 0.f;
        if (i < PN * SLOTS && n < WR_ROWS) *reinterpret_cast<f32x4 *>(lds + n * STR + sl * 4) = v;
    }
    if (has_units) {
        const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4 * NB; ++i)                     // (distinct offsets: identical stores are merged)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, z), rsrc, 0x7ffffff0 - 64 * i, 0, 0);
    }
    __syncthreads();                                         // the weights are in LDS: the waves run free from here
    const float *bl = lds + lrow * STR + q * 4;
    const float *blast = lds + min((PNT - 1) * 16 + lrow, WR_ROWS - 1) * STR + q * 4;   // the last column tile's rows
    auto unit = [&](int uu, f32x4 (&slot)[NCH], int &tok) {
        f32x4 cur[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) cur[c] = slot[c];
#pragma unroll
        for (int i = 0; i < 4; ++i) cur[NCH - 1][i] = tail_keep ? cur[NCH - 1][i] : 0.f;
        load_a(tok, slot);                                   // unit uu + NB stride
        tok = load_tok(uu + 2 * NB * stride);
        const int rt0 = (uu / WR_NQ) * 16, ct0 = (uu % WR_NQ) * 4;
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float *bu = bl + ct0 * 16 * STR;
        if (ct0 + 4 <= PNT) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                f32x4 b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const f32x4 *>(bu + i * 16 * STR + c * PEC);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i] = MFMA4S(cur[c][kk], b[i][kk], acc[i]);
            }
        } else {                                             // the last quad: three column tiles
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                f32x4 b[3];
#pragma unroll
                for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const f32x4 *>(bu + i * 16 * STR + c * PEC);
                b[2] = *reinterpret_cast<const f32x4 *>(blast + c * PEC);       // (rows 300 .. 303: row 299's, never stored)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int i = 0; i < 3; ++i) acc[i] = MFMA4S(cur[c][kk], b[i][kk], acc[i]);
            }
        }
        const int row = rt0 + lrow;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int col = (ct0 + i) * 16 + q * 4;
            const int off = (row < count && col < PROW) ? (row * PSTR + col) * 4 : 0x7ffffff0;   // out of range: dropped by the hardware
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i]), rsrc, off, 0, 0);
        }
    };
    // whole turns of the ring without a branch between the units (a skipped unit would leave a different queue
    // behind, and the merged state costs every iteration a full wait), then the last, partial turn
    for (; u + (NB - 1) * stride < units; u += NB * stride)
        static_for<0, NB>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            unit(u + j * stride, ring[j], tokr[j]);
        });
    static_for<0, NB - 1>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if (u + j * stride < units) unit(u + j * stride, ring[j], tokr[j]);
    });
}

// ---- 3. gather-add-max.  One workgroup = TWO consecutive 128-position segments of the launch's
// document-major segment list (grid = (ceil(N * tiles / 2), ntower)); its 8 workers of 32 lanes each walk a
// 32-position slice, so a worker's dependent chain is 34 tokens / 8 in flight = 5 memory
// round trips (it was 130 / 4 = 33: at batch 128 the kernel was pure latency).  Lane
// wl < 25 owns filters 4wl..4wl+3 (one float4 of each 400-byte tap row).  Token t completes
// position p = t (its tap-2 row), feeds tap 1 of p = t+1 and tap 0 of p = t+2.  The four
// slices of a segment are merged through LDS in position order (strict >, so the first
// maximum wins like PyTorch's max-pool).
constexpr int SLICE = 32;                 // positions per worker (16 -- three dependent rounds per worker instead of five, twice the workgroups -- measured slower)
constexpr int GWPS = SEG / SLICE;         // workers per segment (4 | 8)
constexpr int GSPW = 8 / GWPS;            // segments per workgroup (2 | 1)
static_assert(SLICE == 32 || SLICE == 16, "a worker is 32 lanes; eight workers per workgroup");
constexpr int GATHER_ROTDIV = 16;         // documents per rotation step (A/B at cfg5: 1 .. 64 all help, 16 most: 34.2 -> 30.8 us; cfg3: neutral)
constexpr int GDEPTH = 7;                 // tokens in flight per lane (7: 124 VGPRs, four waves per SIMD -- every workgroup of the
                                          // cfg3 launch resident at once; 8: 136 VGPRs, three, 1 us slower; 4..6 within 0.5 us of 7)

HEAD_TRACE_DEFINE(r4r_debug_gather_trace)
__global__ __launch_bounds__(256) void proj_gather_max_kernel(ProjArgs a) {
    HEAD_STAMP(0)
    __shared__ int sl[8][SLICE + 2];
    __shared__ float sbest[8][PF];
    __shared__ int sbp[8][PF];
    const ProjTower &tw = a.t[blockIdx.y];
    const int worker = threadIdx.x >> 5, wl = threadIdx.x & 31;
    // Which pair of segments this workgroup takes.  Document-major order puts segment pair k of every document on
    // workgroups k mod tiles/2 -- and the dispatcher places workgroups 8 apart on one XCD and (one generation of 1,024
    // resident workgroups on 256 CUs) 256 apart on one CU: with 4 pairs per 1000-word document every CU, and every
    // XCD, got four workgroups of the SAME pair index -- all heads of documents (real words, five dependent rounds)
    // or all zero-padded tails.  The pair index is rotated by document / GATHER_ROTDIV, so that a CU's workgroups, and
    // an XCD's, mix heads and tails.  Same work per workgroup, same outputs.  (What else was tried on this launch's
    // schedule and measured no faster -- segment pairs half a document apart, 16-position slices, a software-pipelined
    // group loop, hot rows staged in LDS, segments dealt to workgroups by cost: DESIGN.md 4.1b, profiles/r05_negatives.txt.)
    // (32-bit arithmetic throughout: the launcher keeps N * tiles under 2^31, and a 64-bit division by a run-time value is a
    // hundred instructions at the head of every workgroup's dependent chain)
    unsigned bx = blockIdx.x;
    if (GSPW == 2 && (a.tiles & 1) == 0 && a.tiles >= 4) {
        const unsigned t2 = (unsigned)a.tiles >> 1;
        const unsigned d = bx / t2, pr = bx - d * t2;
        bx = d * t2 + (pr + d / GATHER_ROTDIV) % t2;
    }
    // segment `unit` of the launch's N * tiles segments (document-major): workers 0-3 take the workgroup's first
    // segment, 4-7 its second -- of the same document, or (odd tile counts, e.g. NARRE's one-tile reviews) the first of
    // the next one
    const unsigned units = (unsigned)(a.N * a.tiles);
    auto unit_of = [&](int h) -> unsigned { return bx * GSPW + h; };
    const unsigned unit = unit_of(worker / GWPS);
    const unsigned udoc = unit < units ? unit / (unsigned)a.tiles : 0u;
    const int64_t doc = udoc;
    const int seg = unit < units ? (int)(unit - udoc * (unsigned)a.tiles) : a.tiles;     // a.tiles: no such segment
    const int T = a.T, P = T + 2;
    const bool act = wl < PF / 4;
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (blockIdx.x == 0 && threadIdx.x == 0) {              // consumed by the GEMM launch before this one
        tw.count[1] = tw.count[0];                          // ... but remembered: the host's measured conv rule reads it
        tw.count[0] = 0;
    }

    // A segment's positions are dealt to its GWPS workers in EQUAL slices: a full 128-position segment in slices of 32, a
    // shorter one -- NARRE's whole 102-position review, the 106-position last segment of a 1000-word document -- in slices of
    // ceil(len / GWPS) (26, 27), so that its last worker does not sit on 6 positions while the others walk 32 in five
    // dependent rounds (four now).  Which positions a slice holds changes nothing downstream: the slices are merged in
    // position order.
#ifndef R4R_GATHER_EVEN_SLICES
#define R4R_GATHER_EVEN_SLICES 1
#endif
    const int seg_len = min(SEG, P - seg * SEG);
    const int slen = (R4R_GATHER_EVEN_SLICES && seg_len > 0 && seg_len < SEG) ? (seg_len + GWPS - 1) / GWPS : SLICE;
    const int p_lo = seg * SEG + (worker % GWPS) * slen;
    const int p_hi = min(min(P, (seg + 1) * SEG), p_lo + slen);
    const int t_lo = p_lo - 2;
    const int ntok = (seg < a.tiles && p_hi > p_lo) ? p_hi - t_lo : 0;     // tokens t_lo .. p_hi-1
    // stage the slice's slots; at the same time find out whether all its tokens are the SAME row
    // (typically the zero-padded tail of a document, data.py:198-199): then every position of the
    // slice has the same window sum and its first position decides the slice (first-max wins).
    bool same = true;
    {
        // a slice is at most SLICE + 2 = 34 tokens: lane wl takes token wl and (wl < 2) token 32 + wl, both
        // requested together -- as a loop the second pass (two lanes) was two more dependent round trips
        const int ta = t_lo + wl, tb = t_lo + 32 + wl;        // (SLICE 16: 18 tokens, the second request is idle)
        const bool va = wl < ntok && ta >= 0 && ta < T, vb = 32 + wl < ntok && tb >= 0 && tb < T;
        int64_t ia = tw.idx[doc * T + (va ? ta : 0)], ib = tw.idx[doc * T + (vb ? tb : 0)];
        asm volatile("" : "+v"(ia), "+v"(ib));                // (both ids requested before either slot: hipcc had ordered id a, slot a, id b, slot b -- three round trips)
        const int sa = tw.slot[ia], sb = tw.slot[ib];
        if (wl < ntok) sl[worker][wl] = va ? sa : -1;
        if (32 + wl < ntok) sl[worker][32 + wl] = vb ? sb : -1;
    }
    __syncthreads();
    HEAD_STAMP(1)
    if (ntok > 0) {
        const int s_first = sl[worker][0];
        for (int k = wl; k < ntok; k += 32) same &= (sl[worker][k] == s_first);
    }
    // all 32 lanes of the worker (one half-wave) must agree
    {
        const unsigned long long m = __ballot(same);
        const unsigned long long mine = (worker & 1) ? (m >> 32) : (m & 0xffffffffull);
        same = (mine == 0xffffffffull);
    }
    const int npos_eff = same ? min(1, ntok - 2) : ntok - 2;   // positions actually walked

    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bp[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    if (ntok > 0 && act) {
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(tw.conv_b + wl * 4);
        const float *base = tw.ptab + wl * 4;
        f32x4 s_a = zero, s_b = zero;                        // partial sums of positions t and t+1
        // halo: tokens t_lo and t_lo+1 only seed the sliding sums (tap 0 of t_lo, taps 1 and 0 of
        // t_lo+1): three loads issued together with the first group, not a round of their own
        auto load_row = [&](int s, int tap) {
            const float *row = base + (size_t)(s < 0 ? 0 : s) * PSTR + tap * PF;   // clamped: no branch
            const f32x4 v = *reinterpret_cast<const f32x4 *>(row);
            return s < 0 ? zero : v;
        };
        const int s0 = sl[worker][0], s1 = sl[worker][1];
        const f32x4 h00 = load_row(s0, 0), h11 = load_row(s1, 1), h10 = load_row(s1, 0);
        const int npos = npos_eff;                           // positions p_lo .. <-> tokens 2 ..; 1 if the slice is uniform
        bool seeded = false;
        // a group of GDEPTH tokens: its three tap rows each, requested together (clamped: no branch)
        auto loadg = [&](int k, f32x4 (&r0)[GDEPTH], f32x4 (&r1)[GDEPTH], f32x4 (&r2)[GDEPTH]) {
#pragma unroll
            for (int u = 0; u < GDEPTH; ++u) {
                const int s = (k + u < npos) ? sl[worker][2 + k + u] : -1;
                r0[u] = load_row(s, 0);
                r1[u] = load_row(s, 1);
                r2[u] = load_row(s, 2);
            }
        };
        auto compg = [&](int k, const f32x4 (&r0)[GDEPTH], const f32x4 (&r1)[GDEPTH], const f32x4 (&r2)[GDEPTH]) {
            if (!seeded) { s_a = h00 + h11; s_b = h10; seeded = true; }
#pragma unroll
            for (int u = 0; u < GDEPTH; ++u) {
                if (k + u < npos) {
                    const int p = p_lo + k + u;
                    const f32x4 y = (s_a + r2[u]) + bias;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (y[c] > best[c]) { best[c] = y[c]; bp[c] = p; }
                    s_a = s_b + r1[u];
                    s_b = r0[u];
                }
            }
        };
        for (int k = 0; k < npos; k += GDEPTH) {
            f32x4 r0[GDEPTH], r1[GDEPTH], r2[GDEPTH];
            loadg(k, r0, r1, r2);
            compg(k, r0, r1, r2);
        }
    }
    if (act) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { sbest[worker][wl * 4 + c] = best[c]; sbp[worker][wl * 4 + c] = bp[c]; }
    }
    __syncthreads();
    HEAD_STAMP(2)
    // merge the 4 slices of each segment in position order; thread f (< 100) of each half
    const int half = threadIdx.x >> 7, f = threadIdx.x & 127;
    const unsigned ounit = unit_of(half);
    if (f < PF && ounit < units && half < GSPW) {
        float mb = sbest[half * GWPS][f];
        int mp = sbp[half * GWPS][f];
#pragma unroll
        for (int w = 1; w < GWPS; ++w) {
            const float v = sbest[half * GWPS + w][f];
            if (v > mb) { mb = v; mp = sbp[half * GWPS + w][f]; }
        }
        const size_t o = (size_t)ounit * NP + f;           // [doc][tile][NP]
        tw.pmax[o] = mb;
        tw.parg[o] = mp;
    }
    HEAD_STAMP(3)
}

// ----------------------------------------------------------------- launchers
int proj_tiles(int T) { return (T + 2 + SEG - 1) / SEG; }
int64_t proj_row_capacity(int64_t N, int T, int64_t V) { return (N * T < V) ? N * T : V; }
size_t proj_ptab_floats(int64_t N, int T, int64_t V) { return (size_t)proj_row_capacity(N, T, V) * PSTR; }

constexpr int GEMM_DEFAULT_FORM = 5;
constexpr int WR_DEFAULT_CHUNKS = 4;   // form 3 hands launches with at most this many K chunks to form 4
static int g_gemm_balanced = -1;       // -1: from the environment (R4R_GEMM=tile | whole pin the tile form) on first use
static int g_gemm_math = 0;            // 0: fp32 MFMA (the default and the headline); 1: fp16-split operands (project_f16.hip)
static float g_table_maxabs = 0.f;     // max |table|, given with mode 1 (the table is frozen: the host computes it once)
static float g_weight_maxabs = 0.f;    // max |conv weights|, likewise (re-read by the host every few steps)
int proj_gemm_f16_launch(const float *table, const ProjTower *tw, int ntower, int cap, int E, float table_maxabs,
                         float weight_maxabs, hipStream_t st);
size_t proj_gemm_f16_wimg_bytes(int E);

static ProjArgs make_args(const float *table, int64_t V, const ProjTower *tw, int ntower,
                          int64_t N, int T, int E, int F) {
    ProjArgs a;
    for (int k = 0; k < MAX_TOWERS; ++k) a.t[k] = tw[k < ntower ? k : 0];
    a.table = table; a.N = N; a.V = V; a.T = T; a.E = E; a.F = F;
    a.nchunk = (E + PEC - 1) / PEC;
    a.tiles = proj_tiles(T);
    a.cap = (int)proj_row_capacity(N, T, V);
    a.ntower = ntower;
    if (g_gemm_balanced < 0) {
        const char *e = getenv("R4R_GEMM");
        g_gemm_balanced = (e && e[0] == 't') ? 0 : ((e && e[0] == 'w') ? 2 : ((e && e[0] == 'a') ? 3 : ((e && e[0] == 'b') ? 1 : ((e && e[0] == 'r') ? 4 : GEMM_DEFAULT_FORM))));
    }
    a.balanced = g_gemm_balanced;
    // narrow tables (E <= 64: four K chunks) take the weight-resident form by default; R4R_GEMM=r asks for it up to E = 128
    if (a.balanced == 5) a.balanced = a.nchunk <= WR_DEFAULT_CHUNKS ? 4 : 3;    // (5 = the default: the form by table width)
    if (a.balanced == 4 && (a.nchunk > WR_MAX_CHUNKS || E % 4 != 0 || (int64_t)a.cap * PSTR * 4 >= (1ll << 31))) a.balanced = 3;
    // the A-resident form holds all of K in LDS (E <= 320) and addresses its output with 32-bit byte offsets
    if (a.balanced == 3 && (a.nchunk > AR_MAX_CHUNKS || (int64_t)a.cap * PSTR * 4 >= (1ll << 31))) a.balanced = 1;
    return a;
}

// Phase A: token state of a batch (flags -> slot / list / count).  Depends only on the indices,
// so a caller may run it for batch k+1 on another stream while step k computes.
int textcnn_proj_tokens_launch(int64_t V, const ProjTower *tw, int ntower, int64_t N, int T,
                               bool zero_state, hipStream_t st) {
    const ProjArgs a = make_args(nullptr, V, tw, ntower, N, T, PEC, PF);
    if (zero_state) {
        int zb = (int)cdiv(V, 256 * 4);
        if (zb > 1024) zb = 1024;
        proj_zero_kernel<<<dim3(zb < 1 ? 1 : zb, ntower), 256, 0, st>>>(a);
    }
    const TokenArgs ta = make_token_args(V, tw, ntower, N, T);
    int mark_blocks = (int)cdiv(N * T * ntower, 256 * 8);
    if (mark_blocks > 4096) mark_blocks = 4096;
    if (mark_blocks < 1) mark_blocks = 1;
    proj_mark_kernel<<<mark_blocks, 256, 0, st>>>(ta);
    proj_compact_kernel<<<dim3((unsigned)cdiv((V + 3) / 4, 1024), ntower), 1024, 0, st>>>(ta);
    return check_launch("textcnn_proj_tokens");
}

template <int NCH>
static void wres_launch_n(const ProjArgs &a, unsigned wgs, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(proj_gemm_wres_kernel<NCH>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, wr_lds_bytes(NCH));
        attr_set = true;
    }
    launch_timed(R4R_TIMING_PROJ_GEMM, proj_gemm_wres_kernel<NCH>, dim3(wgs), dim3(WR_THREADS), wr_lds_bytes(NCH), st, a);
}
static void wres_launch(const ProjArgs &a, unsigned wgs, hipStream_t st) {
    switch (a.nchunk) {
        case 1: wres_launch_n<1>(a, wgs, st); break;
        case 2: wres_launch_n<2>(a, wgs, st); break;
        case 3: wres_launch_n<3>(a, wgs, st); break;
        case 4: wres_launch_n<4>(a, wgs, st); break;
        case 5: wres_launch_n<5>(a, wgs, st); break;
        case 6: wres_launch_n<6>(a, wgs, st); break;
        case 7: wres_launch_n<7>(a, wgs, st); break;
        default: wres_launch_n<8>(a, wgs, st); break;
    }
}

// Phase B: projection GEMM + gather-add-max, given the token state.
int textcnn_proj_compute_launch(const float *table, int64_t V, const ProjTower *tw, int ntower,
                                int64_t N, int T, int E, int F, hipStream_t st) {
    if (F != PF) {
        set_error("project-then-gather path is built for %d filters, got %d", PF, F);
        return R4R_ERR_ARG;
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(proj_gemm_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, AR_MAX_CHUNKS * AR_CHUNK * 4);
        attr_set = true;
    }
    const ProjArgs a = make_args(table, V, tw, ntower, N, T, E, F);
    bool f16 = g_gemm_math >= 1 && proj_gemm_f16_wimg_bytes(E) <= textcnn_wp_floats(E) * 4;
    for (int t = 0; t < ntower; ++t) f16 = f16 && tw[t].wimg != nullptr;       // (callers without the scratch: fp32)
    if (f16) {
        ScopedTiming tm(R4R_TIMING_PROJ_GEMM, st);           // (the opt-in arithmetic: its weight-scale launch included)
        if (int rc = proj_gemm_f16_launch(table, tw, ntower, a.cap, E, g_table_maxabs, g_weight_maxabs, st)) return rc;
    } else if (a.balanced == 4) {
        // one workgroup per CU at most; fewer when the row capacity is small (a workgroup's 8 waves take 8 units per round)
        int64_t wgs = ((int64_t)a.cap + 15) / 16 * WR_NQ / WR_WAVES + ntower;
        // two workgroups per CU where the weight image leaves room AND the launch is long enough to pay for a second
        // 78 KB fill per CU (by row capacity: a million-word vocabulary -- cfg5 46.3 -> 43.9 us, its full / uniform point
        // 174 -> 158; at a 50 k-word vocabulary, three units per wave, the second fill costs more than the occupancy buys:
        // cfg4 16.4 -> 18.6 us; profiles/r06_ab_wres2.txt)
        const int64_t resident = (int64_t)G7_WGS * (wgs >= 8 * G7_WGS ? wr_wgs_per_cu(a.nchunk) : 1);
        if (wgs > resident) wgs = resident;
        if (wgs < ntower) wgs = ntower;
        wres_launch(a, (unsigned)wgs, st);
    } else {
        // persistent: one workgroup per CU at most (86 KB of LDS each), fewer when the row capacity is small
        const int rows_wg = (a.balanced == 3 ? AR_PMIN : G7_ROWS) * 16;     // fewest rows a workgroup may own
        int64_t wgs = ((int64_t)a.cap + rows_wg - 1) / rows_wg * ntower;
        if (wgs > G7_WGS) wgs = G7_WGS;
        const int ares_bytes = a.nchunk * AR_CHUNK * 4;     // form 3 keeps every K chunk of its 128 rows resident
        const int lds_bytes = (a.balanced == 3 && ares_bytes > GEMM_LDS_BYTES) ? ares_bytes : GEMM_LDS_BYTES;
        launch_timed(R4R_TIMING_PROJ_GEMM, proj_gemm_kernel, dim3((unsigned)wgs), dim3(GEMM_THREADS), lds_bytes, st, a);
    }
    {
        ScopedTiming tm(R4R_TIMING_PROJ_GATHER, st, /*chain=*/true);     // (starts where the GEMM's span ended)
        proj_gather_max_kernel<<<dim3((unsigned)cdiv(N * a.tiles, GSPW), ntower), 256, 0, st>>>(a);
    }
    return check_launch("textcnn_proj_fwd");
}

void proj_gemm_set_form(int balanced) { g_gemm_balanced = balanced; }
int proj_gemm_math_mode() { return g_gemm_math; }
void proj_gemm_set_math(int mode, float table_maxabs, float weight_maxabs) {
    g_gemm_math = mode; g_table_maxabs = table_maxabs; g_weight_maxabs = weight_maxabs;
}

int textcnn_proj_fwd_launch(const float *table, int64_t V, const ProjTower *tw, int ntower,
                            int64_t N, int T, int E, int F, bool zero_state, hipStream_t st) {
    if (int rc = textcnn_proj_tokens_launch(V, tw, ntower, N, T, zero_state, st)) return rc;
    return textcnn_proj_compute_launch(table, V, tw, ntower, N, T, E, F, st);
}

}  // namespace r4r
