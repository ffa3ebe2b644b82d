// Opt-in arithmetic for the projection GEMM of project-then-gather (project.hip): fp16-SPLIT operands on
// the f16 matrix cores with fp32 accumulation -- SURVEY 7 step 8 ("bf16-split or fp16 MFMA with fp32
// accumulate only if parity holds").  NEVER the default and never the fp32 headline: `r4r_gemm_math(1, ..)`
// / R4R_GEMM_MATH=f16x2 switch it on, bench.py reports it under its own dtype.
//
// Why: the fp32 MFMA (v_mfma_f32_16x16x4_f32, 64 flop/clk/SIMD) bounds the headline step; the f16 MFMA
// (v_mfma_f32_16x16x32_f16, 1024 flop/clk/SIMD) is 16x faster.  An fp32 number splits EXACTLY into
// x = hi + lo with hi = fp16(x), lo = fp16(x - hi) (11 + 11 mantissa bits), and
//     a b = a_hi b_hi + a_hi b_lo + a_lo b_hi            (+ a_lo b_lo, 2^-22 relative: dropped)
// is three f16 MFMAs into ONE fp32 accumulator: 16 / 3 = 5.3x the fp32-MFMA rate with a relative error
// per product of ~2^-21 (fp32's own is 2^-24; the K = 300 accumulation is fp32 either way).  fp16's narrow
// exponent is handled by two exact power-of-two scales: the frozen table's maximum (given once by the host)
// and the weights' maximum (a one-workgroup kernel per launch) put both operands' largest magnitude at
// 2^13, so `lo` stays a normal fp16 number for every element within 2^-17 of the maximum (smaller ones
// lose relative, not absolute, accuracy: < 2^-28 of the largest product); the product of the scales is
// divided out in the epilogue.  Measured accuracy and speed: DESIGN.md 4.1d.
//
// The B operand (the conv weights: the same 304 x K matrix for every workgroup, 71 % of the rows a tile
// stages) is split and packed ONCE per launch into its final LDS image -- [chunk][320 rows][144 B], in the
// tower's weight-image scratch (the region the direct conv packs its weights into, idle on this path) -- by
// the one-workgroup-per-tower kernel that also finds the weights' scale; the GEMM then streams that image
// with fully coalesced 16-byte loads (a wave-wide load = one contiguous kilobyte) and converts only its A
// rows.  With the matrix pipe 5x faster the kernel is bound by the L1 / TA line-request rate of its
// staging, so gathered 128-byte row pieces for B (first version: 50 us) were the wrong shape.
//
// Structure = the tile form of project.hip (persistent grid, 128-row x 304-column tiles, LDS double-buffered,
// staging registers a chunk ahead) with squarer wave tiles: 8 waves of 64 rows x (5 | 5 | 5 | 4) column tiles
// read 18 instead of 24 operand pieces from LDS per 60 MFMAs -- with the matrix pipe this fast, LDS reads
// are the next limiter; a K chunk is
// 32 wide (one MFMA k-step) and holds both planes: a row is [32 hi | 32 lo] fp16 = 128 B + 16 B pad
// (row stride 144 B = 16 x odd: the 16 rows a ds_read_b128 touches fall into distinct bank groups).
#include "textcnn.h"

namespace r4r {

typedef float h_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int HK = 32;                 // K per chunk
constexpr int HROW = 144;              // LDS bytes per operand row
constexpr int HM = 128;                // rows per tile
constexpr int HB_ROWS = 320;           // B rows staged (304 padded to 5 x 64)
constexpr int HBUF = (HM + HB_ROWS) * HROW;        // bytes per LDS buffer (64,512)
constexpr int H_LDS_BYTES = 2 * HBUF;               // 129,024
constexpr int H_THREADS = 512;
constexpr int HPF = 100, HPROW = 300, HNH_COLS = 160;

struct F16Tower {
    const float *conv_w;
    const int *list;
    const int *count;                  // [0] live count
    float *ptab;
    char *wimg;                        // packed B: [nchunk][HB_ROWS][HROW bytes]
};
struct F16Args {
    F16Tower t[MAX_TOWERS];
    const float *table;
    int E, ntower, nchunk;
    float a_scale, b_scale, out_scale; // exact powers of two: table, weights, 1 / their product
};

// The scaled weights, split into hi / lo fp16 planes, in the GEMM's LDS row format (row n = tap j * 100 +
// filter f; rows >= 300 and k >= E are zero).  grid = (blocks, towers); one thread per (chunk, row, octet).
// The scale comes from the host (r4r_gemm_math: max |w| re-read every few steps, with headroom), so nothing
// here waits for a reduction over the weights.
__global__ __launch_bounds__(256) void proj_wpack_kernel(F16Args a) {
    const F16Tower &tw = a.t[blockIdx.y];
    const int E = a.E;
    const int items = a.nchunk * HB_ROWS * 4;
    const int it = blockIdx.x * 256 + threadIdx.x;
    if (it >= items) return;
    const int o = it & 3, row = (it >> 2) % HB_ROWS, c = (it >> 2) / HB_ROWS;
    const int j = row / HPF, f = row - j * HPF;
    const float *src = tw.conv_w + ((long)f * 3 + j) * E;
    const int k0 = c * HK + o * 8;
    h_f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0;
    if (row < HPROW && k0 < E) q0 = *reinterpret_cast<const h_f32x4 *>(src + k0);          // (E % 4 == 0)
    if (row < HPROW && k0 + 4 < E) q1 = *reinterpret_cast<const h_f32x4 *>(src + k0 + 4);
    f16x8 hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x = q0[i] * a.b_scale, y = q1[i] * a.b_scale;
        const _Float16 hx = (_Float16)x, hy = (_Float16)y;
        hi[i] = hx; hi[4 + i] = hy;
        lo[i] = (_Float16)(x - (float)hx); lo[4 + i] = (_Float16)(y - (float)hy);
    }
    char *dst = tw.wimg + ((size_t)c * HB_ROWS + row) * HROW + o * 16;
    *reinterpret_cast<f16x8 *>(dst) = hi;
    *reinterpret_cast<f16x8 *>(dst + 64) = lo;
}

__device__ __forceinline__ void split8(const h_f32x4 &q0, const h_f32x4 &q1, float scale, f16x8 &hi, f16x8 &lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x = q0[i] * scale, y = q1[i] * scale;
        const _Float16 hx = (_Float16)x, hy = (_Float16)y;
        hi[i] = hx; hi[4 + i] = hy;
        lo[i] = (_Float16)(x - (float)hx); lo[4 + i] = (_Float16)(y - (float)hy);
    }
}

template <int NTILE>
__device__ __forceinline__ void proj_gemm_f16_body(const F16Args &a, char *lds, int tower, int row0) {
    const F16Tower &tw = a.t[tower];
    const int count = tw.count[0];
    const float out_scale = a.out_scale;
    const float *__restrict__ table = a.table;
    const int E = a.E, nchunk = a.nchunk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kq = lane >> 4;
    const int rg = wave & 1, cg = wave >> 1, col0 = cg * 80;   // wave tile: rows 64 rg .. +63, column tiles 5 cg .. (+NTILE)
    // staging roles.  A: octet o (k = 8 o .. 8 o + 7) of row (tid >> 2), gathered from the table and split
    // here.  B: the packed image of a chunk is HB_ROWS * HROW = 46,080 contiguous bytes = 2,880 pieces of 16:
    // thread tid copies pieces tid + 512 i (i < 6; the last round is partial)
    const int o = tid & 3, srow = tid >> 2;
    const float *aptr = table + (long)tw.list[min(row0 + srow, count - 1)] * E;
    constexpr int BPIECES = HB_ROWS * HROW / 16, BROUNDS = (BPIECES + H_THREADS - 1) / H_THREADS;
    h_f32x4 ar[2], br[BROUNDS];
    auto issue_loads = [&](int c) {                          // unconditional: past the row's end the last float4 is re-read
        const int e0 = min(c * HK + o * 8, E - 4), e1 = min(c * HK + o * 8 + 4, E - 4);
        ar[0] = *reinterpret_cast<const h_f32x4 *>(aptr + e0);
        ar[1] = *reinterpret_cast<const h_f32x4 *>(aptr + e1);
        const char *img = tw.wimg + (size_t)min(c, nchunk - 1) * (HB_ROWS * HROW);
#pragma unroll
        for (int i = 0; i < BROUNDS; ++i)
            br[i] = *reinterpret_cast<const h_f32x4 *>(img + (size_t)min(tid + H_THREADS * i, BPIECES - 1) * 16);
    };
    auto write_lds = [&](char *buf, int c) {
        // (the K tail: B's image is zero where k >= E, so whatever A re-read there multiplies zero)
        f16x8 hi, lo;
        split8(ar[0], ar[1], a.a_scale, hi, lo);
        *reinterpret_cast<f16x8 *>(buf + srow * HROW + o * 16) = hi;
        *reinterpret_cast<f16x8 *>(buf + srow * HROW + 64 + o * 16) = lo;
        char *Bl = buf + HM * HROW;
#pragma unroll
        for (int i = 0; i < BROUNDS; ++i)
            if (tid + H_THREADS * i < BPIECES) *reinterpret_cast<h_f32x4 *>(Bl + (size_t)(tid + H_THREADS * i) * 16) = br[i];
    };
    constexpr int MT = 4;                                   // row tiles per wave
    h_f32x4 acc[MT][NTILE];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NTILE; ++ni) acc[mi][ni] = (h_f32x4){0.f, 0.f, 0.f, 0.f};
    const int aoff = (rg * 64 + lrow) * HROW + kq * 16;
    const int boff = (HM + col0 + lrow) * HROW + kq * 16;
    auto compute = [&](const char *buf) {
        f16x8 ah[MT], al[MT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            ah[mi] = *reinterpret_cast<const f16x8 *>(buf + aoff + mi * 16 * HROW);
            al[mi] = *reinterpret_cast<const f16x8 *>(buf + aoff + mi * 16 * HROW + 64);
        }
        f16x8 bh[NTILE], bl[NTILE];
#pragma unroll
        for (int ni = 0; ni < NTILE; ++ni) {
            bh[ni] = *reinterpret_cast<const f16x8 *>(buf + boff + ni * 16 * HROW);
            bl[ni] = *reinterpret_cast<const f16x8 *>(buf + boff + ni * 16 * HROW + 64);
        }
        // term by term over all MT x NTILE accumulators: the three MFMAs of one accumulator are dependent, 20
        // independent ones between them keep the matrix pipe issuing back to back (small terms first)
#pragma unroll
        for (int ni = 0; ni < NTILE; ++ni)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < NTILE; ++ni)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < NTILE; ++ni)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
    };
    // chunk c: LDS buffer c & 1 holds it; the staging registers hold chunk c + 1 (loaded a chunk ago)
    issue_loads(0);
    write_lds(lds, 0);
    issue_loads(1);
    for (int c = 0; c < nchunk; ++c) {
        __syncthreads();                                    // chunk c visible; buffer (c + 1) & 1 fully read
        write_lds(lds + ((c + 1) & 1) * HBUF, c + 1);
        issue_loads(c + 2);
        compute(lds + (c & 1) * HBUF);
    }
    __syncthreads();                                        // all operand reads done: LDS is free
    // epilogue: as the fp32 form -- per wave one 16-row tile at a time through its own LDS slab, float4 stores
    constexpr int TS = NTILE * 16 + 4;
    float *slab = reinterpret_cast<float *>(lds) + wave * (16 * (HNH_COLS + 4));
    constexpr int NV = NTILE * 4;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
        for (int ni = 0; ni < NTILE; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[(kq * 4 + r) * TS + ni * 16 + lrow] = acc[mi][ni][r] * out_scale;
        for (int i = lane; i < 16 * NV; i += 64) {
            const int rr = i / NV, cv = i - rr * NV;
            const int row = row0 + rg * 64 + mi * 16 + rr;
            const int col = col0 + cv * 4;
            if (row < count && col < HPROW)
                *reinterpret_cast<h_f32x4 *>(tw.ptab + (size_t)row * HPROW + col) =
                    *reinterpret_cast<const h_f32x4 *>(slab + rr * TS + cv * 4);
        }
    }
}

__global__ __launch_bounds__(H_THREADS) void proj_gemm_f16_kernel(F16Args a) {
    extern __shared__ __attribute__((aligned(16))) char hsmem[];
    int nt[MAX_TOWERS], total = 0;
#pragma unroll
    for (int t = 0; t < MAX_TOWERS; ++t) {
        nt[t] = t < a.ntower ? __builtin_amdgcn_readfirstlane((a.t[t].count[0] + HM - 1) / HM) : 0;
        total += nt[t];
    }
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        int t = 0, local = tile;
#pragma unroll
        for (int k = 0; k < MAX_TOWERS - 1; ++k)
            if (t == k && local >= nt[k]) { local -= nt[k]; t = k + 1; }
        if ((threadIdx.x >> 7) == 3) proj_gemm_f16_body<4>(a, hsmem, t, local * HM);   // waves 6, 7: column tiles 15..18
        else proj_gemm_f16_body<5>(a, hsmem, t, local * HM);
        __syncthreads();
    }
}

// bytes of weight-image scratch the fp16-split form needs per tower
size_t proj_gemm_f16_wimg_bytes(int E) { return (size_t)((E + HK - 1) / HK) * HB_ROWS * HROW; }

int proj_gemm_f16_launch(const float *table, const ProjTower *tw, int ntower, int cap, int E, float table_maxabs,
                         float weight_maxabs, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(proj_gemm_f16_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS_BYTES);
        attr_set = true;
    }
    F16Args a;
    for (int k = 0; k < MAX_TOWERS; ++k) {
        const ProjTower &s = tw[k < ntower ? k : 0];
        a.t[k].conv_w = s.conv_w; a.t[k].list = s.list; a.t[k].count = s.count; a.t[k].ptab = s.ptab;
        a.t[k].wimg = reinterpret_cast<char *>(s.wimg);
    }
    a.table = table; a.E = E; a.ntower = ntower; a.nchunk = (E + HK - 1) / HK;
    // the largest |table| lands in [2^12, 2^13); the weights' (a value the host re-reads every few steps) two
    // binades lower, so that they can quadruple before the host looks again without overflowing fp16
    int ea = 0, eb = 0;
    if (table_maxabs > 0.f && table_maxabs < INFINITY) (void)frexpf(table_maxabs, &ea);
    if (weight_maxabs > 0.f && weight_maxabs < INFINITY) (void)frexpf(weight_maxabs, &eb);
    a.a_scale = ldexpf(1.f, 13 - ea);
    a.b_scale = ldexpf(1.f, 11 - eb);
    a.out_scale = ldexpf(1.f, (ea - 13) + (eb - 11));
    const int items = a.nchunk * HB_ROWS * 4;
    proj_wpack_kernel<<<dim3((unsigned)cdiv(items, 256), ntower), 256, 0, st>>>(a);
    int64_t wgs = ((int64_t)cap + HM - 1) / HM * ntower;
    if (wgs > 256) wgs = 256;
    if (wgs < 1) wgs = 1;
    proj_gemm_f16_kernel<<<dim3((unsigned)wgs), H_THREADS, H_LDS_BYTES, st>>>(a);
    return check_launch("proj_gemm_f16");
}

}  // namespace r4r
