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
// The B operand (the conv weights: the same 304 x K matrix for every workgroup) is split and packed ONCE per
// launch (proj_wpack_kernel) into the tower's weight-image scratch -- the region the direct conv packs its
// weights into, idle on this path -- in the MFMA's B-FRAGMENT order, so a wave reads its column tiles'
// fragments straight into the operand registers with one contiguous kilobyte per load and B never touches
// LDS; only the gathered, freshly split A rows go through LDS (two 18 KB buffers, operand registers
// double-buffered).  Structure otherwise = the tile form of project.hip (persistent grid, 128-row x
// 304-column tiles) with squarer wave tiles: 8 waves of 64 rows x (5 | 5 | 5 | 4) column tiles; a K chunk
// is 32 wide (one MFMA k-step); an A row in LDS is [32 hi | 32 lo] fp16 = 128 B + 16 B pad (row stride 144
// B = 16 x odd: the 16 rows a ds_read_b128 touches fall into distinct bank groups).
// History and the timing-only ablations behind this shape: DESIGN.md 4.1d.
#include "textcnn.h"
#include "trace_device.h"

namespace r4r {

typedef float h_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int HK = 32;                 // K per chunk (one MFMA k-step)
constexpr int HROW = 144;              // LDS bytes per A row: [32 hi | 32 lo] fp16 + 16 B pad
constexpr int HM = 128;                // rows per tile
constexpr int HCT = 20;                // column tiles of 16 in the packed B image (19 real + 1 of zeros)
constexpr int HABUF = HM * HROW;       // bytes per A buffer (18,432)
constexpr int H_THREADS = 512;
constexpr int HPF = 100, HPROW = 300, HNH_COLS = 160;
#ifndef R4R_PSTR
#define R4R_PSTR 304
#endif
constexpr int HPSTR = R4R_PSTR;   // row stride of the projected-row table (project.hip: PSTR)
constexpr int H_LDS_BYTES = 8 * 16 * (HNH_COLS + 4) * 4;   // the epilogue's slabs (83,968) > the loop's two A buffers
static_assert(2 * HABUF <= H_LDS_BYTES, "the A buffers live in the epilogue's region");
constexpr int HIMG_CHUNK = HCT * 2 * 64 * 16;              // bytes of packed B per chunk (40,960)

struct F16Tower {
    const float *conv_w;
    const int *list;
    const int *count;                  // [0] live count
    float *ptab;
    char *wimg;                        // packed B: [nchunk][HCT column tiles][hi, lo][64 lanes][16 B]
};
struct F16Args {
    F16Tower t[MAX_TOWERS];
    const float *table;
    int E, ntower, nchunk;
    float a_scale, b_scale, out_scale; // exact powers of two: table, weights, 1 / their product
};

__device__ __forceinline__ void split8(const h_f32x4 &q0, const h_f32x4 &q1, float scale, f16x8 &hi, f16x8 &lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x = q0[i] * scale, y = q1[i] * scale;
        const _Float16 hx = (_Float16)x, hy = (_Float16)y;
        hi[i] = hx; hi[4 + i] = hy;
        lo[i] = (_Float16)(x - (float)hx); lo[4 + i] = (_Float16)(y - (float)hy);
    }
}

// The scaled weights, split into hi / lo fp16 planes, in the MFMA's B-FRAGMENT order: the 16 bytes lane l of
// a wave feeds into v_mfma_f32_16x16x32_f16 for column tile ct of chunk c -- column n = 16 ct + (l & 15) (row
// n = tap j * 100 + filter f of the [300][K] matrix; n >= 300 and k >= E are zero), k = 32 c + 8 (l >> 4) ..
// + 7 -- are contiguous per lane and a wave's fragment is one contiguous kilobyte.  grid = (blocks, towers);
// one thread per (chunk, column tile, lane).  The scale comes from the host (r4r_gemm_math: max |w| re-read
// every few steps, with headroom), so nothing here waits for a reduction over the weights.
__global__ __launch_bounds__(256) void proj_wpack_kernel(F16Args a) {
    const F16Tower &tw = a.t[blockIdx.y];
    const int E = a.E;
    const int items = a.nchunk * HCT * 64;
    const int it = blockIdx.x * 256 + threadIdx.x;
    if (it >= items) return;
    const int lane = it & 63, ct = (it >> 6) % HCT, c = (it >> 6) / HCT;
    const int n = ct * 16 + (lane & 15);
    const int j = n / HPF, f = n - j * HPF;
    const float *src = tw.conv_w + ((long)f * 3 + j) * E;
    const int k0 = c * HK + (lane >> 4) * 8;
    h_f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0;
    if (n < HPROW && k0 < E) q0 = *reinterpret_cast<const h_f32x4 *>(src + k0);          // (E % 4 == 0)
    if (n < HPROW && k0 + 4 < E) q1 = *reinterpret_cast<const h_f32x4 *>(src + k0 + 4);
    f16x8 hi, lo;
    split8(q0, q1, a.b_scale, hi, lo);
    char *dst = tw.wimg + (size_t)c * HIMG_CHUNK + ((size_t)ct * 2 * 64 + lane) * 16;
    *reinterpret_cast<f16x8 *>(dst) = hi;
    *reinterpret_cast<f16x8 *>(dst + 64 * 16) = lo;
}

// One 128-row x 304-column tile.  Wave w: rows 64 (w & 1) .. +63 (4 row tiles), column tiles 5 (w >> 1) ..
// (+NTILE).  Per chunk of K = 32:
//   B  never touches LDS: a wave reads ITS column tiles' fragments (hi, lo) straight from the packed image
//      into the MFMA operand registers -- one contiguous kilobyte per load, L2-resident, the same for every
//      workgroup -- and reloads a column's pair right after that column's last MFMA, for the next chunk;
//   A  (the gathered table rows, split here) goes through two 18 KB LDS buffers; operand registers are
//      double-buffered, so chunk c + 1's ds_reads and chunk c + 2's staging sit among chunk c's MFMAs.
// (First version: both operands through LDS, 204 KB of LDS traffic per chunk and CU -- as long as the
// chunk's MFMAs, and serialised with them by the chunk barrier: 28 us.  DESIGN.md 4.1d.)
HEAD_TRACE_DEFINE(r4r_debug_f16_gemm_trace)
template <int NTILE>
__device__ __forceinline__ void proj_gemm_f16_body(const F16Args &a, char *lds, int tower, int row0) {
    HEAD_STAMP(0)
    const F16Tower &tw = a.t[tower];
    const int count = tw.count[0];
    const float out_scale = a.out_scale;
    const float *__restrict__ table = a.table;
    const int E = a.E, nchunk = a.nchunk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kq = lane >> 4;
    const int rg = wave & 1, cg = wave >> 1, col0 = cg * 80;
    // A staging role: octet o (k = 8 o .. 8 o + 7) of row (tid >> 2), gathered from the table and split here
    const int o = tid & 3, srow = tid >> 2;
    const float *aptr = table + (long)tw.list[min(row0 + srow, count - 1)] * E;
    h_f32x4 s0[2], s1[2];                                   // two staging sets: a chunk's loads get two intervals to arrive
    auto issue_a = [&](int c, h_f32x4 *st) {                // unconditional: past the row's end the last float4 is re-read
        const int e0 = min(c * HK + o * 8, E - 4), e1 = min(c * HK + o * 8 + 4, E - 4);
        st[0] = *reinterpret_cast<const h_f32x4 *>(aptr + e0);
        st[1] = *reinterpret_cast<const h_f32x4 *>(aptr + e1);
    };
    auto write_a = [&](char *buf, const h_f32x4 *st) {
        // (the K tail: B's image is zero where k >= E, so whatever A re-read there multiplies zero)
        f16x8 hi, lo;
        split8(st[0], st[1], a.a_scale, hi, lo);
        *reinterpret_cast<f16x8 *>(buf + srow * HROW + o * 16) = hi;
        *reinterpret_cast<f16x8 *>(buf + srow * HROW + 64 + o * 16) = lo;
    };
    constexpr int MT = 4;                                   // row tiles per wave
    h_f32x4 acc[MT][NTILE];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NTILE; ++ni) acc[mi][ni] = (h_f32x4){0.f, 0.f, 0.f, 0.f};
    const int aoff = (rg * 64 + lrow) * HROW + kq * 16;
    const char *bimg = tw.wimg + ((size_t)(cg * 5) * 2 * 64 + lane) * 16;      // + c HIMG_CHUNK + ni 2048 (+ 1024: lo)
    f16x8 bh[NTILE], bl[NTILE];
    auto load_b = [&](int c, int ni) {
        const char *q = bimg + (size_t)min(c, nchunk - 1) * HIMG_CHUNK + ni * 2048;
        bh[ni] = *reinterpret_cast<const f16x8 *>(q);
        bl[ni] = *reinterpret_cast<const f16x8 *>(q + 1024);
    };
    auto read_a = [&](const char *buf, f16x8 *ah, f16x8 *al) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            ah[mi] = *reinterpret_cast<const f16x8 *>(buf + aoff + mi * 16 * HROW);
            al[mi] = *reinterpret_cast<const f16x8 *>(buf + aoff + mi * 16 * HROW + 64);
        }
    };
    // column by column: the three dependent MFMAs of an accumulator are four independent ones apart (small
    // terms first); a column's B pair is dead after its 12 MFMAs and is reloaded at once for chunk c + 1
    auto mfma_chunk = [&](const f16x8 *ah, const f16x8 *al, int c) {
#pragma unroll
        for (int ni = 0; ni < NTILE; ++ni) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
            load_b(c + 1, ni);
        }
    };
    char *buf0 = lds, *buf1 = lds + HABUF;
    f16x8 ah0[MT], al0[MT], ah1[MT], al1[MT];
    // prologue: chunk 0 in buffer 0 and (after the barrier) in operand set 0; chunk 1 in buffer 1; chunks 2, 3 in flight
    issue_a(0, s0);
    issue_a(1, s1);
#pragma unroll
    for (int ni = 0; ni < NTILE; ++ni) load_b(0, ni);
    write_a(buf0, s0);
    issue_a(2, s0);
    write_a(buf1, s1);
    issue_a(3, s1);
    __syncthreads();
    HEAD_STAMP(1)
    read_a(buf0, ah0, al0);
    // interval c (even: operand set 0 holds chunk c, buffer 1 chunk c + 1, s0 chunk c + 2, s1 chunk c + 3)
    for (int c = 0; c < nchunk; c += 2) {
        __syncthreads();                                    // buffer 1 (chunk c + 1) complete; every read of buffer 0 done
        read_a(buf1, ah1, al1);
        mfma_chunk(ah0, al0, c);
        write_a(buf0, s0);                                  // chunk c + 2
        issue_a(c + 4, s0);
        if (c + 1 < nchunk) {                               // uniform
            __syncthreads();                                // buffer 0 (chunk c + 2) complete; every read of buffer 1 done
            read_a(buf0, ah0, al0);
            mfma_chunk(ah1, al1, c + 1);
            write_a(buf1, s1);                              // chunk c + 3
            issue_a(c + 5, s1);
        }
    }
    __syncthreads();                                        // all operand reads done: LDS is free
    HEAD_STAMP(2)
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
                *reinterpret_cast<h_f32x4 *>(tw.ptab + (size_t)row * HPSTR + col) =
                    *reinterpret_cast<const h_f32x4 *>(slab + rr * TS + cv * 4);
        }
    }
    HEAD_STAMP(3)
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
size_t proj_gemm_f16_wimg_bytes(int E) { return (size_t)((E + HK - 1) / HK) * HIMG_CHUNK; }

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
    const int items = a.nchunk * HCT * 64;
    proj_wpack_kernel<<<dim3((unsigned)cdiv(items, 256), ntower), 256, 0, st>>>(a);
    int64_t wgs = ((int64_t)cap + HM - 1) / HM * ntower;
    if (wgs > 256) wgs = 256;
    if (wgs < 1) wgs = 1;
    proj_gemm_f16_kernel<<<dim3((unsigned)wgs), H_THREADS, H_LDS_BYTES, st>>>(a);
    return check_launch("proj_gemm_f16");
}

}  // namespace r4r
