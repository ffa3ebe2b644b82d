// Data parallel: the glue between a rank's compact gradient entries and the gathered arrays the
// `r4r_*_rows_apply` launches read -- ONE packing launch in front of the step's ONE all_gather, one unpacking launch
// behind it, whatever the family's fields (ids, d loss / d pred, a gradient row per ID table).  The reference is
// single-process (main.py:407); this replaces what the engines did with a dozen ATen slicing / fill / cat kernels
// and a second collective per step.
//   block (4-byte units): field f at off[f] = sum of the earlier fields' B_pad * width, each rounded up to 64 units;
//                         entries past the rank's own n carry the field's padding (ids: -1 = all ones; values: 0)
//   gathered arrays:      field f of all ranks, rank-major: dst[f][(r * B_pad + e) * width + u]
#include "common.h"

namespace r4r {

constexpr int DP_MAX_FIELDS = 8;

struct DpPack {
    const unsigned *src[DP_MAX_FIELDS];    // pack: this rank's field arrays [n, width]; unpack: unused
    unsigned *dst[DP_MAX_FIELDS];          // unpack: the gathered field arrays [world * B_pad, width]
    int width[DP_MAX_FIELDS];              // 4-byte units per entry (int64 ids: 2)
    int ones[DP_MAX_FIELDS];               // padding entries: all ones (id -1) instead of zero
    int64_t off[DP_MAX_FIELDS];            // field offset inside a block, units
    unsigned *block;                       // pack: the block to fill
    const unsigned *blocks;                // unpack: [world] blocks
    int64_t n, B_pad, blk_units;
    int world;
};

// (the field is the workgroup's blockIdx.y: a uniform index into the argument arrays is a scalar load, not the
// per-lane pointer fetch a lane-dependent index would be)
__global__ __launch_bounds__(256) void dp_pack_kernel(DpPack a) {
    const int f = blockIdx.y;
    const int w = a.width[f];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.B_pad * w) return;
    const int64_t e = i / w;
    a.block[a.off[f] + i] = e < a.n ? a.src[f][i] : (a.ones[f] ? 0xffffffffu : 0u);
}

__global__ __launch_bounds__(256) void dp_unpack_kernel(DpPack a) {
    const int f = blockIdx.y;
    const int64_t per = a.B_pad * a.width[f];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= per * a.world) return;
    const int64_t r = i / per;
    a.dst[f][i] = a.blocks[r * a.blk_units + a.off[f] + (i - r * per)];
}

static int64_t dp_layout(int nfields, const int *widths, int64_t B_pad, int64_t *off) {
    int64_t o = 0;
    for (int f = 0; f < nfields; ++f) {
        if (off) off[f] = o;
        o += (B_pad * widths[f] + 63) & ~(int64_t)63;
    }
    return o;
}

}  // namespace r4r

using namespace r4r;

extern "C" size_t r4r_dp_block_bytes(int nfields, const int *widths, int64_t B_pad) {
    if (nfields < 1 || nfields > DP_MAX_FIELDS || !widths || B_pad < 0) return 0;
    for (int f = 0; f < nfields; ++f)
        if (widths[f] < 1) return 0;
    return (size_t)dp_layout(nfields, widths, B_pad, nullptr) * 4;
}

static int dp_args(DpPack &a, const char *who, int nfields, const int *widths, int64_t B_pad) {
    R4R_REQUIRE(nfields >= 1 && nfields <= DP_MAX_FIELDS && widths, "%s: 1..%d fields", who, DP_MAX_FIELDS);
    int64_t maxw = 0;
    for (int f = 0; f < DP_MAX_FIELDS; ++f) {
        a.width[f] = f < nfields ? widths[f] : 1;
        R4R_REQUIRE(a.width[f] >= 1, "%s: field %d: width %d", who, f, a.width[f]);
        a.src[f] = nullptr; a.dst[f] = nullptr; a.ones[f] = 0; a.off[f] = 0;
        if (f < nfields && a.width[f] > maxw) maxw = a.width[f];
    }
    a.B_pad = B_pad;
    a.blk_units = dp_layout(nfields, widths, B_pad, a.off);
    return (int)maxw;
}

extern "C" int r4r_dp_pack(int nfields, const uint64_t *src, const int *widths, const int *pad_ones, int64_t n,
                           int64_t B_pad, void *block, void *stream) {
    R4R_REQUIRE(src && pad_ones && block && n >= 0 && B_pad >= n, "dp_pack: null pointer, or B_pad < n");
    DpPack a{};
    const int maxw = dp_args(a, "dp_pack", nfields, widths, B_pad);
    if (maxw <= 0) return R4R_ERR_ARG;
    if (B_pad == 0) return R4R_OK;
    for (int f = 0; f < nfields; ++f) {
        a.src[f] = reinterpret_cast<const unsigned *>(src[f]);
        a.ones[f] = pad_ones[f];
        R4R_REQUIRE(n == 0 || a.src[f], "dp_pack: field %d: null source", f);
    }
    a.block = static_cast<unsigned *>(block); a.n = n; a.world = 1;
    dp_pack_kernel<<<dim3((unsigned)cdiv(B_pad * maxw, 256), (unsigned)nfields), 256, 0, as_stream(stream)>>>(a);
    return check_launch("dp_pack");
}

extern "C" int r4r_dp_unpack(int nfields, const uint64_t *dst, const int *widths, const void *blocks, int world,
                             int64_t B_pad, void *stream) {
    R4R_REQUIRE(dst && blocks && world >= 1 && B_pad >= 0, "dp_unpack: null pointer or bad sizes");
    DpPack a{};
    const int maxw = dp_args(a, "dp_unpack", nfields, widths, B_pad);
    if (maxw <= 0) return R4R_ERR_ARG;
    if (B_pad == 0) return R4R_OK;
    for (int f = 0; f < nfields; ++f) {
        a.dst[f] = reinterpret_cast<unsigned *>(dst[f]);
        R4R_REQUIRE(a.dst[f], "dp_unpack: field %d: null destination", f);
    }
    a.blocks = static_cast<const unsigned *>(blocks); a.world = world;
    dp_unpack_kernel<<<dim3((unsigned)cdiv((int64_t)world * B_pad * maxw, 256), (unsigned)nfields), 256, 0, as_stream(stream)>>>(a);
    return check_launch("dp_unpack");
}
