// r4r_batch_build: the five index tensors of a review batch, built on the device from HBM-resident
// token pools in ONE launch (include/r4r.h; reviews4rec_amd/data.py holds the layout).
//
// Replaces the per-batch host work of the reference's loader: remove_overlap + pad_and_join /
// pad_only + the neighbour-list padding (data.py:144-236, 273-279) and the eight LongTensor
// constructions + H2D copies per batch (data.py:293-301, data_fast.py:101-109).
//
// Pure integer / byte work, HBM-bound by its OUTPUT (cfg3: 3 x 128 x 1000 int64 = 3 MB per batch
// against ~0.4 MB of int32 tokens read): one workgroup per (rating, slot); lanes walk the row's
// output positions, so the stores are fully coalesced 8-byte writes and the token reads are
// contiguous runs (a document is one token range with at most one hole).
#include "common.h"

namespace r4r {

struct PoolArgs {
    const int32_t *tok;
    const int64_t *rev_off;
    const int64_t *first;
    const int64_t *nb;
};

struct BatchArgs {
    PoolArgs users, items;
    const int32_t *held_tok;
    const int64_t *held_off;
    const int64_t *u, *i, *nb_item, *ku, *ki, *held;
    int64_t *out;
    int64_t n;
    int T, R, W, train;
    int64_t pad_user, pad_item;
};

constexpr int NEIGHBOURS = 10;

// tokens of reviews [rb, re) of a pool minus review rb + skip, as a [T] document
__device__ __forceinline__ void write_doc(const int32_t *tok, const int64_t *rev_off, int64_t rb, int64_t re,
                                          int64_t skip, int T, int64_t *dst) {
    if (skip >= re - rb) skip = -1;      // an index past this owner's list removes nothing (data.py:223-235: `i != indices[k]` holds for every i)
    const int64_t start = rev_off[rb], end = rev_off[re > rb ? re : rb];
    int64_t hb = end, hl = 0;
    if (skip >= 0) {
        hb = rev_off[rb + skip];
        hl = rev_off[rb + skip + 1] - hb;
    }
    for (int p = threadIdx.x; p < T; p += blockDim.x) {
        int64_t src = start + p;
        if (src >= hb) src += hl;
        dst[p] = src < end ? (int64_t)tok[src] : 0;
    }
}

// the same reviews in NARRE's [R, W] layout: slot r = the r-th remaining review, cut / padded to W
__device__ __forceinline__ void write_reviews(const int32_t *tok, const int64_t *rev_off, int64_t rb, int64_t re,
                                              int64_t skip, int R, int W, int64_t *dst) {
    if (skip >= re - rb) skip = -1;
    for (int p = threadIdx.x; p < R * W; p += blockDim.x) {
        const int r = p / W, w = p - r * W;
        const int64_t rev = rb + r + ((skip >= 0 && r >= skip) ? 1 : 0);
        int64_t v = 0;
        if (rev < re) {
            const int64_t rs = rev_off[rev];
            if (w < rev_off[rev + 1] - rs) v = tok[rs + w];
        }
        dst[p] = v;
    }
}

__device__ __forceinline__ void write_neighbours(const int64_t *nb, int64_t rb, int64_t re, int64_t skip,
                                                 int64_t pad, int64_t *dst) {
    if (skip >= re - rb) skip = -1;
    if (threadIdx.x < NEIGHBOURS) {
        const int r = threadIdx.x;
        const int64_t at = rb + r + ((skip >= 0 && r >= skip) ? 1 : 0);
        dst[r] = at < re ? nb[at] : pad;
    }
}

// grid (n, 5): slot 0 = the rating's own review, 1 = users who reviewed the item, 2 = items the user
// reviewed, 3 = the user's document, 4 = the item's document (data_fast.py:101-105 order).
// Output block: [n, doc] [n, 10] [n, 10] [n, doc] [n, doc] int64, doc = T or R * W.
__global__ void __launch_bounds__(256) batch_build_kernel(const BatchArgs a) {
    const int64_t row = blockIdx.x;
    const int slot = blockIdx.y;
    const bool narre = a.R > 0;
    const int64_t doc = narre ? (int64_t)a.R * a.W : a.T;
    const int64_t n = a.n;
    const int64_t u = a.u[row], ku = a.ku[row], ki = a.ki[row];
    if (slot == 1) {                     // i_to_u_map of the pair's own item, minus the rating's entry
        const int64_t it = a.nb_item[row];
        write_neighbours(a.items.nb, a.items.first[it], a.items.first[it + 1], ki, a.pad_user,
                         a.out + n * doc + row * NEIGHBOURS);
        return;
    }
    if (slot == 2) {
        write_neighbours(a.users.nb, a.users.first[u], a.users.first[u + 1], ku, a.pad_item,
                         a.out + n * doc + n * NEIGHBOURS + row * NEIGHBOURS);
        return;
    }
    const int32_t *tok;
    const int64_t *off;
    int64_t rb, re, skip, *dst;
    if (slot == 0) {
        dst = a.out + row * doc;
        skip = -1;
        if (a.train) {                   // data.py:221: the user's review ku
            tok = a.users.tok; off = a.users.rev_off;
            rb = a.users.first[u] + ku; re = rb + 1;
        } else {                         // data.py:243-245: the held-out review, or [0]
            const int64_t h = a.held[row];
            tok = a.held_tok; off = a.held_off;
            rb = h >= 0 ? h : 0; re = h >= 0 ? h + 1 : 0;
        }
    } else if (slot == 3) {
        dst = a.out + n * doc + 2 * n * NEIGHBOURS + row * doc;
        tok = a.users.tok; off = a.users.rev_off;
        rb = a.users.first[u]; re = a.users.first[u + 1]; skip = ku;
    } else {
        const int64_t it = a.i[row];
        dst = a.out + 2 * n * doc + 2 * n * NEIGHBOURS + row * doc;
        tok = a.items.tok; off = a.items.rev_off;
        rb = a.items.first[it]; re = a.items.first[it + 1]; skip = ki;
    }
    if (narre) write_reviews(tok, off, rb, re, skip, a.R, a.W, dst);
    else write_doc(tok, off, rb, re, skip, a.T, dst);
}

}  // namespace r4r

extern "C" int r4r_batch_build(const int32_t *user_tok, const int64_t *user_rev_off, const int64_t *user_first,
                               const int64_t *user_nb, const int32_t *item_tok, const int64_t *item_rev_off,
                               const int64_t *item_first, const int64_t *item_nb, const int32_t *held_tok,
                               const int64_t *held_off, const int64_t *u, const int64_t *i, const int64_t *nb_item,
                               const int64_t *ku, const int64_t *ki, const int64_t *held, int train, int64_t *out,
                               int64_t n, int T, int R, int W, int64_t pad_user, int64_t pad_item, void *stream) {
    using namespace r4r;
    if (n == 0) return R4R_OK;
    R4R_REQUIRE(user_tok && user_rev_off && user_first && user_nb && item_tok && item_rev_off && item_first &&
                    item_nb && held_tok && held_off && u && i && nb_item && ku && ki && held && out,
                "r4r_batch_build: null pointer");
    R4R_REQUIRE(n > 0 && n < (1ll << 31) && ((R > 0 && W > 0) || (R == 0 && T > 0)),
                "r4r_batch_build: bad shape n=%lld T=%d R=%d W=%d", (long long)n, T, R, W);
    BatchArgs a;
    a.users = {user_tok, user_rev_off, user_first, user_nb};
    a.items = {item_tok, item_rev_off, item_first, item_nb};
    a.held_tok = held_tok; a.held_off = held_off;
    a.u = u; a.i = i; a.nb_item = nb_item; a.ku = ku; a.ki = ki; a.held = held;
    a.out = out; a.n = n; a.T = T; a.R = R; a.W = W; a.train = train;
    a.pad_user = pad_user; a.pad_item = pad_item;
    hipLaunchKernelGGL(batch_build_kernel, dim3((unsigned)n, 5), dim3(256), 0, as_stream(stream), a);
    return check_launch("r4r_batch_build");
}
