// K training steps per host call (include/r4r.h, "spans"): the loop body of main.py:23-60 --
//     for data, y in reader.iter(): zero_grad, forward, loss, backward, optimizer.step()
// -- enqueued from C for K consecutive FULL batches of an epoch: the loader's batch construction
// (r4r_batch_build) for a GROUP of batches per launch, then the family's native step per batch with
// its per-step values (batch pointers, Adam step, dropout stream position, token buffer, sweep
// schedule) advanced here exactly as reviews4rec_amd/engine.py advances them between two
// train_step calls.  Host code only: every kernel is the one the per-step entry points launch, so
// a span and K single steps give identical bits (tests/test_gpu_span.py).
//
// Why groups: one r4r_batch_build is a chain of four dependent loads (rating -> owner -> review
// offsets -> tokens), ~6 us whatever it builds; G batches per launch cost little more than one,
// and the next group is enqueued before the last step of the current one, so no step waits for it
// beyond its share.  The ring holds two groups; a group's slot is last read by the final step of
// the group before the previous one's successor -- everything is ordered by the ONE stream.
#include "common.h"

namespace r4r {

constexpr int NB = 10;                                      // neighbour ids per side (data.py:273-279)

struct SpanLoader {
    const int32_t *user_tok, *item_tok, *held_tok;
    const int64_t *user_rev_off, *user_first, *user_nb, *item_rev_off, *item_first, *item_nb, *held_off;
    const int64_t *u, *i, *ku, *ki, *held;
    const float *y;
    int train, T, R, W;
    int64_t pad_user, pad_item;
    int64_t *ring;
    int64_t stride, N, G, B;
    const uint64_t *resident;                               // HOST table [G][8] of built batches, cycled (word 27)
    bool review;                                            // false: ids only (iter_simple, data.py:336-372)
    int64_t doc() const { return R > 0 ? (int64_t)R * W : T; }
    int64_t full_batches() const { return B > 0 ? N / B : 0; }
    int64_t group_ratings(int64_t g) const {                // ratings of the full batches in group g
        const int64_t left = full_batches() * B - g * G * B;
        return left < G * B ? left : G * B;
    }
};

struct SpanBatch {                                          // device pointers of one batch (the 7-slot list + y)
    const int64_t *this_doc, *who, *what, *user_doc, *item_doc, *uid, *iid;
    const float *y;
};

static int decode_loader(const uint64_t *w, SpanLoader &L) {
    R4R_REQUIRE(w, "span: null loader descriptor");
    auto p32 = [&](int k) { return reinterpret_cast<const int32_t *>(w[k]); };
    auto p64 = [&](int k) { return reinterpret_cast<const int64_t *>(w[k]); };
    L.user_tok = p32(0); L.user_rev_off = p64(1); L.user_first = p64(2); L.user_nb = p64(3);
    L.item_tok = p32(4); L.item_rev_off = p64(5); L.item_first = p64(6); L.item_nb = p64(7);
    L.held_tok = p32(8); L.held_off = p64(9);
    L.u = p64(10); L.i = p64(11); L.ku = p64(12); L.ki = p64(13); L.held = p64(14);
    L.y = reinterpret_cast<const float *>(w[15]);
    L.train = (int)w[16]; L.T = (int)w[17]; L.R = (int)w[18]; L.W = (int)w[19];
    L.pad_user = (int64_t)w[20]; L.pad_item = (int64_t)w[21];
    L.ring = reinterpret_cast<int64_t *>(w[22]);
    L.stride = (int64_t)w[23]; L.N = (int64_t)w[24]; L.G = (int64_t)w[25]; L.B = (int64_t)w[26];
    L.resident = reinterpret_cast<const uint64_t *>(w[27]);
    L.review = L.ring != nullptr;
    if (L.resident) {
        R4R_REQUIRE(!L.ring && L.G > 0 && L.B > 0 && L.N >= 0, "span: a resident descriptor has a table of G > 0 batches and no ring");
        return R4R_OK;
    }
    R4R_REQUIRE(L.u && L.i && L.y && L.N >= 0 && L.B > 0, "span: the descriptor needs u, i, y and a batch size");
    if (L.review) {
        R4R_REQUIRE(L.G > 0 && L.stride >= L.G * L.B * (3 * L.doc() + 2 * NB),
                    "span: ring stride %lld < a group of %lld x %lld ratings", (long long)L.stride, (long long)L.G,
                    (long long)L.B);
        R4R_REQUIRE(L.ku && L.ki && L.held, "span: a review loader needs ku, ki, held");
    }
    return R4R_OK;
}

static int build_group(const SpanLoader &L, int64_t g, hipStream_t st) {
    const int64_t at = g * L.G * L.B, n = L.group_ratings(g);
    return r4r_batch_build(L.user_tok, L.user_rev_off, L.user_first, L.user_nb, L.item_tok, L.item_rev_off,
                           L.item_first, L.item_nb, L.held_tok, L.held_off, L.u + at, L.i + at, L.i + at, L.ku + at,
                           L.ki + at, L.held + at, L.train, L.ring + (g & 1) * L.stride, n, L.T, L.R, L.W, L.pad_user,
                           L.pad_item, st);
}

static SpanBatch batch_at(const SpanLoader &L, int64_t b) {
    SpanBatch s{};
    if (L.resident) {
        const uint64_t *t = L.resident + (b % L.G) * 8;
        auto p64 = [&](int k) { return reinterpret_cast<const int64_t *>(t[k]); };
        s.this_doc = p64(0); s.who = p64(1); s.what = p64(2); s.user_doc = p64(3); s.item_doc = p64(4);
        s.uid = p64(5); s.iid = p64(6); s.y = reinterpret_cast<const float *>(t[7]);
        return s;
    }
    s.uid = L.u + b * L.B; s.iid = L.i + b * L.B; s.y = L.y + b * L.B;
    if (!L.review) return s;
    const int64_t g = b / L.G, j = b - g * L.G, n = L.group_ratings(g), doc = L.doc();
    const int64_t *base = L.ring + (g & 1) * L.stride;
    s.this_doc = base + j * L.B * doc;
    s.who = base + n * doc + j * L.B * NB;
    s.what = base + n * doc + n * NB + j * L.B * NB;
    s.user_doc = base + n * doc + 2 * n * NB + j * L.B * doc;
    s.item_doc = base + 2 * n * doc + 2 * n * NB + j * L.B * doc;
    return s;
}

// step(k, batch, next batch or nullptr) -> rc.  `announce`: a full batch follows the span and the caller wants it
// announced to the last step (its token marks ride on that step's backward launch).
template <class Step>
static int run_span(const uint64_t *loader, int64_t first_batch, int64_t steps, int announce, int64_t *built_group,
                    int64_t *steps_done, hipStream_t st, Step step) {
    SpanLoader L;
    if (int rc = decode_loader(loader, L)) return rc;
    R4R_REQUIRE(first_batch >= 0 && steps >= 0 && first_batch + steps + (announce ? 1 : 0) <= L.full_batches(),
                "span: batches %lld .. %lld (+%d announced) of %lld full ones", (long long)first_batch,
                (long long)(first_batch + steps), announce ? 1 : 0, (long long)L.full_batches());
    if (steps_done) *steps_done = 0;
    R4R_REQUIRE(!L.review || built_group, "span: built_group is the ring's state, it cannot be null");
    auto need = [&](int64_t b) -> int {
        if (!L.review) return R4R_OK;
        const int64_t g = b / L.G;
        if (*built_group >= g) return R4R_OK;
        // (a span never skips a group: the slot of g was last read by a step before the one being enqueued)
        if (int rc = build_group(L, g, st)) return rc;
        *built_group = g;
        return R4R_OK;
    };
    for (int64_t k = 0; k < steps; ++k) {
        const int64_t b = first_batch + k;
        const bool has_next = k + 1 < steps || announce;
        if (int rc = need(b)) return rc;
        if (has_next)
            if (int rc = need(b + 1)) return rc;
        const SpanBatch cur = batch_at(L, b), nxt = has_next ? batch_at(L, b + 1) : SpanBatch{};
        if (int rc = step(k, cur, has_next ? &nxt : nullptr)) return rc;
        if (steps_done) *steps_done = k + 1;
    }
    return R4R_OK;
}

struct Sweep {                                              // engine.py _SweepSchedule, one step at a time
    int period, want;
    int64_t base;
    int all() const { return (want != period || period == 1) ? 1 : 0; }
    void done(int64_t step) { if (all()) { base = step; period = want; } }
};

}  // namespace r4r

using namespace r4r;

extern "C" int r4r_span_build(const uint64_t *loader, int64_t batch, int64_t *built_group, void *stream) {
    SpanLoader L;
    if (int rc = decode_loader(loader, L)) return rc;
    R4R_REQUIRE(L.review && built_group && batch >= 0 && batch < L.full_batches(), "span_build: bad batch %lld",
                (long long)batch);
    const int64_t g = batch / L.G;
    if (*built_group >= g) return R4R_OK;
    if (int rc = build_group(L, g, as_stream(stream))) return rc;
    *built_group = g;
    return R4R_OK;
}

extern "C" int r4r_span_batch(const uint64_t *loader, int64_t batch, uint64_t *slots) {
    SpanLoader L;
    if (int rc = decode_loader(loader, L)) return rc;
    R4R_REQUIRE(slots && batch >= 0 && batch < L.full_batches(), "span_batch: bad batch %lld", (long long)batch);
    const SpanBatch s = batch_at(L, batch);
    const void *p[8] = {s.this_doc, s.who, s.what, s.user_doc, s.item_doc, s.uid, s.iid, s.y};
    for (int k = 0; k < 8; ++k) slots[k] = reinterpret_cast<uint64_t>(p[k]);
    return R4R_OK;
}

extern "C" int r4r_deepconn_span(const uint64_t *loader, int64_t first_batch, int64_t steps, int announce,
                                 int64_t *built_group, int64_t *steps_done, const float *table, int64_t V, float *flat_p, float *flat_g,
                                 float *pred, float *se, float *sse_accum, void *ws, size_t ws_bytes, int T, int E,
                                 int L, float dropout_p, int training, uint64_t seed, uint64_t offset, uint64_t draws_per_step,
                                 float inv_denom, int conv_algo, int token_buffer, int tokens_ready, float *flat_m, float *flat_v,
                                 float lr, double beta1, double beta2, float eps, float weight_decay,
                                 int64_t adam_step, void *stream) {
    R4R_REQUIRE(flat_g && flat_m && flat_v && adam_step >= 1, "deepconn_span: training steps only (gradients, moments, adam_step >= 1)");
    const int64_t B = loader ? (int64_t)loader[26] : 0;
    return run_span(loader, first_batch, steps, announce, built_group, steps_done, as_stream(stream),
                    [&](int64_t k, const SpanBatch &c, const SpanBatch *n) {
                        return r4r_deepconn_step(table, V, c.user_doc, c.item_doc, c.y, flat_p, flat_g, pred, se,
                                                 sse_accum, ws, ws_bytes, B, T, E, L, dropout_p, training, seed,
                                                 offset + (uint64_t)k * draws_per_step, inv_denom, conv_algo,
                                                 (token_buffer + (int)k) & 1, k ? 1 : tokens_ready,
                                                 n ? n->user_doc : nullptr, n ? n->item_doc : nullptr, flat_m, flat_v,
                                                 lr, beta1, beta2, eps, weight_decay, adam_step + k, stream);
                    });
}

// ---- data parallel: the headline family's span with the gradient exchange between every step's
// gradients and its update, all of it enqueued from C.  The collective is RCCL's own entry point, called through the
// address the caller resolved in the library it already holds (reviews4rec_amd/dist.py: StreamRccl) on the
// communicator it built -- no second copy of RCCL in the process, nothing linked here.
typedef int (*rccl_allreduce_fn)(const void *, void *, size_t, int, int, void *, void *);
typedef int (*rccl_allgather_fn)(const void *, void *, size_t, int, void *, void *);
constexpr int RCCL_FLOAT32 = 7, RCCL_SUM = 0;              // ncclDataType_t / ncclRedOp_t (rccl.h)

extern "C" int r4r_deepconn_span_dp(const uint64_t *loader, int64_t first_batch, int64_t steps, int announce,
                                    int64_t *built_group, int64_t *steps_done, const float *table, int64_t V,
                                    float *flat_p, float *flat_g, float *pred, float *se, float *sse_accum, void *ws,
                                    size_t ws_bytes, int T, int E, int L, float dropout_p, int training, uint64_t seed,
                                    uint64_t offset, uint64_t draws_per_step, float inv_denom, int conv_algo,
                                    int token_buffer, int tokens_ready, float *flat_m, float *flat_v, int64_t total,
                                    float lr, double beta1, double beta2, float eps, float weight_decay,
                                    int64_t adam_step, int exchange, void *collective, void *comm, int world,
                                    float *gathered, void *stream) {
    R4R_REQUIRE(flat_g && flat_m && flat_v && adam_step >= 1 && total > 0, "deepconn_span_dp: training steps only");
    R4R_REQUIRE(collective && comm && world >= 1 && (exchange == 0 || (exchange == 1 && gathered)),
                "deepconn_span_dp: exchange 0 (all-reduce) or 1 (all-gather into `gathered` [world][total]) with RCCL's "
                "entry point and a communicator");
    const int64_t B = loader ? (int64_t)loader[26] : 0;
    return run_span(
        loader, first_batch, steps, announce, built_group, steps_done, as_stream(stream),
        [&](int64_t k, const SpanBatch &c, const SpanBatch *n) {
            if (int rc = r4r_deepconn_step(table, V, c.user_doc, c.item_doc, c.y, flat_p, flat_g, pred, se, sse_accum, ws,
                                           ws_bytes, B, T, E, L, dropout_p, training, seed,
                                           offset + (uint64_t)k * draws_per_step, inv_denom, conv_algo,
                                           (token_buffer + (int)k) & 1, k ? 1 : tokens_ready, n ? n->user_doc : nullptr,
                                           n ? n->item_doc : nullptr, nullptr, nullptr, lr, beta1, beta2, eps,
                                           weight_decay, 0, stream))
                return rc;
            if (exchange == 0) {
                const int rc = reinterpret_cast<rccl_allreduce_fn>(collective)(flat_g, flat_g, (size_t)total, RCCL_FLOAT32,
                                                                              RCCL_SUM, comm, stream);
                if (rc) { set_error("deepconn_span_dp: ncclAllReduce returned %d", rc); return R4R_ERR_LAUNCH; }
                const uint64_t p1[1] = {(uint64_t)flat_p}, g1[1] = {(uint64_t)flat_g}, m1[1] = {(uint64_t)flat_m},
                               v1[1] = {(uint64_t)flat_v};
                const int64_t n1[1] = {total};
                return r4r_adam_multi(1, p1, g1, m1, v1, n1, lr, beta1, beta2, eps, weight_decay, adam_step + k, nullptr,
                                      stream);
            }
            const int rc = reinterpret_cast<rccl_allgather_fn>(collective)(flat_g, gathered, (size_t)total, RCCL_FLOAT32,
                                                                          comm, stream);
            if (rc) { set_error("deepconn_span_dp: ncclAllGather returned %d", rc); return R4R_ERR_LAUNCH; }
            return r4r_adam_gathered(flat_p, gathered, world, flat_g, flat_m, flat_v, total, lr, beta1, beta2, eps,
                                     weight_decay, adam_step + k, stream);
        });
}

extern "C" int r4r_deepconnpp_span(const uint64_t *loader, int64_t first_batch, int64_t steps, int announce,
                                   int64_t *built_group, int64_t *steps_done, const float *table, int64_t V, float *flat_p, float *flat_g,
                                   float *flat_m, float *flat_v, const uint64_t *rows_p, const uint64_t *rows_m,
                                   const uint64_t *rows_v, int64_t n_users, int64_t n_items, float *pred, float *se,
                                   float *sse_accum, void *ws, size_t ws_bytes, int T, int E, int L, float dropout_p,
                                   int training, uint64_t seed, uint64_t offset, uint64_t draws_per_step, float inv_denom,
                                   int conv_algo, int token_buffer, int tokens_ready, float lr, double beta1, double beta2, float eps,
                                   float weight_decay, int64_t adam_step, void *stream) {
    R4R_REQUIRE(flat_g && flat_m && flat_v && adam_step >= 1, "deepconnpp_span: training steps only");
    const int64_t B = loader ? (int64_t)loader[26] : 0;
    return run_span(loader, first_batch, steps, announce, built_group, steps_done, as_stream(stream),
                    [&](int64_t k, const SpanBatch &c, const SpanBatch *n) {
                        return r4r_deepconnpp_step(table, V, c.user_doc, c.item_doc, c.uid, c.iid, c.y, flat_p, flat_g,
                                                   flat_m, flat_v, rows_p, rows_m, rows_v, n_users, n_items, pred, se,
                                                   sse_accum, ws, ws_bytes, B, T, E, L, dropout_p, training, seed,
                                                   offset + (uint64_t)k * draws_per_step, inv_denom, conv_algo,
                                                   (token_buffer + (int)k) & 1, k ? 1 : tokens_ready,
                                                   n ? n->user_doc : nullptr, n ? n->item_doc : nullptr, lr, beta1,
                                                   beta2, eps, weight_decay, adam_step + k, stream);
                    });
}

extern "C" int r4r_narre_span(const uint64_t *loader, int64_t first_batch, int64_t steps, int announce,
                              int64_t *built_group, int64_t *steps_done, const float *table, int64_t V, float *flat_p, float *flat_g,
                              float *flat_m, float *flat_v, const uint64_t *rows_p, const uint64_t *rows_m,
                              const uint64_t *rows_v, int64_t n_users, int64_t n_items, float *pred, float *se,
                              float *sse_accum, void *ws, size_t ws_bytes, int R, int T, int E, int L,
                              float dropout_p, int training, uint64_t seed, uint64_t offset, uint64_t draws_per_step,
                              float inv_denom, int conv_algo, int token_buffer, int tokens_ready, float lr, double beta1, double beta2,
                              float eps, float weight_decay, int64_t adam_step, void *stream) {
    R4R_REQUIRE(flat_g && flat_m && flat_v && adam_step >= 1, "narre_span: training steps only");
    const int64_t B = loader ? (int64_t)loader[26] : 0;
    return run_span(loader, first_batch, steps, announce, built_group, steps_done, as_stream(stream),
                    [&](int64_t k, const SpanBatch &c, const SpanBatch *n) {
                        // NARRE.py:87-96: reviewed_items = data[2], users_who_reviewed = data[1]
                        return r4r_narre_step(table, V, c.user_doc, c.item_doc, c.what, c.who, c.uid, c.iid, c.y,
                                              flat_p, flat_g, flat_m, flat_v, rows_p, rows_m, rows_v, n_users, n_items,
                                              pred, se, sse_accum, ws, ws_bytes, B, R, T, E, L, dropout_p, training,
                                              seed, offset + (uint64_t)k * draws_per_step, inv_denom, conv_algo,
                                              (token_buffer + (int)k) & 1, k ? 1 : tokens_ready,
                                              n ? n->user_doc : nullptr, n ? n->item_doc : nullptr, lr, beta1, beta2,
                                              eps, weight_decay, adam_step + k, stream);
                    });
}

extern "C" int r4r_transnet_span(const uint64_t *loader, int64_t first_batch, int64_t steps, int announce,
                                 int64_t *built_group, int64_t *steps_done, const float *table, int64_t V, float *flat_p, float *flat_g,
                                 float *flat_m, float *flat_v, const uint64_t *rows_p, const uint64_t *rows_m,
                                 const uint64_t *rows_v, int64_t n_users, int64_t n_items, float *pred, float *se,
                                 float *sse_accum, void *ws, size_t ws_bytes, int T, int E, int L, int plus,
                                 float dropout_p, int training, uint64_t seed, uint64_t offset, uint64_t draws_per_step,
                                 float inv_denom, int conv_algo, int token_buffer, int tokens_ready, int sweep_period,
                                 int sweep_want, int64_t *sweep_base, int *sweep_period_out, float lr, double beta1,
                                 double beta2, float eps, float weight_decay, int64_t adam_step, void *stream) {
    R4R_REQUIRE(flat_g && flat_m && flat_v && adam_step >= 1 && sweep_base && sweep_period_out,
                "transnet_span: training steps only; sweep_base / sweep_period_out are the schedule's state");
    const int64_t B = loader ? (int64_t)loader[26] : 0;
    Sweep sw{sweep_period, sweep_want, *sweep_base};
    const int rc = run_span(
        loader, first_batch, steps, announce, built_group, steps_done, as_stream(stream),
        [&](int64_t k, const SpanBatch &c, const SpanBatch *n) {
            const int rc1 = r4r_transnet_step(
                table, V, c.user_doc, c.item_doc, c.this_doc, c.uid, c.iid, c.y, flat_p, flat_g, flat_m, flat_v, rows_p,
                rows_m, rows_v, n_users, n_items, pred, se, sse_accum, ws, ws_bytes, B, T, E, L, plus, dropout_p,
                training, seed, offset + (uint64_t)k * draws_per_step, inv_denom, conv_algo, (token_buffer + (int)k) & 1,
                k ? 1 : tokens_ready, n ? n->user_doc : nullptr, n ? n->item_doc : nullptr, n ? n->this_doc : nullptr,
                sw.period, sw.base, sw.all(), lr, beta1, beta2, eps, weight_decay, adam_step + k, stream);
            if (rc1 == R4R_OK) sw.done(adam_step + k);
            return rc1;
        });
    *sweep_base = sw.base;
    *sweep_period_out = sw.period;
    return rc;
}

extern "C" int r4r_mf_span(const uint64_t *loader, int64_t first_batch, int64_t steps, int64_t *steps_done, const uint64_t *p,
                           const uint64_t *m, const uint64_t *v, int64_t n_users, int64_t n_items, int D, float *pred,
                           float *se, float *sse_accum, void *ws, size_t ws_bytes, float dropout_p, int training,
                           uint64_t seed, uint64_t offset, uint64_t draws_per_step, float inv_denom, int sweep_period, int sweep_want,
                           int64_t *sweep_base, int *sweep_period_out, float lr, double beta1, double beta2, float eps,
                           float weight_decay, int64_t adam_step, void *stream) {
    R4R_REQUIRE(m && v && adam_step >= 1 && sweep_base && sweep_period_out, "mf_span: training steps only");
    const int64_t B = loader ? (int64_t)loader[26] : 0;
    Sweep sw{sweep_period, sweep_want, *sweep_base};
    const int rc = run_span(loader, first_batch, steps, 0, nullptr, steps_done, as_stream(stream),
                            [&](int64_t k, const SpanBatch &c, const SpanBatch *) {
                                const int rc1 = r4r_mf_step(c.uid, c.iid, c.y, p, m, v, n_users, n_items, D, pred, se,
                                                            sse_accum, ws, ws_bytes, B, dropout_p, training, seed,
                                                            offset + (uint64_t)k * draws_per_step, inv_denom, sw.period, sw.base,
                                                            sw.all(), lr, beta1, beta2, eps, weight_decay,
                                                            adam_step + k, stream);
                                if (rc1 == R4R_OK) sw.done(adam_step + k);
                                return rc1;
                            });
    *sweep_base = sw.base;
    *sweep_period_out = sw.period;
    return rc;
}

extern "C" int r4r_idnet_span(const uint64_t *loader, int64_t first_batch, int64_t steps, int64_t *steps_done, int variant, float *flat_p,
                              float *flat_g, float *flat_m, float *flat_v, const uint64_t *rows_p,
                              const uint64_t *rows_m, const uint64_t *rows_v, int64_t n_users, int64_t n_items,
                              float *pred, float *se, float *sse_accum, void *ws, size_t ws_bytes, int L,
                              float dropout_p, int training, uint64_t seed, uint64_t offset, uint64_t draws_per_step,
                              float inv_denom, int sweep_period, int sweep_want, int64_t *sweep_base,
                              int *sweep_period_out, float lr, double beta1, double beta2, float eps,
                              float weight_decay, int64_t adam_step, void *stream) {
    R4R_REQUIRE(flat_g && flat_m && flat_v && adam_step >= 1 && sweep_base && sweep_period_out,
                "idnet_span: training steps only");
    const int64_t B = loader ? (int64_t)loader[26] : 0;
    Sweep sw{sweep_period, sweep_want, *sweep_base};
    const int rc = run_span(loader, first_batch, steps, 0, nullptr, steps_done, as_stream(stream),
                            [&](int64_t k, const SpanBatch &c, const SpanBatch *) {
                                const int rc1 = r4r_idnet_step(variant, c.uid, c.iid, c.y, flat_p, flat_g, flat_m, flat_v,
                                                               rows_p, rows_m, rows_v, n_users, n_items, pred, se,
                                                               sse_accum, ws, ws_bytes, B, L, dropout_p, training, seed,
                                                               offset + (uint64_t)k * draws_per_step, inv_denom,
                                                               sw.period, sw.base, sw.all(), lr, beta1, beta2, eps,
                                                               weight_decay, adam_step + k, stream);
                                if (rc1 == R4R_OK) sw.done(adam_step + k);
                                return rc1;
                            });
    *sweep_base = sw.base;
    *sweep_period_out = sw.period;
    return rc;
}
