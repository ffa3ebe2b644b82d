// Fused native training step for the ID-only recommenders of pytorch_models/MF.py --
// model_type 'MF_dot' (MF.py:41-58: bias gathers, two ID-embedding gathers, dropout on each,
// row dot product) and 'bias_only' (MF.py:39-46) -- with the loss (loss.py:7-11), the backward
// pass and the dense Adam update (main.py:94-96,60) in TWO launches:
//
//   1. mf_fwd_bwd_kernel   one wave per rating: gathers, Philox dropout, dot, prediction, SE,
//                          and the rating's gradient rows kept COMPACT ([B, D] per table + the
//                          scalar d loss / d pred); marks the rows it touched with the step's tag
//                          and elects, per touched row, its first and last rating (atomicMax on
//                          (step, ~rating) / (step, rating): the result does not depend on order)
//   2. mf_adam_kernel      every parameter moves every step (L2 weight decay), but
//                          the dense table gradient is never materialised.  Two kinds of workgroup:
//                          SWEEP workgroups stream every element (24 B: read p, m, v, write p, m, v)
//                          and give the rows no rating touched (tag != this step) the gradient-zero
//                          update; ENTRY waves take the ratings: the FIRST rating of a row owns it
//                          and applies the sum of the row's gradient rows -- a row named once
//                          directly (a lane group of D / 4 lanes per rating, four ratings per wave
//                          at D = 64), a row named several times by scanning the batch's ids
//                          between its first and last rating and accumulating the matches in
//                          ascending order into fixed accumulators, combined in a fixed order:
//                          deterministic, no float atomics -- and updates the table row and its
//                          bias element.
//
// The op-by-op module path needs ~25 launches and a zero-filled dense gradient per table for the
// same step (28 B/element + the fill); on Amazon-Electronics-sized tables (16.6 M parameters)
// the sweep is the whole cost, so this kernel is the HBM-bound leg of SURVEY.md 8d for MF.
#include "adam_device.h"
#include "common.h"
#include "rows_device.h"
#include "peer_device.h"
#include "trace_device.h"

namespace r4r {

constexpr int MF_MAX_D = 256;          // latent size: <= 4 elements per lane of the rating's wave
constexpr int MF_SLOTS = 5;            // user table, item table, user bias, item bias, global bias
constexpr int MF_MAX_B = 32768;        // the stand-alone row ops' entry waves keep a side's ids in LDS (4 B each: 128 KB of the CU's 160)
constexpr int MF_MAX_B_STEP = 1 << 20; // r4r_mf_step / r4r_mf_apply: owners come from the forward's election slots, ids from global memory

// Owner election: every rating atomicMax-es (step, ~rating number) into its rows' slots; the slot
// then names the row's FIRST rating of this step (any order of arrival gives the same result).
__host__ __device__ inline unsigned long long mf_first_pack(int step, int64_t k) {
    return ((unsigned long long)(unsigned)step << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)k);
}

struct MfStep {
    const int64_t *uid, *iid;          // [B]
    const float *y;                    // [B] or NULL
    float *p[MF_SLOTS], *m[MF_SLOTS], *v[MF_SLOTS];
    int64_t rows[MF_SLOTS];            // U+1, I+1, U+1, I+1, 1
    int width[MF_SLOTS];               // D, D, 1, 1, 1
    // workspace
    float *gu, *gi;                    // [B, D] compact gradient rows
    float *g;                          // [B] d mean(SE) / d pred
    float *mult;                       // [B, 2D] dropout multipliers
    int *tag_u, *tag_i;                // [U+1], [I+1]: step tag of the last step that touched the row
    unsigned long long *first_u, *first_i;   // [U+1], [I+1]: mf_first_pack of the row's first rating
    unsigned long long *last_u, *last_i;     // [U+1], [I+1]: (step << 32 | the row's last rating)
    int *uid32, *iid32;                      // [B] compact copies of the ids
    int *ctag_u = nullptr, *ctag_i = nullptr;       // per sweep chunk of the tables: the last step that touched a row in it
    float *pred, *se, *sse_accum;
    int64_t B;
    int64_t B_pad = 0;                 // data parallel: rows [B, B_pad) of the entry arrays are filled as padding (id -1)
    int register_rows = 1;             // 0: no row tags / owner election here (data parallel: done over the gathered entries)
    int D, training, want_grad, tag;
    float p_drop, inv_denom;
    uint64_t seed, offset;
    // scheduled sweep (rows_device.h): the rows a rating reads are brought to step now - 1 in registers
    MfTimeBlock tb = MfTimeBlock{};
    AdamScalars sc0 = AdamScalars{};
    int now = 0;
    // PUSH form (data parallel over peer-mapped memory, r4r_mf_grad_push): the entry arrays are written into this rank's
    // slot of EVERY rank's gathered buffer (uid32 / iid32 / g / gu / gi above are then offsets from push[r], in 4-byte units)
    char *push[PEER_MAX_WORLD];
    unsigned *flags[PEER_MAX_WORLD];   // rank r's flag array
    unsigned *arrive = nullptr;        // this rank's arrival counter (zero between launches)
    int rank = 0, world = 0;
    unsigned epoch = 0;
};

// One rating per wave.  PUSH: every store of an entry goes to all ranks' buffers.
template <bool PUSH>
__device__ __forceinline__ void mf_fwd_bwd_wave(const MfStep &a) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    // entry stores: element `i` of the entry array `arr` (a pointer, or -- PUSH -- its offset inside a rank's slot)
    auto put_f = [&](float *arr, int64_t i, float x) {
        if constexpr (PUSH) {
#pragma unroll
            for (int r = 0; r < PEER_MAX_WORLD; ++r)
                if (r < a.world) reinterpret_cast<float *>(a.push[r])[reinterpret_cast<uintptr_t>(arr) + i] = x;
        } else {
            arr[i] = x;
        }
    };
    auto put_i = [&](int *arr, int64_t i, int x) {
        if constexpr (PUSH) {
#pragma unroll
            for (int r = 0; r < PEER_MAX_WORLD; ++r)
                if (r < a.world) reinterpret_cast<int *>(a.push[r])[reinterpret_cast<uintptr_t>(arr) + i] = x;
        } else {
            arr[i] = x;
        }
    };
    if (b >= a.B) {                                         // whole wave
        if (b < a.B_pad && lane == 0) { put_i(a.uid32, b, -1); put_i(a.iid32, b, -1); put_f(a.g, b, 0.f); }
        return;
    }
    const int D = a.D;
    const int64_t u = a.uid[b], i = a.iid[b];
    const float base = (a.p[2][u] + a.p[3][i]) + a.p[4][0];
    float xu[MF_MAX_D / 64], xi[MF_MAX_D / 64], mu[MF_MAX_D / 64], mi[MF_MAX_D / 64];
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < MF_MAX_D / 64; ++k) {
        const int d = lane + 64 * k;
        xu[k] = xi[k] = 0.f;
        if (d < D) {
            xu[k] = a.p[0][u * D + d];
            xi[k] = a.p[1][i * D + d];
        }
    }
    if (a.tb.rlast_u) {
        // scheduled sweep: the pending gradient-zero updates of the two rows, in registers (nothing is written back;
        // the step loop outermost: a step's scalars are fetched once)
        float ms[2][MF_MAX_D / 64], vs[2][MF_MAX_D / 64];
        int cur[2][MF_MAX_D / 64];
#pragma unroll
        for (int k = 0; k < MF_MAX_D / 64; ++k) {
            const int d = lane + 64 * k;
            cur[0][k] = cur[1][k] = a.now;
            ms[0][k] = ms[1][k] = vs[0][k] = vs[1][k] = 0.f;
            if (d < D) {
                ms[0][k] = a.m[0][u * D + d]; vs[0][k] = a.v[0][u * D + d];
                ms[1][k] = a.m[1][i * D + d]; vs[1][k] = a.v[1][i * D + d];
                cur[0][k] = tb_current(a.tb, a.tb.rlast_u, u * D + d, u, a.now);
                cur[1][k] = tb_current(a.tb, a.tb.rlast_i, i * D + d, i, a.now);
            }
        }
#pragma unroll
        for (int j = 0; j < MF_TB_MAX - 1; ++j) {            // steps now - 7 .. now - 1
            const int sj = a.now - (MF_TB_MAX - 1 - j);
            AdamScalars sc = a.sc0;
            sc.lr_over_bc1 = a.tb.lr_bc1[j];
            sc.inv_sqrt_bc2 = a.tb.isb2[j];
#pragma unroll
            for (int k = 0; k < MF_MAX_D / 64; ++k) {
                if (sj > cur[0][k]) adam_elem_fast(xu[k], 0.f, ms[0][k], vs[0][k], sc);
                if (sj > cur[1][k]) adam_elem_fast(xi[k], 0.f, ms[1][k], vs[1][k], sc);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < MF_MAX_D / 64; ++k) {
        const int d = lane + 64 * k;
        mu[k] = mi[k] = 1.f;
        if (d < D) {
            if (a.training && a.p_drop > 0.f) {
                const float keep = 1.f / (1.f - a.p_drop);
                const uint32_t ru = philox_first_word(a.offset + (uint64_t)(b * 2 * D + d), a.seed);
                const uint32_t ri = philox_first_word(a.offset + (uint64_t)(b * 2 * D + D + d), a.seed);
                mu[k] = ((float)(ru >> 8) * (1.0f / 16777216.0f) >= a.p_drop) ? keep : 0.f;
                mi[k] = ((float)(ri >> 8) * (1.0f / 16777216.0f) >= a.p_drop) ? keep : 0.f;
            }
            xu[k] *= mu[k];
            xi[k] *= mi[k];
            part = fmaf(xu[k], xi[k], part);
            if (a.mult) { a.mult[b * 2 * D + d] = mu[k]; a.mult[b * 2 * D + D + d] = mi[k]; }
        }
    }
    const float pred = D > 0 ? base + wave_sum(part) : base;
    if (lane == 0) a.pred[b] = pred;
    if (!a.y) return;
    const float d = pred - a.y[b];
    if (lane == 0) a.se[b] = d * d;
    if (!a.want_grad) return;
    const float g = 2.f * d * a.inv_denom;
    if (lane == 0) {
        put_f(a.g, b, g);
        if (a.register_rows) {
            a.tag_u[u] = a.tag;
            a.tag_i[i] = a.tag;
            atomicMax(a.first_u + u, mf_first_pack(a.tag, b));
            atomicMax(a.first_i + i, mf_first_pack(a.tag, b));
            const unsigned long long lastv = ((unsigned long long)(unsigned)a.tag << 32) | (unsigned long long)b;
            atomicMax(a.last_u + u, lastv);
            atomicMax(a.last_i + i, lastv);
            if (a.ctag_u && D > 0) {                        // chunk tags (a row may straddle two chunks)
                a.ctag_u[u * D / MF_CHUNK] = a.tag; a.ctag_u[(u * D + D - 1) / MF_CHUNK] = a.tag;
                a.ctag_i[i * D / MF_CHUNK] = a.tag; a.ctag_i[(i * D + D - 1) / MF_CHUNK] = a.tag;
            }
        }
        put_i(a.uid32, b, (int)u);
        put_i(a.iid32, b, (int)i);
    }
#pragma unroll
    for (int k = 0; k < MF_MAX_D / 64; ++k) {
        const int dd = lane + 64 * k;
        if (dd < D) {
            put_f(a.gu, b * D + dd, g * mu[k] * xi[k]);    // d pred / d U[u, d] = mult_u * (dropped item value)
            put_f(a.gi, b * D + dd, g * mi[k] * xu[k]);
        }
    }
}

__global__ __launch_bounds__(256) void mf_fwd_bwd_kernel(MfStep a) { mf_fwd_bwd_wave<false>(a); }

// The PUSH form: when the launch's last workgroup is done, this rank's flag goes up in every rank's flag array (the
// protocol of peer.hip: system-scope release after the stores, acquire on the waiting side -- r4r_mf_apply_peer).
__global__ __launch_bounds__(256) void mf_fwd_bwd_push_kernel(MfStep a) {
    mf_fwd_bwd_wave<true>(a);
    // every thread's stores reach the peers' memories before this workgroup counts as arrived; the arrival itself is a
    // relaxed add (an agent-scope acq_rel one writes back and invalidates the L2 once per workgroup), and the workgroup
    // that arrives last fences again before it raises the flags: fence - relaxed RMW - relaxed RMW - fence orders every
    // workgroup's stores before the flag stores
    __threadfence_system();
    __syncthreads();
    __shared__ unsigned last;
    if (threadIdx.x == 0) {
        const unsigned k = __hip_atomic_fetch_add(a.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (k == gridDim.x - 1);
    }
    __syncthreads();
    if (last && threadIdx.x < (unsigned)a.world) {
        __threadfence_system();
        if (threadIdx.x == 0) __hip_atomic_store(a.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
        unsigned *f = a.flags[0];
#pragma unroll
        for (int r = 1; r < PEER_MAX_WORLD; ++r)
            if ((int)threadIdx.x == r) f = a.flags[r];
        __hip_atomic_store(f + a.rank, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

constexpr int MF_THREADS = 256;        // (MF_CHUNK, the elements per sweep workgroup of a table: rows_device.h)

constexpr int MF_CHUNK_BIAS = 1024;    // bias vectors: short workgroups, so they are not the tail

// Scalar fields only: an array member indexed by the workgroup's slot number (even through a
// chain of constant-index selects, which LLVM turns back into an indexed access) is copied to
// scratch by hipcc, and a kernel that owns scratch streamed at 3.7 instead of 5+ TB/s.
struct MfSweep {
    float *p0, *p1, *p2, *p3, *p4;
    float *m0, *m1, *m2, *m3, *m4;
    float *v0, *v1, *v2, *v3, *v4;
    int64_t n0, n1, n2, n3;            // elements of the tables and bias vectors
    int cb1, cb2, cb3, cb_global, cb_entries;   // first workgroup of slots 1..3, of the global-bias group, of the entry waves
    int n_entry_wgs, epw;              // entry workgroups (both sides); entries per wave
    const unsigned long long *first_u, *first_i;   // per row: (step << 32 | ~first rating), or NULL (entry waves scan LDS ids)
    const unsigned long long *last_u, *last_i;     // per row: (step << 32 | last rating)
    const int *uid32, *iid32;                      // the batch's ids, compact (with first_*)
    const int64_t *uid, *iid;
    const float *gu, *gi, *g, *se;
    int64_t se_n = -1;                 // entries of `se` (-1: B; data parallel: this rank's own ratings, not the gathered ones)
    float *sse_accum;
    const int *tag_u, *tag_i;
    const int *ctag_u = nullptr, *ctag_i = nullptr;   // per sweep chunk of the tables: the last step that touched a row in it (optional)
    int64_t B;
    int D, now;
    int nt = 0;                        // untouched chunks: nontemporal loads / stores (tables far larger than the Infinity Cache)
    // SCAN form (data parallel, r4r_mf_apply): the entries live in the ranks' gathered blocks -- entry e is entry
    // e % B_pad of block e / B_pad, `blk_units` 4-byte units apart; uid32 / iid32 / g / gu / gi point into block 0.
    // Nothing was registered: every workgroup finds the rows the step names by scanning the ids.
    int64_t B_pad = 0, blk_units = 0;
    int *ctag_wu = nullptr, *ctag_wi = nullptr;      // the chunk tags, for the owners to stamp (scheduled sweep)
    // ... and, when the blocks arrive over peer-mapped memory (r4r_mf_apply_peer), every workgroup first waits (bounded)
    // until all `world` ranks' flags in THIS rank's flag array have reached `wait_epoch`
    const unsigned *wait_flags = nullptr;
    unsigned *timed_out = nullptr;
    unsigned long long max_ticks = 0;
    unsigned wait_epoch = 0;
    int world = 0;
    MfTimeBlock tb = MfTimeBlock{};    // temporally blocked sweep (rows_device.h); tb.rlast_u == NULL: off
    AdamScalars s;
};

typedef float mf_f32x4 __attribute__((ext_vector_type(4)));

// offset (4-byte units) of entry e's field of `width` units per entry: compact arrays, or the gathered blocks
template <bool SCAN>
__device__ __forceinline__ int64_t mf_eoff(const MfSweep &w, int64_t e, int width) {
    if constexpr (SCAN) {
        const int r = (int)((unsigned)e / (unsigned)w.B_pad);
        return (int64_t)r * w.blk_units + (e - (int64_t)r * w.B_pad) * width;
    } else {
        return e * width;
    }
}
constexpr int MF_SCAN_MAX_B = 2048;    // gathered entries up to which r4r_mf_apply's workgroups scan the ids themselves
constexpr int MF_SCAN_WORDS = MF_CHUNK / 32 + 8;   // named-row bitmap of a sweep workgroup: <= MF_CHUNK + 2 rows

// The untouched-chunk stream of the sweep: `cnt` elements of p / m / v get the gradient-zero update, two float4
// per array and thread in flight.  NT: nontemporal accesses -- a 55 M-parameter table (1.3 GB per sweep) passes
// through L2 and the 256 MB Infinity Cache exactly once per step, caching it only evicts what could be reused
// (guide: streamed loads land ~18 % sooner under nt); a table that fits the caches (cfg2: 400 MB per sweep, a
// quarter of it served on-die) keeps the default policy.
template <bool NT>
__device__ __forceinline__ void mf_stream_chunk(float *p, float *m, float *v, int64_t cnt, int tid, const AdamScalars &sc) {
    auto ld = [](const float *a, int64_t i) {
        const mf_f32x4 *q = reinterpret_cast<const mf_f32x4 *>(a) + i;
        return NT ? __builtin_nontemporal_load(q) : *q;
    };
    auto st = [](float *a, int64_t i, mf_f32x4 x) {
        mf_f32x4 *q = reinterpret_cast<mf_f32x4 *>(a) + i;
        if (NT) __builtin_nontemporal_store(x, q);
        else *q = x;
    };
    auto upd = [&](mf_f32x4 &P, mf_f32x4 &M, mf_f32x4 &V) {  // (two elements per packed instruction: adam_pair_fast)
        const adam_f32x2 z = {0.f, 0.f};
        adam_f32x2 pa = {P[0], P[1]}, ma = {M[0], M[1]}, va = {V[0], V[1]}, pb = {P[2], P[3]}, mb = {M[2], M[3]}, vb = {V[2], V[3]};
        adam_pair_fast(pa, z, ma, va, sc);
        adam_pair_fast(pb, z, mb, vb, sc);
        P = (mf_f32x4){pa.x, pa.y, pb.x, pb.y}; M = (mf_f32x4){ma.x, ma.y, mb.x, mb.y}; V = (mf_f32x4){va.x, va.y, vb.x, vb.y};
    };
    const int64_t nvec = cnt >> 2;
    int64_t i = tid;
    for (; i + MF_THREADS < nvec; i += 2 * MF_THREADS) {
        const int64_t j = i + MF_THREADS;
        mf_f32x4 P0 = ld(p, i), P1 = ld(p, j), M0 = ld(m, i), M1 = ld(m, j), V0 = ld(v, i), V1 = ld(v, j);
        upd(P0, M0, V0);
        upd(P1, M1, V1);
        st(p, i, P0); st(m, i, M0); st(v, i, V0);
        st(p, j, P1); st(m, j, M1); st(v, j, V1);
    }
    for (; i < nvec; i += MF_THREADS) {
        mf_f32x4 P = ld(p, i), M = ld(m, i), V = ld(v, i);
        upd(P, M, V);
        st(p, i, P); st(m, i, M); st(v, i, V);
    }
    const int64_t k = (nvec << 2) + tid;                    // cnt % 4 elements at the end of a table
    if (k < cnt) {
        float P = p[k], M = m[k], V = v[k];
        adam_elem_fast(P, 0.f, M, V, sc);
        p[k] = P; m[k] = M; v[k] = V;
    }
}

// Cache policy of the untouched-chunk stream: nontemporal for a sweep of more than 3 x the Infinity Cache
// (R4R_SWEEP_NT=0 / 1 pins it for A/B runs)
static int mf_sweep_nt(int64_t table_elements) {
    static const char *e = getenv("R4R_SWEEP_NT");
    return e ? (e[0] == '1') : (table_elements * 24 > ((int64_t)768 << 20));
}

// entries per wave: one up to batch 1024, then enough that a side has <= 256 workgroups (each stages
// the side's B ids in LDS once)
static int mf_epw(int64_t B) { return (int)(B <= 1024 ? 1 : (B + 1023) / 1024); }

__host__ __device__ inline int mf_chunk(int t) { return t < 2 ? MF_CHUNK : MF_CHUNK_BIAS; }

// the wide form of the entry waves applies (see mf_entry)
__host__ __device__ inline bool mf_wide(int D, const float *p, const float *m, const float *v) {
    const int lpr = D >> 2;
    return D >= 4 && (D & 3) == 0 && lpr <= 64 && (lpr & (lpr - 1)) == 0 &&
           (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0);
}

// scheduled sweep: a row's float4 brought to step now - 1 before its entry wave applies step now's gradient
__device__ __forceinline__ void tb_catch_up4(float4 &P, float4 &M, float4 &V, int cur, int now, const AdamScalars &s,
                                             const MfTimeBlock &tb) {
    tb_f32x4 p4[1] = {{P.x, P.y, P.z, P.w}}, m4[1] = {{M.x, M.y, M.z, M.w}}, v4[1] = {{V.x, V.y, V.z, V.w}};
    const int c1[1] = {cur};
    tb_catch_up_v<1>(p4, m4, v4, c1, now - 1, now, s, tb);
    P = make_float4(p4[0][0], p4[0][1], p4[0][2], p4[0][3]); M = make_float4(m4[0][0], m4[0][1], m4[0][2], m4[0][3]);
    V = make_float4(v4[0][0], v4[0][1], v4[0][2], v4[0][3]);
}

// One entry of the batch (rating k's id on side t), run by one wave: nothing to do unless k is the
// FIRST entry of its row; the first entry owns the row and applies the sum of the row's entries.
//
// Ownership and the scan range come from the forward kernel's election slots when the caller has
// them (`first` / `last`: a row named once needs no scan at all, one named twice is scanned
// between its two ratings) -- ids are then read from the compact int32 copy in global memory,
// eight 64-id chunks per round trip; otherwise (`sid`: the side's ids staged in LDS) by scanning.
//
// The row's entries, in ascending order, are appended to a per-wave pending list (LDS) chunk by
// chunk and consumed `cap` at a time, so entry number r of the row always lands in the same lane
// group and accumulator; accumulators, then lane groups, are combined in a fixed order:
// deterministic, no atomics.
//   wide form (D % 4 == 0, D / 4 a power of two): D / 4 lanes read one gradient row as float4, so
//     the wave reads epl = 256 / D entries per load instruction, eight instructions in flight (a
//     popular item in a batch of thousands has > 1000 entries: not one dependent load each);
//   generic form: lanes are columns (lane, lane + 64, ...), four entries in flight.
template <int NACC, int DL, bool WIDE, bool SCAN = false>
__device__ __forceinline__ void mf_entry(const MfSweep &w, const int *sid, int *pl, int t, int64_t k, int lane) {
    const unsigned long long *first = t ? w.first_i : w.first_u;
    const int *gid = t ? w.iid32 : w.uid32;
    const int kc = (int)(k / 64);
    int row, c_end = (int)((w.B + 63) / 64);                // chunks [kc, c_end) hold the row's entries
    bool single = false;
    if (first) {
        row = gid[k];
        if (row < 0) return;                                // a padded entry (data-parallel gather of ragged shards)
        if (first[row] != mf_first_pack(w.now, k)) return;
        const unsigned last_k = (unsigned)(t ? w.last_i : w.last_u)[row];   // low word: the row's last rating
        single = last_k == (unsigned)k;
        c_end = (int)(last_k / 64u) + 1;
    } else {
        row = sid[k];
        if (row < 0) return;                                // a padded entry (gathered ragged shards)
        for (int c = 0; c <= kc; ++c) {                     // chunks up to k's own
            const int j = c * 64 + lane;
            const unsigned long long mask = __ballot(j < w.B && sid[j] == row);
            if (mask && (int64_t)c * 64 + (__ffsll((long long)mask) - 1) < k) return;   // an earlier entry owns the row
        }
    }
    const int D = w.D;
    const float *rows = t ? w.gi : w.gu;
    const int last_j = (int)w.B - 1;
    if constexpr (SCAN) {
        // nobody registered the step's rows: the owner stamps its row's chunk tags for the scheduled sweep's later visits
        // (a due chunk's own workgroup, in this launch, sees the row in its scan of the ids either way)
        int *ct = t ? w.ctag_wi : w.ctag_wu;
        if (ct && D > 0 && lane == 0) {
            const int64_t e0 = (int64_t)row * D;
            ct[e0 / MF_CHUNK] = w.now; ct[(e0 + D - 1) / MF_CHUNK] = w.now;
        }
    }

    // scan chunks [kc, c_end), appending matches to pl and handing `cap` of them at a time to flush(base, n)
    auto scan = [&](auto &&flush, int cap) {
        if (single) {                                       // the row's only rating: nothing to look for
            if (lane == 0) pl[0] = (int)k;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            flush(0, 1);
            return;
        }
        int np = 0;
        for (int c = kc; c < c_end; c += NACC) {
            int idv[NACC];
#pragma unroll
            for (int q = 0; q < NACC; ++q) {                // clamped addresses: the loads stay unconditional
                const int j = (c + q) * 64 + lane;
                const int jj = j < last_j ? j : last_j;
                idv[q] = first ? gid[jj] : sid[jj];
            }
#pragma unroll
            for (int q = 0; q < NACC; ++q) {
                const int j = (c + q) * 64 + lane;
                const bool match = c + q < c_end && j <= last_j && idv[q] == row;
                const unsigned long long mask = __ballot(match);
                if (!mask) continue;
                const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                 __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                if (match) pl[np + below] = j;
                np += __popcll(mask);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (np >= cap) {
                    int head = 0;
                    for (; np - head >= cap; head += cap) flush(head, cap);
                    const int left = np - head;             // < cap <= 64
                    const int keep = lane < left ? pl[head + lane] : 0;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (lane < left) pl[lane] = keep;
                    np = left;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (np) flush(0, np);
    };

    float gb;                                               // the bias element's gradient
    const int lpr = D >> 2;                                 // lanes per gradient row, a float4 each
    if (WIDE && mf_wide(D, t ? w.p1 : w.p0, t ? w.m1 : w.m0, t ? w.v1 : w.v0)) {
        const int sh = __ffs(lpr) - 1, epl = 64 >> sh;
        const int grp = lane >> sh, sub = lane & (lpr - 1);
        const int nacc = epl * NACC > 64 ? (64 / epl) : NACC, cap = nacc * epl;   // cap <= 64
        float4 acc[NACC];
        float gs[NACC];
#pragma unroll
        for (int q = 0; q < NACC; ++q) { acc[q] = make_float4(0.f, 0.f, 0.f, 0.f); gs[q] = 0.f; }
        scan([&](int base, int n) {
            float4 tmp[NACC];
            float tg[NACC];
#pragma unroll
            for (int q = 0; q < NACC; ++q) {
                const int idx = q * epl + grp;
                const int e = pl[base + ((q < nacc && idx < n) ? idx : 0)];
                tmp[q] = reinterpret_cast<const float4 *>(rows + mf_eoff<SCAN>(w, e, D))[sub];
                tg[q] = w.g ? w.g[mf_eoff<SCAN>(w, e, 1)] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < NACC; ++q) {
                const bool on = q < nacc && q * epl + grp < n;
                acc[q].x += on ? tmp[q].x : 0.f; acc[q].y += on ? tmp[q].y : 0.f;
                acc[q].z += on ? tmp[q].z : 0.f; acc[q].w += on ? tmp[q].w : 0.f;
                gs[q] += on ? tg[q] : 0.f;
            }
        }, cap);
#pragma unroll
        for (int h = NACC / 2; h > 0; h >>= 1)              // fixed tree over the accumulators
#pragma unroll
            for (int q = 0; q < h; ++q) {
                acc[q].x += acc[q + h].x; acc[q].y += acc[q + h].y; acc[q].z += acc[q + h].z; acc[q].w += acc[q + h].w;
                gs[q] += gs[q + h];
            }
        float4 G = acc[0];
        gb = gs[0];
        for (int off = lpr; off < 64; off <<= 1) {          // lane groups (a + b == b + a: every lane ends with the same bits)
            G.x += __shfl_xor(G.x, off); G.y += __shfl_xor(G.y, off);
            G.z += __shfl_xor(G.z, off); G.w += __shfl_xor(G.w, off);
            gb += __shfl_xor(gb, off);
        }
        if (grp == 0) {
            float *bp = t ? w.p1 : w.p0, *bm = t ? w.m1 : w.m0, *bv = t ? w.v1 : w.v0;
            const int64_t o = ((int64_t)row * D >> 2) + sub;
            float4 P = reinterpret_cast<float4 *>(bp)[o], M = reinterpret_cast<float4 *>(bm)[o];
            float4 V = reinterpret_cast<float4 *>(bv)[o];
            if (w.tb.rlast_u) {
                int *rl = t ? w.tb.rlast_i : w.tb.rlast_u;
                tb_catch_up4(P, M, V, tb_current(w.tb, rl, o * 4, row, w.now), w.now, w.s, w.tb);
                if (sub == 0) rl[row] = w.now;              // (every lane of the row has read it: one wave, program order)
            }
            adam_elem_fast(P.x, G.x, M.x, V.x, w.s); adam_elem_fast(P.y, G.y, M.y, V.y, w.s);
            adam_elem_fast(P.z, G.z, M.z, V.z, w.s); adam_elem_fast(P.w, G.w, M.w, V.w, w.s);
            reinterpret_cast<float4 *>(bp)[o] = P; reinterpret_cast<float4 *>(bm)[o] = M;
            reinterpret_cast<float4 *>(bv)[o] = V;
        }
    } else {
        float acc[4][DL], gsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int x = 0; x < DL; ++x) acc[q][x] = 0.f;
        scan([&](int base, int n) {
            float tmp[4][DL], tg[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = pl[base + (q < n ? q : 0)];
                const int64_t eo = mf_eoff<SCAN>(w, e, D);
#pragma unroll
                for (int x = 0; x < DL; ++x)
                    tmp[q][x] = (lane + 64 * x < D) ? rows[eo + lane + 64 * x] : 0.f;
                tg[q] = w.g ? w.g[mf_eoff<SCAN>(w, e, 1)] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int x = 0; x < DL; ++x) acc[q][x] += q < n ? tmp[q][x] : 0.f;
                gsum[q] += q < n ? tg[q] : 0.f;
            }
        }, 4);
        gb = (gsum[0] + gsum[1]) + (gsum[2] + gsum[3]);
        if (D > 0) {
            float *bp = t ? w.p1 : w.p0, *bm = t ? w.m1 : w.m0, *bv = t ? w.v1 : w.v0;
            float Px[DL], Mx[DL], Vx[DL];
            int cx[DL];
#pragma unroll
            for (int x = 0; x < DL; ++x) {
                const int col = lane + 64 * x;
                cx[x] = w.now;
                Px[x] = Mx[x] = Vx[x] = 0.f;
                if (col < D) {
                    const int64_t o = (int64_t)row * D + col;
                    Px[x] = bp[o]; Mx[x] = bm[o]; Vx[x] = bv[o];
                    if (w.tb.rlast_u) cx[x] = tb_current(w.tb, t ? w.tb.rlast_i : w.tb.rlast_u, o, row, w.now);
                }
            }
            if (w.tb.rlast_u) {                             // scheduled sweep: the row's pending gradient-zero updates first
#pragma unroll
                for (int j = 0; j < MF_TB_MAX - 1; ++j) {    // steps now - 7 .. now - 1, a step's scalars fetched once
                    const int sj = w.now - (MF_TB_MAX - 1 - j);
                    AdamScalars sc = w.s;
                    sc.lr_over_bc1 = w.tb.lr_bc1[j];
                    sc.inv_sqrt_bc2 = w.tb.isb2[j];
#pragma unroll
                    for (int x = 0; x < DL; ++x)
                        if (sj > cx[x]) adam_elem_fast(Px[x], 0.f, Mx[x], Vx[x], sc);
                }
            }
#pragma unroll
            for (int x = 0; x < DL; ++x) {
                const int col = lane + 64 * x;
                if (col < D) {
                    const int64_t o = (int64_t)row * D + col;
                    const float G = (acc[0][x] + acc[1][x]) + (acc[2][x] + acc[3][x]);
                    adam_elem_fast(Px[x], G, Mx[x], Vx[x], w.s);
                    bp[o] = Px[x]; bm[o] = Mx[x]; bv[o] = Vx[x];
                }
            }
            if (w.tb.rlast_u) {                             // (after every lane's read of it: same wave, program order)
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) (t ? w.tb.rlast_i : w.tb.rlast_u)[row] = w.now;
            }
        }
    }
    if (lane == 0 && (t ? w.p3 : w.p2)) {                   // the row's bias element (a caller may have none)
        float *bp = t ? w.p3 : w.p2, *bm = t ? w.m3 : w.m2, *bv = t ? w.v3 : w.v2;
        float P = bp[row], M = bm[row], V = bv[row];
        adam_elem_fast(P, gb, M, V, w.s);
        bp[row] = P; bm[row] = M; bv[row] = V;
    }
}

template <int NACC, int DL, bool WIDE, bool SCAN>
__device__ __forceinline__ void mf_adam_body(const MfSweep &w) {
    extern __shared__ int sid[];                            // entry waves: the side's ids
    __shared__ float red[MF_THREADS];
    __shared__ unsigned nm[SCAN ? MF_SCAN_WORDS : 1];       // SCAN, sweep workgroups: the rows of the chunk the step names
    // r4r_mf_apply_peer: lane r of the first wave polls rank r's flag -- relaxed system-scope loads and no invalidation
    // behind them (peer_device.h): the blocks are read from the fine-grained segment itself, which no cache holds
    auto wait_peers = [&]() {                               // (uniform over the workgroup)
        if constexpr (SCAN) {
            if (w.wait_flags) {
                if (threadIdx.x < (unsigned)w.world)
                    peer_wait_lane<false>(w.wait_flags, threadIdx.x, w.wait_epoch, w.timed_out, w.max_ticks);
                __syncthreads();
            }
        }
    };
    __shared__ int pend[MF_THREADS / 64][128];              // entry waves, wide form: the row's pending entries
    const int tid = threadIdx.x;
    // Entry workgroups are dispatched FIRST (the owner of a popular row is the launch's longest
    // workgroup; interleaving them with the sweep workgroups measured slower), but keep the highest
    // slot numbers.
    // (then the bias-vector chunks and the global-bias workgroup -- short, and a tail when they come last -- then
    // the table chunks)
    const int nshort = w.cb_entries - w.cb2, rest = (int)blockIdx.x - w.n_entry_wgs;
    const int bx = rest < 0 ? w.cb_entries + (int)blockIdx.x : (rest < nshort ? w.cb2 + rest : rest - nshort);
    if (bx >= w.cb_entries) {
        // ---- entry waves: 4 per workgroup, all of one side (user side's groups first)
        const int lane = tid & 63;
        const int groups = w.n_entry_wgs >> 1;
        int gi = bx - w.cb_entries;
        const int t = gi >= groups;
        if (t) gi -= groups;
        if (w.first_u) {
            // election slots: a wave takes `epw` ratings.  In the wide form a lane group
            // (D / 4 lanes, a float4 each) handles one of them on its own when it is its row's only
            // rating -- no scan, three round trips; rows with more ratings then get the whole wave,
            // one after the other.
            // (ratings wv, wv + nw, wv + 2 nw, ...: a row's FIRST rating is its owner, and popular rows
            // first appear early in the batch -- consecutive ratings would put them in the same wave)
            const int64_t base = (int64_t)gi * 4 + (tid >> 6), nw = (int64_t)(w.n_entry_wgs >> 1) * 4;
            if (base >= w.B) return;
            unsigned long long multi = 0;
            if (WIDE && w.epw > 1) {
                const int D = w.D, lpr = D >> 2, sh = __ffs(lpr) - 1;
                const int grp = lane >> sh, sub = lane & (lpr - 1);
                const int64_t k = base + grp * nw;
                const int row_raw = (t ? w.iid32 : w.uid32)[k < w.B ? k : base];
                const bool valid = k < w.B && row_raw >= 0;  // (-1: a padded entry of a gathered ragged shard)
                const int row = valid ? row_raw : 0;
                const unsigned long long f = (t ? w.first_i : w.first_u)[row], l = (t ? w.last_i : w.last_u)[row];
                const bool owner = valid && f == mf_first_pack(w.now, k);
                const bool single = owner && (unsigned)l == (unsigned)k;
                multi = __ballot(owner && !single && sub == 0);
                if (single) {
                    const float4 G = reinterpret_cast<const float4 *>((t ? w.gi : w.gu) + k * D)[sub];
                    float *bp = t ? w.p1 : w.p0, *bm = t ? w.m1 : w.m0, *bv = t ? w.v1 : w.v0;
                    const int64_t o = ((int64_t)row * D >> 2) + sub;
                    float4 P = reinterpret_cast<float4 *>(bp)[o], M = reinterpret_cast<float4 *>(bm)[o];
                    float4 V = reinterpret_cast<float4 *>(bv)[o];
                    if (w.tb.rlast_u) {
                        int *rl = t ? w.tb.rlast_i : w.tb.rlast_u;
                        tb_catch_up4(P, M, V, tb_current(w.tb, rl, o * 4, row, w.now), w.now, w.s, w.tb);
                        if (sub == 0) rl[row] = w.now;
                    }
                    adam_elem_fast(P.x, G.x, M.x, V.x, w.s); adam_elem_fast(P.y, G.y, M.y, V.y, w.s);
                    adam_elem_fast(P.z, G.z, M.z, V.z, w.s); adam_elem_fast(P.w, G.w, M.w, V.w, w.s);
                    reinterpret_cast<float4 *>(bp)[o] = P; reinterpret_cast<float4 *>(bm)[o] = M;
                    reinterpret_cast<float4 *>(bv)[o] = V;
                    if (sub == 0) {                         // the row's bias element
                        float *cp = t ? w.p3 : w.p2, *cm = t ? w.m3 : w.m2, *cv = t ? w.v3 : w.v2;
                        float Pb = cp[row], Mb = cm[row], Vb = cv[row];
                        adam_elem_fast(Pb, w.g[k], Mb, Vb, w.s);
                        cp[row] = Pb; cm[row] = Mb; cv[row] = Vb;
                    }
                }
                while (multi) {
                    const int q = (__ffsll((long long)multi) - 1) >> sh;
                    multi &= multi - 1;
                    mf_entry<NACC, DL, WIDE>(w, sid, pend[tid >> 6], t, base + q * nw, lane);
                }
            } else {
                mf_entry<NACC, DL, WIDE>(w, sid, pend[tid >> 6], t, base, lane);
            }
            return;
        }
        wait_peers();
        if constexpr (SCAN) {                               // no election slots: stage the side's ids (out of the gathered blocks)
            const int *ids32 = t ? w.iid32 : w.uid32;
            for (int64_t j = tid; j < w.B; j += MF_THREADS) sid[j] = ids32[mf_eoff<true>(w, j, 1)];
        } else {
            const int64_t *ids = t ? w.iid : w.uid;
            for (int64_t j = tid; j < w.B; j += MF_THREADS) sid[j] = (int)ids[j];
        }
        __syncthreads();
        for (int it = 0; it < w.epw; ++it) {                // entries of this wave: interleaved with its neighbours
            const int64_t k = ((int64_t)gi * w.epw + it) * 4 + (tid >> 6);
            if (k >= w.B) break;
            mf_entry<NACC, DL, WIDE, SCAN>(w, sid, pend[tid >> 6], t, k, lane);
        }
        return;
    }
    if (bx >= w.cb_global) {
        // ---- global bias (gradient = sum of d loss / d pred over the batch) + the running SE:
        // strided per-thread sums, then a fixed tree -- deterministic
        wait_peers();
        float a = 0.f, e = 0.f;
        const int64_t se_n = w.se ? (w.se_n >= 0 ? w.se_n : w.B) : 0;
        for (int64_t b = tid; b < w.B; b += MF_THREADS) a += w.g[mf_eoff<SCAN>(w, b, 1)];
        for (int64_t b = tid; b < se_n; b += MF_THREADS) e += w.se[b];
        red[tid] = a;
        __syncthreads();
        for (int off = MF_THREADS / 2; off > 0; off >>= 1) { if (tid < off) red[tid] += red[tid + off]; __syncthreads(); }
        const float gsum = red[0];
        __syncthreads();
        red[tid] = e;
        __syncthreads();
        for (int off = MF_THREADS / 2; off > 0; off >>= 1) { if (tid < off) red[tid] += red[tid + off]; __syncthreads(); }
        if (tid == 0) {
            if (w.p4) {                                     // (absent when the caller keeps its global bias elsewhere)
                float P = w.p4[0], M = w.m4[0], V = w.v4[0];
                adam_elem_fast(P, gsum, M, V, w.s);
                w.p4[0] = P; w.m4[0] = M; w.v4[0] = V;
            }
            if (w.sse_accum) w.sse_accum[0] += red[0];
        }
        return;
    }
    // ---- sweep workgroups: slot 0 user table, 1 item table, 2 user bias, 3 item bias
    const int t = (bx >= w.cb1) + (bx >= w.cb2) + (bx >= w.cb3);
    float *bp = w.p0, *bm = w.m0, *bv = w.v0;
    int64_t numel = w.n0;
    int cb = 0;
    if (t == 1) { bp = w.p1; bm = w.m1; bv = w.v1; numel = w.n1; cb = w.cb1; }
    else if (t == 2) { bp = w.p2; bm = w.m2; bv = w.v2; numel = w.n2; cb = w.cb2; }
    else if (t == 3) { bp = w.p3; bm = w.m3; bv = w.v3; numel = w.n3; cb = w.cb3; }
    const int W = t < 2 ? w.D : 1;
    int64_t ci64 = bx - cb;
    const bool sched = w.tb.rlast_u && t < 2;               // the scheduled form of the blocked sweep (rows_device.h)
    if (sched && !w.tb.flush) ci64 = tb_due_chunk(ci64, w.now, w.tb.period);   // the due chunks only
    const int64_t start = ci64 * mf_chunk(t);
    if (start >= numel) return;                             // (the last block's due chunk may lie past the table)
    int64_t cnt = numel - start;
    if (cnt > mf_chunk(t)) cnt = mf_chunk(t);
    float *p = bp + start, *m = bm + start, *v = bv + start;
    const int *tag = (t == 0 || t == 2) ? w.tag_u : w.tag_i;
    const bool aligned = (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0);
    const int *ctag = t == 0 ? w.ctag_u : (t == 1 ? w.ctag_i : nullptr);
    // Which rows of the chunk does the step name?  The row tags the forward kernel stamped -- or, SCAN, a bitmap this
    // workgroup builds from the step's ids (every id requested at once, next to nothing: B <= MF_SCAN_MAX_B).
    const int64_t nm_row0 = start / W;
    bool any_named = false;
    auto build_named = [&]() {                              // (called once the chunk's own elements have been requested)
      if constexpr (SCAN) {
        constexpr int NID = MF_SCAN_MAX_B / MF_THREADS;
        wait_peers();
        const int *ids32 = (t == 0 || t == 2) ? w.uid32 : w.iid32;
        int idv[NID];
#pragma unroll
        for (int u = 0; u < NID; ++u) {
            const int j = tid + u * MF_THREADS;
            idv[u] = j < w.B ? ids32[mf_eoff<true>(w, j, 1)] : -1;
        }
        if (tid < MF_SCAN_WORDS) nm[tid] = 0u;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NID; ++u) {
            const int64_t d = (int64_t)idv[u] - nm_row0;
            if (idv[u] >= 0 && d >= 0 && d < MF_SCAN_WORDS * 32) atomicOr(&nm[d >> 5], 1u << (d & 31));
        }
        __syncthreads();
        const int64_t row_end = (start + cnt - 1) / W;      // the chunk's last row
        unsigned anyw = 0;
        for (int q = 0; q < MF_SCAN_WORDS; ++q) {           // (uniform: every thread reads the same words)
            const int64_t lo = (int64_t)q * 32;
            if (lo > row_end - nm_row0) break;
            unsigned word = nm[q];
            const int64_t over = lo + 31 - (row_end - nm_row0);
            if (over > 0) word &= 0xffffffffu >> over;      // rows past the chunk's last do not count as the chunk's
            anyw |= word;
        }
        any_named = anyw != 0;
      }
    };
    auto named = [&](int64_t row) -> bool {
        if constexpr (SCAN) {
            const unsigned d = (unsigned)(row - nm_row0);
            return (nm[d >> 5] >> (d & 31)) & 1u;
        } else {
            return tag[row] == w.now;
        }
    };
    if (sched) {
        // through which step is the chunk current?  its last scheduled visit, or the base
        int vis = tb_prev_visit(ci64, w.now - w.tb.inc, w.tb.period);
        if (vis < w.tb.base) vis = w.tb.base;
        const int pend = w.now - vis;
        if (pend > MF_TB_MAX) { if (tid == 0) *w.tb.err = 2; return; }   // (the caller left the schedule without a flush)
        if (pend <= 0) return;
        constexpr int NV = MF_CHUNK / 4 / MF_THREADS;       // float4 per thread and array
        const int64_t nvec = cnt >> 2;
        auto ld = [&](const float *a, int64_t i) {
            const mf_f32x4 *q = reinterpret_cast<const mf_f32x4 *>(a) + i;
            return w.nt ? __builtin_nontemporal_load(q) : *q;
        };
        auto st = [&](float *a, int64_t i, mf_f32x4 x) {
            mf_f32x4 *q = reinterpret_cast<mf_f32x4 *>(a) + i;
            if (w.nt) __builtin_nontemporal_store(x, q);
            else *q = x;
        };
        // the elements are requested BEFORE the chunk tag is known (one memory round trip per workgroup instead of two)
        mf_f32x4 P[NV], M[NV], V[NV];
        int cur[NV];
        bool on[NV];
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int64_t i = tid + (int64_t)u * MF_THREADS, ii = i < nvec ? i : 0;   // (clamped; nvec == 0: handled by the tail below)
            on[u] = i < nvec;
            cur[u] = vis;
            if (nvec > 0) { P[u] = ld(p, ii); M[u] = ld(m, ii); V[u] = ld(v, ii); }
        }
        build_named();
        const bool recent = ctag[ci64] > vis || any_named;  // a row of the chunk was touched after the visit (or is, now)
        const int *rlast = t == 0 ? w.tb.rlast_u : w.tb.rlast_i;
        const bool skip_now = w.tb.inc != 0;
        const int64_t row_start = start / W;
        const unsigned col_start = (unsigned)(start - row_start * W);
        if (!recent || W % 4 == 0) {
            if (recent) {
                // rows touched since the visit are current through their own last update (rlast); rows this step
                // touches belong to their entry waves.  A float4 never straddles a row here.
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const int64_t ii = on[u] ? tid + (int64_t)u * MF_THREADS : 0;
                    const int64_t row = row_start + (col_start + (unsigned)ii * 4u) / (unsigned)W;
                    const bool tg = named(row);
                    const int rl = rlast[row];
                    if (rl > cur[u]) cur[u] = rl;
                    if (skip_now && tg) on[u] = false;
                }
            }
            if (nvec > 0) {
                int lim[NV];
#pragma unroll
                for (int u = 0; u < NV; ++u) lim[u] = on[u] ? cur[u] : w.now;
                // (the step-invariant scalars as VALUES, read once here: through the argument pointer they were re-loaded --
                // an s_load and a full scalar wait -- inside each of the loop's eight predicated blocks)
                const AdamScalars sc = w.s;
                tb_catch_up_v<NV>(P, M, V, lim, w.now, w.now, sc, w.tb);
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const int64_t i = tid + (int64_t)u * MF_THREADS;
                    if (on[u] && cur[u] < w.now) { st(p, i, P[u]); st(m, i, M[u]); st(v, i, V[u]); }
                }
            }
        } else {                                            // 5-wide ID vectors ...: element by element
            for (int64_t i = tid; i < nvec; i += MF_THREADS) {
                const mf_f32x4 Pq = reinterpret_cast<const mf_f32x4 *>(p)[i], Mq = reinterpret_cast<const mf_f32x4 *>(m)[i];
                const mf_f32x4 Vq = reinterpret_cast<const mf_f32x4 *>(v)[i];
                float Pn[4], Mn[4], Vn[4];
                int ce[4];
                bool all = true, oe[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int64_t row = row_start + (col_start + (unsigned)i * 4u + c) / (unsigned)W;
                    const int r = rlast[row];
                    ce[c] = r > vis ? r : vis;
                    oe[c] = !(skip_now && named(row)) && ce[c] < w.now;
                    all = all && oe[c];
                    if (!oe[c]) ce[c] = w.now;
                    Pn[c] = Pq[c]; Mn[c] = Mq[c]; Vn[c] = Vq[c];
                }
#pragma unroll
                for (int j = 0; j < MF_TB_MAX; ++j) {
                    const int sj = w.now - (MF_TB_MAX - 1 - j);
                    AdamScalars sc = w.s;
                    sc.lr_over_bc1 = w.tb.lr_bc1[j];
                    sc.inv_sqrt_bc2 = w.tb.isb2[j];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (sj > ce[c]) adam_elem_fast(Pn[c], 0.f, Mn[c], Vn[c], sc);
                }
                if (all) {
                    reinterpret_cast<mf_f32x4 *>(p)[i] = mf_f32x4{Pn[0], Pn[1], Pn[2], Pn[3]};
                    reinterpret_cast<mf_f32x4 *>(m)[i] = mf_f32x4{Mn[0], Mn[1], Mn[2], Mn[3]};
                    reinterpret_cast<mf_f32x4 *>(v)[i] = mf_f32x4{Vn[0], Vn[1], Vn[2], Vn[3]};
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (oe[c]) { p[4 * i + c] = Pn[c]; m[4 * i + c] = Mn[c]; v[4 * i + c] = Vn[c]; }
                }
            }
        }
        const int64_t k = (nvec << 2) + tid;                // cnt % 4 elements at the end of a table
        if (k < cnt) {
            const int64_t row = (start + k) / W;
            const int r = recent ? rlast[row] : 0, c1 = r > vis ? r : vis;
            if (!(recent && skip_now && named(row)) && c1 < w.now) {
                float Pk = p[k], Mk = m[k], Vk = v[k];
                tb_catch_up(Pk, Mk, Vk, c1, w.now, w.now, w.s, w.tb);
                p[k] = Pk; m[k] = Mk; v[k] = Vk;
            }
        }
        return;
    }
    build_named();
    if ((SCAN ? t < 2 && !any_named : (ctag && ctag[bx - cb] != w.now)) && aligned) {
        // no rating touched a row of this chunk (all but a handful of chunks of a 10^7-row table):
        // stream it -- no row tags, no row / column bookkeeping
        if (w.nt) mf_stream_chunk<true>(p, m, v, cnt, tid, w.s);
        else mf_stream_chunk<false>(p, m, v, cnt, tid, w.s);
        return;
    }
    // (row, column) of a thread's element advance incrementally: a 64-bit division per element
    // would cost more than the 24 bytes the element moves
    const int64_t row_start = start / W;
    const int col_start = (int)(start - row_start * W);
    if (W % 4 == 0 && aligned) {                            // table rows: a float4 never straddles a row
        const int64_t nvec = cnt >> 2;
        const unsigned first = col_start + tid * 4u;
        int64_t row = row_start + first / (unsigned)W;
        int col = (int)(first % (unsigned)W);
        const int step_row = (MF_THREADS * 4) / W, step_col = (MF_THREADS * 4) % W;
        // two float4 per round, every load of the round (p, m, v and the row tags) issued before
        // the first use: one memory round trip per round, eight requests in flight per lane
        int64_t i = tid;
        for (; i + MF_THREADS < nvec; i += 2 * MF_THREADS) {
            int64_t row1 = row + step_row;
            int col1 = col + step_col;
            if (col1 >= W) { col1 -= W; ++row1; }
            const int64_t j = i + MF_THREADS;
            float4 P0 = reinterpret_cast<float4 *>(p)[i], P1 = reinterpret_cast<float4 *>(p)[j];
            float4 M0 = reinterpret_cast<float4 *>(m)[i], M1 = reinterpret_cast<float4 *>(m)[j];
            float4 V0 = reinterpret_cast<float4 *>(v)[i], V1 = reinterpret_cast<float4 *>(v)[j];
            const bool n0 = named(row), n1 = named(row1);
            if (!n0) {                                      // touched rows belong to their entry wave
                adam_elem_fast(P0.x, 0.f, M0.x, V0.x, w.s); adam_elem_fast(P0.y, 0.f, M0.y, V0.y, w.s);
                adam_elem_fast(P0.z, 0.f, M0.z, V0.z, w.s); adam_elem_fast(P0.w, 0.f, M0.w, V0.w, w.s);
                reinterpret_cast<float4 *>(p)[i] = P0; reinterpret_cast<float4 *>(m)[i] = M0;
                reinterpret_cast<float4 *>(v)[i] = V0;
            }
            if (!n1) {
                adam_elem_fast(P1.x, 0.f, M1.x, V1.x, w.s); adam_elem_fast(P1.y, 0.f, M1.y, V1.y, w.s);
                adam_elem_fast(P1.z, 0.f, M1.z, V1.z, w.s); adam_elem_fast(P1.w, 0.f, M1.w, V1.w, w.s);
                reinterpret_cast<float4 *>(p)[j] = P1; reinterpret_cast<float4 *>(m)[j] = M1;
                reinterpret_cast<float4 *>(v)[j] = V1;
            }
            row = row1 + step_row;
            col = col1 + step_col;
            if (col >= W) { col -= W; ++row; }
        }
        for (; i < nvec; i += MF_THREADS) {
            float4 P = reinterpret_cast<float4 *>(p)[i];
            float4 M = reinterpret_cast<float4 *>(m)[i];
            float4 V = reinterpret_cast<float4 *>(v)[i];
            if (!named(row)) {
                adam_elem_fast(P.x, 0.f, M.x, V.x, w.s); adam_elem_fast(P.y, 0.f, M.y, V.y, w.s);
                adam_elem_fast(P.z, 0.f, M.z, V.z, w.s); adam_elem_fast(P.w, 0.f, M.w, V.w, w.s);
                reinterpret_cast<float4 *>(p)[i] = P; reinterpret_cast<float4 *>(m)[i] = M;
                reinterpret_cast<float4 *>(v)[i] = V;
            }
            row += step_row;
            col += step_col;
            if (col >= W) { col -= W; ++row; }
        }
    } else if (W >= 4 && aligned) {
        // table rows whose width is not a multiple of four (TransNet++'s 5-wide ID vectors): a float4
        // covers at most two rows.  Elements of a touched row are neither updated nor STORED (their
        // entry wave is writing them).
        const int64_t nvec = cnt >> 2;
        const unsigned first = col_start + tid * 4u;
        int64_t row = row_start + first / (unsigned)W;
        int col = (int)(first % (unsigned)W);
        const int step_row = (MF_THREADS * 4) / W, step_col = (MF_THREADS * 4) % W;
        auto finish = [&](int64_t i, float4 P, float4 M, float4 V, int col_i, bool n_lo, bool n_hi) {
            const bool f0 = !n_lo, f1 = !(col_i + 1 >= W ? n_hi : n_lo),
                       f2 = !(col_i + 2 >= W ? n_hi : n_lo), f3 = !(col_i + 3 >= W ? n_hi : n_lo);
            if (f0) adam_elem_fast(P.x, 0.f, M.x, V.x, w.s);
            if (f1) adam_elem_fast(P.y, 0.f, M.y, V.y, w.s);
            if (f2) adam_elem_fast(P.z, 0.f, M.z, V.z, w.s);
            if (f3) adam_elem_fast(P.w, 0.f, M.w, V.w, w.s);
            if (f0 && f1 && f2 && f3) {
                reinterpret_cast<float4 *>(p)[i] = P; reinterpret_cast<float4 *>(m)[i] = M;
                reinterpret_cast<float4 *>(v)[i] = V;
            } else {
                if (f0) { p[4 * i] = P.x; m[4 * i] = M.x; v[4 * i] = V.x; }
                if (f1) { p[4 * i + 1] = P.y; m[4 * i + 1] = M.y; v[4 * i + 1] = V.y; }
                if (f2) { p[4 * i + 2] = P.z; m[4 * i + 2] = M.z; v[4 * i + 2] = V.z; }
                if (f3) { p[4 * i + 3] = P.w; m[4 * i + 3] = M.w; v[4 * i + 3] = V.w; }
            }
        };
        int64_t i = tid;
        for (; i + MF_THREADS < nvec; i += 2 * MF_THREADS) {
            int64_t row1 = row + step_row;
            int col1 = col + step_col;
            if (col1 >= W) { col1 -= W; ++row1; }
            const int64_t j = i + MF_THREADS;
            const float4 P0 = reinterpret_cast<float4 *>(p)[i], P1 = reinterpret_cast<float4 *>(p)[j];
            const float4 M0 = reinterpret_cast<float4 *>(m)[i], M1 = reinterpret_cast<float4 *>(m)[j];
            const float4 V0 = reinterpret_cast<float4 *>(v)[i], V1 = reinterpret_cast<float4 *>(v)[j];
            const bool a0 = named(row), a1 = named(row + (col + 3 >= W)), b0 = named(row1), b1 = named(row1 + (col1 + 3 >= W));
            finish(i, P0, M0, V0, col, a0, a1);
            finish(j, P1, M1, V1, col1, b0, b1);
            row = row1 + step_row;
            col = col1 + step_col;
            if (col >= W) { col -= W; ++row; }
        }
        for (; i < nvec; i += MF_THREADS) {
            const float4 P = reinterpret_cast<float4 *>(p)[i], M = reinterpret_cast<float4 *>(m)[i];
            const float4 V = reinterpret_cast<float4 *>(v)[i];
            finish(i, P, M, V, col, named(row), named(row + (col + 3 >= W)));
            row += step_row;
            col += step_col;
            if (col >= W) { col -= W; ++row; }
        }
        const int64_t k = (nvec << 2) + tid;                // the chunk's last cnt % 4 elements
        if (k < cnt) {
            const int64_t r = (start + k) / W;
            float P = p[k], M = m[k], V = v[k];
            if (!named(r)) {
                adam_elem_fast(P, 0.f, M, V, w.s);
                p[k] = P; m[k] = M; v[k] = V;
            }
        }
    } else {                                                // bias vectors (W = 1) and unaligned tables
        const unsigned first = col_start + tid;
        int64_t row = row_start + first / (unsigned)W;
        int col = (int)(first % (unsigned)W);
        const int step_row = MF_THREADS / W, step_col = MF_THREADS % W;
        for (int64_t i = tid; i < cnt; i += MF_THREADS) {
            float P = p[i], M = m[i], V = v[i];
            if (!named(row)) {
                adam_elem_fast(P, 0.f, M, V, w.s);
                p[i] = P; m[i] = M; v[i] = V;
            }
            row += step_row;
            col += step_col;
            if (col >= W) { col -= W; ++row; }
        }
    }
}

BWD_TRACE_DEFINE(r4r_debug_mf_adam_trace)
// DL: elements per lane of a row in the generic entry form (1: rows of <= 64 elements); WIDE: the float4 entry forms are
// compiled in.  <.., 1, false> is the LIGHT variant -- 52 VGPRs instead of 94, seven waves per SIMD instead of five:
// every wave of the launch, the table chunks' included, is allocated what the entry waves' multi-row accumulation
// needs -- taken for rows of <= 64 elements at <= MF_LIGHT_MAX_B ratings (one rating per entry wave).
template <int NACC, int DL = 4, bool WIDE = true, bool SCAN = false>
__global__ __launch_bounds__(MF_THREADS) void mf_adam_kernel(MfSweep) {
    const MfSweep &w = kernel_args<MfSweep>();              // (fields loaded at their uses, not all up front: common.h)
    BWD_STAMP(0, wall_clock64());                           // (instrumented builds only: tools/sweep_trace.py)
    mf_adam_body<NACC, DL, WIDE, SCAN>(w);
#ifdef R4R_TRACE
    const int nshort = w.cb_entries - w.cb2, rest = (int)blockIdx.x - w.n_entry_wgs;
    const int bx = rest < 0 ? w.cb_entries + (int)blockIdx.x : (rest < nshort ? w.cb2 + rest : rest - nshort);
    BWD_STAMP(1, wall_clock64());
    BWD_STAMP(2, bx >= w.cb_entries ? 5 : bx >= w.cb_global ? 4 : bx >= w.cb2 ? 3 : 2);   // role + 1: tables, bias vectors, global, entries
#endif
}

struct MfWs {
    float *gu, *gi, *g, *mult;
    int *tag_u, *tag_i;
    unsigned long long *first_u, *first_i, *last_u, *last_i;
    int *uid32, *iid32;
    int *ctag_u, *ctag_i, *rlast_u, *rlast_i, *tb_err;   // the temporally blocked sweep's state (rows_device.h, scheduled form)
    size_t bytes, persist;             // persist: the head of the buffer that carries state across steps
};

static size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }

static MfWs mf_carve(void *ws, int64_t B, int D, int64_t n_users, int64_t n_items) {
    char *p = static_cast<char *>(ws);
    size_t o = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + o : nullptr; o += a256(bytes); return r; };
    MfWs w;
    w.tag_u = reinterpret_cast<int *>(take((size_t)n_users * 4));   // tags first: they must persist (zeroed once)
    w.tag_i = reinterpret_cast<int *>(take((size_t)n_items * 4));
    w.first_u = reinterpret_cast<unsigned long long *>(take((size_t)n_users * 8));
    w.first_i = reinterpret_cast<unsigned long long *>(take((size_t)n_items * 8));
    w.last_u = reinterpret_cast<unsigned long long *>(take((size_t)n_users * 8));
    w.last_i = reinterpret_cast<unsigned long long *>(take((size_t)n_items * 8));
    const size_t cu = (size_t)cdiv(n_users * (int64_t)D, MF_CHUNK), ci = (size_t)cdiv(n_items * (int64_t)D, MF_CHUNK);
    w.ctag_u = reinterpret_cast<int *>(take(cu * 4)); w.ctag_i = reinterpret_cast<int *>(take(ci * 4));
    w.rlast_u = reinterpret_cast<int *>(take((size_t)n_users * 4)); w.rlast_i = reinterpret_cast<int *>(take((size_t)n_items * 4));
    w.tb_err = reinterpret_cast<int *>(take(4));
    w.persist = o;
    w.uid32 = reinterpret_cast<int *>(take((size_t)B * 4));
    w.iid32 = reinterpret_cast<int *>(take((size_t)B * 4));
    w.gu = reinterpret_cast<float *>(take((size_t)B * D * 4));
    w.gi = reinterpret_cast<float *>(take((size_t)B * D * 4));
    w.g = reinterpret_cast<float *>(take((size_t)B * 4));
    w.mult = reinterpret_cast<float *>(take((size_t)B * 2 * D * 4));
    w.bytes = o;
    return w;
}

// Adam on two ID bias vectors whose gradient is d loss / d pred of the ratings that name the id
// (DeepCoNN++'s user_bias / item_bias, DeepCoNN.py:69-71): the D = 0 form of the sweep above --
// untouched elements take the gradient-zero update, a touched element the fixed-order sum of its
// ratings.  `tag_*`: per-element step tags the caller's forward kernel set to `now`.
// (more than the default 64 KB of dynamic LDS once a side has more than 16,384 ids)
constexpr int MF_LIGHT_MAX_B = 1024;   // beyond it popular rows have hundreds of entries: the wide forms' four entries per load instruction pay
static bool mf_light(int D, int64_t B) {
    return D <= 64 && B <= MF_LIGHT_MAX_B;
}
static void mf_ids_lds_attr() {
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mf_adam_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  MF_MAX_B * (int)sizeof(int));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mf_adam_kernel<4, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  MF_MAX_B * (int)sizeof(int));
        done = true;
    }
}

int mf_bias_rows_launch(float *ub, float *ub_m, float *ub_v, float *ib, float *ib_m, float *ib_v,
                        int64_t n_users, int64_t n_items, const int64_t *uid, const int64_t *iid, const float *g,
                        const int *tag_u, const int *tag_i, int64_t B, int now, const AdamScalars &sc, hipStream_t st) {
    if (B > MF_MAX_B) {
        set_error("bias rows: batch %lld > %d", (long long)B, MF_MAX_B);
        return R4R_ERR_ARG;
    }
    MfSweep sw{};
    sw.p2 = ub; sw.m2 = ub_m; sw.v2 = ub_v; sw.p3 = ib; sw.m3 = ib_m; sw.v3 = ib_v;
    sw.n0 = sw.n1 = 0; sw.n2 = n_users; sw.n3 = n_items;
    int64_t chunks = 0;
    sw.cb1 = sw.cb2 = 0;                                    // no tables: slots 0, 1 are empty
    chunks += cdiv(n_users, mf_chunk(2));
    sw.cb3 = (int)chunks;
    chunks += cdiv(n_items, mf_chunk(3));
    sw.cb_global = (int)chunks;                             // no global-bias workgroup either
    sw.cb_entries = (int)chunks;
    sw.epw = mf_epw(B);
    sw.n_entry_wgs = (int)(2 * cdiv(B, 4 * sw.epw));
    chunks += sw.n_entry_wgs;
    sw.first_u = sw.first_i = sw.last_u = sw.last_i = nullptr;
    sw.uid32 = sw.iid32 = nullptr;
    sw.uid = uid; sw.iid = iid; sw.g = g; sw.se = nullptr; sw.sse_accum = nullptr;
    sw.tag_u = tag_u; sw.tag_i = tag_i; sw.B = B; sw.D = 0; sw.now = now; sw.s = sc;
    {
            mf_ids_lds_attr();
            if (mf_light(sw.D, B)) mf_adam_kernel<4, 1, false><<<(unsigned)chunks, MF_THREADS, (size_t)B * sizeof(int), st>>>(sw);
            else mf_adam_kernel<4><<<(unsigned)chunks, MF_THREADS, (size_t)B * sizeof(int), st>>>(sw);
        }
    return check_launch("bias rows");
}

void mf_time_block_scalars(MfTimeBlock &tb, float lr, double beta1, double beta2, float eps, float weight_decay, int64_t now) {
    for (int j = 0; j < MF_TB_MAX; ++j) {
        const int64_t step = now - (MF_TB_MAX - 1 - j);
        tb.lr_bc1[j] = tb.isb2[j] = 0.f;
        if (step >= 1) {
            const AdamScalars s = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, step, nullptr);
            tb.lr_bc1[j] = s.lr_over_bc1;
            tb.isb2[j] = s.inv_sqrt_bc2;
        }
    }
}

// sweep workgroups of one table: every chunk, or -- scheduled form outside an all-chunks launch -- one per block of `period`
static int64_t mf_sweep_wgs(int64_t numel, int chunk, const MfTimeBlock &tb) {
    const int64_t nch = cdiv(numel, chunk);
    return (tb.rlast_u && !tb.flush) ? tb_due_wgs(nch, tb.period) : nch;
}
static bool mf_tb_args_ok(const MfTimeBlock *tb, const int *ctag_u, const int *ctag_i, uintptr_t all) {
    return ctag_u && ctag_i && tb->rlast_u && tb->rlast_i && tb->err && !(all & 15) && tb->period >= 1 && tb->period <= MF_TB_MAX &&
           tb->base >= 0;
}

// Adam on two ID tables of width D whose gradient rows are compact ([B, D] per table, one row per
// rating; TransNet++'s user / item vectors, TransNet.py:75-76): the sweep + entry waves above
// without bias vectors.  `tag_*`: per-row step tags the caller's forward kernel set to `now`.
int mf_table_rows_launch(float *ut, float *ut_m, float *ut_v, float *it, float *it_m, float *it_v,
                         int64_t n_users, int64_t n_items, int D, const int64_t *uid, const int64_t *iid,
                         const float *gu, const float *gi, const int *tag_u, const int *tag_i,
                         const int *ctag_u, const int *ctag_i, int64_t B, int now,
                         const AdamScalars &sc, hipStream_t st, const MfTimeBlock *tb) {
    if (B > MF_MAX_B || D < 1 || D > MF_MAX_D) {
        set_error("table rows: batch %lld > %d or width %d outside 1..%d", (long long)B, MF_MAX_B, D, MF_MAX_D);
        return R4R_ERR_ARG;
    }
    MfSweep sw{};
    if (tb) {
        const uintptr_t all = reinterpret_cast<uintptr_t>(ut) | reinterpret_cast<uintptr_t>(ut_m) | reinterpret_cast<uintptr_t>(ut_v) |
                              reinterpret_cast<uintptr_t>(it) | reinterpret_cast<uintptr_t>(it_m) | reinterpret_cast<uintptr_t>(it_v);
        if (!mf_tb_args_ok(tb, ctag_u, ctag_i, all)) {
            set_error("table rows: the temporally blocked sweep needs chunk tags, its state arrays, 16-byte aligned tables "
                      "and a period in 1..%d", MF_TB_MAX);
            return R4R_ERR_ARG;
        }
        sw.tb = *tb;
    }
    sw.p0 = ut; sw.m0 = ut_m; sw.v0 = ut_v; sw.p1 = it; sw.m1 = it_m; sw.v1 = it_v;
    sw.n0 = n_users * D; sw.n1 = n_items * D; sw.n2 = sw.n3 = 0;
    int64_t chunks = mf_sweep_wgs(sw.n0, mf_chunk(0), sw.tb);
    sw.cb1 = (int)chunks;
    chunks += mf_sweep_wgs(sw.n1, mf_chunk(1), sw.tb);
    sw.cb2 = sw.cb3 = sw.cb_global = sw.cb_entries = (int)chunks;   // no bias vectors, no global-bias workgroup
    sw.epw = mf_epw(B);
    sw.n_entry_wgs = (int)(2 * cdiv(B, 4 * sw.epw));
    chunks += sw.n_entry_wgs;
    if (chunks >= (1ll << 31)) {
        set_error("table rows: too many workgroups");
        return R4R_ERR_ARG;
    }
    sw.uid = uid; sw.iid = iid; sw.gu = gu; sw.gi = gi; sw.g = nullptr; sw.se = nullptr; sw.sse_accum = nullptr;
    sw.tag_u = tag_u; sw.tag_i = tag_i; sw.ctag_u = ctag_u; sw.ctag_i = ctag_i;
    sw.B = B; sw.D = D; sw.now = now; sw.s = sc;
    sw.nt = mf_sweep_nt(sw.n0 + sw.n1);
    {
        mf_ids_lds_attr();
        if (mf_light(sw.D, B)) launch_timed(R4R_TIMING_ADAM, mf_adam_kernel<4, 1, false>, dim3((unsigned)chunks), dim3(MF_THREADS), (size_t)B * sizeof(int), st, sw);
        else launch_timed(R4R_TIMING_ADAM, mf_adam_kernel<4>, dim3((unsigned)chunks), dim3(MF_THREADS), (size_t)B * sizeof(int), st, sw);
    }
    return check_launch("table rows");
}

// mf_table_rows_launch over the ranks' gathered blocks (data parallel; rows_device.h: mf_block's layout, `world` blocks
// of B_pad entries each, ids -1 = padding): ONE launch -- nobody tags the rows first, the workgroups find the rows the
// step names by scanning the ids (the SCAN form of the sweep), the owners stamp the chunk tags.  world * B_pad <=
// MF_SCAN_MAX_B.
int mf_table_rows_blocks_launch(float *ut, float *ut_m, float *ut_v, float *it, float *it_m, float *it_v,
                                int64_t n_users, int64_t n_items, int D, const void *blocks, int world, int64_t B_pad,
                                int *ctag_u, int *ctag_i, int now, const AdamScalars &sc, hipStream_t st, const MfTimeBlock *tb) {
    const int64_t B = (int64_t)world * B_pad;
    if (B > MF_SCAN_MAX_B || B < 1 || D < 1 || D > MF_MAX_D) {
        set_error("table rows (blocks): %lld gathered entries outside 1..%d or width %d outside 1..%d", (long long)B,
                  MF_SCAN_MAX_B, D, MF_MAX_D);
        return R4R_ERR_ARG;
    }
    MfSweep sw{};
    if (tb) {
        const uintptr_t all = reinterpret_cast<uintptr_t>(ut) | reinterpret_cast<uintptr_t>(ut_m) | reinterpret_cast<uintptr_t>(ut_v) |
                              reinterpret_cast<uintptr_t>(it) | reinterpret_cast<uintptr_t>(it_m) | reinterpret_cast<uintptr_t>(it_v);
        if (!mf_tb_args_ok(tb, ctag_u, ctag_i, all)) {
            set_error("table rows (blocks): the temporally blocked sweep needs chunk tags, its state arrays, 16-byte aligned "
                      "tables and a period in 1..%d", MF_TB_MAX);
            return R4R_ERR_ARG;
        }
        sw.tb = *tb;
        sw.ctag_u = ctag_u; sw.ctag_i = ctag_i; sw.ctag_wu = ctag_u; sw.ctag_wi = ctag_i;
    }
    sw.p0 = ut; sw.m0 = ut_m; sw.v0 = ut_v; sw.p1 = it; sw.m1 = it_m; sw.v1 = it_v;
    sw.n0 = n_users * D; sw.n1 = n_items * D; sw.n2 = sw.n3 = 0;
    int64_t chunks = mf_sweep_wgs(sw.n0, mf_chunk(0), sw.tb);
    sw.cb1 = (int)chunks;
    chunks += mf_sweep_wgs(sw.n1, mf_chunk(1), sw.tb);
    sw.cb2 = sw.cb3 = sw.cb_global = sw.cb_entries = (int)chunks;   // no bias vectors, no global-bias workgroup
    sw.epw = mf_epw(B);
    sw.n_entry_wgs = (int)(2 * cdiv(B, 4 * sw.epw));
    chunks += sw.n_entry_wgs;
    if (chunks >= (1ll << 31)) {
        set_error("table rows (blocks): too many workgroups");
        return R4R_ERR_ARG;
    }
    const MfBlock k = mf_block(B_pad, D);
    const char *b0 = static_cast<const char *>(blocks);
    sw.uid32 = reinterpret_cast<const int *>(b0 + k.uid); sw.iid32 = reinterpret_cast<const int *>(b0 + k.iid);
    sw.gu = reinterpret_cast<const float *>(b0 + k.gu); sw.gi = reinterpret_cast<const float *>(b0 + k.gi);
    sw.g = nullptr; sw.se = nullptr; sw.sse_accum = nullptr;
    sw.B_pad = B_pad; sw.blk_units = (int64_t)(k.bytes / 4);
    sw.B = B; sw.D = D; sw.now = now; sw.s = sc;
    sw.nt = mf_sweep_nt(sw.n0 + sw.n1);
    {
        const size_t lds = (size_t)B * sizeof(int);
        if (mf_light(sw.D, B)) launch_timed(R4R_TIMING_ADAM, mf_adam_kernel<4, 1, false, true>, dim3((unsigned)chunks), dim3(MF_THREADS), lds, st, sw);
        else launch_timed(R4R_TIMING_ADAM, mf_adam_kernel<4, 4, true, true>, dim3((unsigned)chunks), dim3(MF_THREADS), lds, st, sw);
    }
    return check_launch("table rows (blocks)");
}

// Both of the above in ONE launch: two ID tables of width D with compact gradient rows AND the two ID bias
// vectors whose gradient is d loss / d pred of the ratings that name the id (r4r_idnet_step: MF, GMF, MLP, NeuMF's
// first table pair).  An entry wave that owns a row updates the table row and the row's bias element together.
int mf_table_bias_rows_launch(float *ut, float *ut_m, float *ut_v, float *it, float *it_m, float *it_v,
                              float *ub, float *ub_m, float *ub_v, float *ib, float *ib_m, float *ib_v,
                              int64_t n_users, int64_t n_items, int D, const int64_t *uid, const int64_t *iid,
                              const float *gu, const float *gi, const float *g, const int *tag_u, const int *tag_i,
                              int64_t B, int now, const AdamScalars &sc, hipStream_t st,
                              const int *ctag_u, const int *ctag_i, const MfTimeBlock *tb) {
    if (B > MF_MAX_B || D < 1 || D > MF_MAX_D) {
        set_error("table + bias rows: batch %lld > %d or width %d outside 1..%d", (long long)B, MF_MAX_B, D, MF_MAX_D);
        return R4R_ERR_ARG;
    }
    MfSweep sw{};
    if (tb) {
        const uintptr_t all = reinterpret_cast<uintptr_t>(ut) | reinterpret_cast<uintptr_t>(ut_m) | reinterpret_cast<uintptr_t>(ut_v) |
                              reinterpret_cast<uintptr_t>(it) | reinterpret_cast<uintptr_t>(it_m) | reinterpret_cast<uintptr_t>(it_v);
        if (!mf_tb_args_ok(tb, ctag_u, ctag_i, all)) {
            set_error("table + bias rows: the temporally blocked sweep needs chunk tags, its state arrays, 16-byte aligned "
                      "tables and a period in 1..%d", MF_TB_MAX);
            return R4R_ERR_ARG;
        }
        sw.tb = *tb;
        sw.ctag_u = ctag_u; sw.ctag_i = ctag_i;
        sw.nt = mf_sweep_nt((int64_t)n_users * D + (int64_t)n_items * D);
    }
    sw.p0 = ut; sw.m0 = ut_m; sw.v0 = ut_v; sw.p1 = it; sw.m1 = it_m; sw.v1 = it_v;
    sw.p2 = ub; sw.m2 = ub_m; sw.v2 = ub_v; sw.p3 = ib; sw.m3 = ib_m; sw.v3 = ib_v;
    sw.n0 = n_users * D; sw.n1 = n_items * D; sw.n2 = n_users; sw.n3 = n_items;
    int64_t chunks = mf_sweep_wgs(sw.n0, mf_chunk(0), sw.tb);
    sw.cb1 = (int)chunks;
    chunks += mf_sweep_wgs(sw.n1, mf_chunk(1), sw.tb);
    sw.cb2 = (int)chunks;
    chunks += cdiv(sw.n2, mf_chunk(2));
    sw.cb3 = (int)chunks;
    chunks += cdiv(sw.n3, mf_chunk(3));
    sw.cb_global = sw.cb_entries = (int)chunks;             // no global-bias workgroup: the caller keeps it elsewhere
    sw.epw = mf_epw(B);
    sw.n_entry_wgs = (int)(2 * cdiv(B, 4 * sw.epw));
    chunks += sw.n_entry_wgs;
    if (chunks >= (1ll << 31)) {
        set_error("table + bias rows: too many workgroups");
        return R4R_ERR_ARG;
    }
    sw.uid = uid; sw.iid = iid; sw.gu = gu; sw.gi = gi; sw.g = g; sw.se = nullptr; sw.sse_accum = nullptr;
    sw.tag_u = tag_u; sw.tag_i = tag_i; sw.B = B; sw.D = D; sw.now = now; sw.s = sc;
    {
        mf_ids_lds_attr();
        if (mf_light(sw.D, B)) launch_timed(R4R_TIMING_ADAM, mf_adam_kernel<4, 1, false>, dim3((unsigned)chunks), dim3(MF_THREADS), (size_t)B * sizeof(int), st, sw);
        else launch_timed(R4R_TIMING_ADAM, mf_adam_kernel<4>, dim3((unsigned)chunks), dim3(MF_THREADS), (size_t)B * sizeof(int), st, sw);
    }
    return check_launch("table + bias rows");
}

}  // namespace r4r

using namespace r4r;

extern "C" size_t r4r_mf_ws_bytes(int64_t B, int D, int64_t n_users, int64_t n_items) {
    if (B < 0 || D < 0 || n_users <= 0 || n_items <= 0) return 0;
    return mf_carve(nullptr, B, D, n_users, n_items).bytes;
}

extern "C" size_t r4r_mf_ws_persist_bytes(int64_t B, int D, int64_t n_users, int64_t n_items) {
    if (B < 0 || D < 0 || n_users <= 0 || n_items <= 0) return 0;
    return mf_carve(nullptr, B, D, n_users, n_items).persist;
}

extern "C" size_t r4r_mf_ws_mult_offset(int64_t B, int D, int64_t n_users, int64_t n_items) {
    const MfWs w = mf_carve(reinterpret_cast<void *>(256), B, D, n_users, n_items);
    return (size_t)(reinterpret_cast<char *>(w.mult) - reinterpret_cast<char *>(256));
}

extern "C" size_t r4r_mf_ws_grad_offset(int64_t B, int D, int64_t n_users, int64_t n_items, int which) {
    const MfWs w = mf_carve(reinterpret_cast<void *>(256), B, D, n_users, n_items);
    const char *q = which == 0 ? reinterpret_cast<char *>(w.gu) : which == 1 ? reinterpret_cast<char *>(w.gi)
                                                                              : reinterpret_cast<char *>(w.g);
    return (size_t)(q - reinterpret_cast<char *>(256));
}

extern "C" int r4r_mf_step(const int64_t *uid, const int64_t *iid, const float *y,
                           const uint64_t *p, const uint64_t *m, const uint64_t *v,
                           int64_t n_users, int64_t n_items, int D,
                           float *pred, float *se, float *sse_accum, void *ws, size_t ws_bytes, int64_t B,
                           float dropout_p, int training, uint64_t seed, uint64_t offset, float inv_denom,
                           int sweep_period, int64_t sweep_base, int sweep_all,
                           float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                           void *stream) {
    R4R_REQUIRE(uid && iid && p && pred && ws, "mf_step: null pointer");
    R4R_REQUIRE(sweep_period >= 1 && sweep_period <= MF_TB_MAX, "mf_step: sweep_period %d outside 1..%d", sweep_period, MF_TB_MAX);
    R4R_REQUIRE(sweep_base >= 0 && (!m || sweep_base < adam_step), "mf_step: sweep_base %lld outside 0..adam_step - 1",
                (long long)sweep_base);
    R4R_REQUIRE(n_users > 0 && n_items > 0 && B >= 0, "mf_step: bad sizes");
    R4R_REQUIRE(D >= 0 && D <= MF_MAX_D, "mf_step: latent_size %d outside 0..%d", D, MF_MAX_D);
    R4R_REQUIRE(!m == !v, "mf_step: m and v go together");
    R4R_REQUIRE(!m || (y && se && adam_step >= 1), "mf_step: a training step needs ratings, the se buffer and "
                                                   "adam_step >= 1");
    R4R_REQUIRE(!y || se, "mf_step: se buffer required when y is given");
    R4R_REQUIRE(!m || B <= MF_MAX_B_STEP, "mf_step: batch %lld > %d", (long long)B, MF_MAX_B_STEP);
    R4R_REQUIRE(adam_step < (1ll << 31), "mf_step: step tag overflow");
    R4R_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "mf_step: dropout %f outside [0,1)", (double)dropout_p);
    if (ws_bytes < r4r_mf_ws_bytes(B, D, n_users, n_items)) {
        set_error("mf_step: workspace %zu < %zu bytes", ws_bytes, r4r_mf_ws_bytes(B, D, n_users, n_items));
        return R4R_ERR_WORKSPACE;
    }
    for (int k = (D > 0 ? 0 : 2); k < MF_SLOTS; ++k) {
        R4R_REQUIRE(p[k] && (!m || (m[k] && v[k])), "mf_step: slot %d: null parameter / moment pointer", k);
    }
    if (B == 0) return R4R_OK;
    hipStream_t st = as_stream(stream);
    const MfWs w = mf_carve(ws, B, D, n_users, n_items);
    MfStep a;
    a.uid = uid; a.iid = iid; a.y = y;
    const int64_t rows[MF_SLOTS] = {n_users, n_items, n_users, n_items, 1};
    const int width[MF_SLOTS] = {D, D, 1, 1, 1};
    for (int k = 0; k < MF_SLOTS; ++k) {
        a.p[k] = reinterpret_cast<float *>(p[k]);
        a.m[k] = m ? reinterpret_cast<float *>(m[k]) : nullptr;
        a.v[k] = v ? reinterpret_cast<float *>(v[k]) : nullptr;
        a.rows[k] = rows[k]; a.width[k] = width[k];
    }
    a.gu = w.gu; a.gi = w.gi; a.g = w.g; a.mult = w.mult; a.tag_u = w.tag_u; a.tag_i = w.tag_i;
    a.first_u = w.first_u; a.first_i = w.first_i; a.last_u = w.last_u; a.last_i = w.last_i;
    a.uid32 = w.uid32; a.iid32 = w.iid32;
    a.first_u = w.first_u; a.first_i = w.first_i; a.last_u = w.last_u; a.last_i = w.last_i;
    a.uid32 = w.uid32; a.iid32 = w.iid32;
    a.pred = pred; a.se = se; a.sse_accum = sse_accum;
    a.B = B; a.D = D; a.training = training; a.want_grad = m != nullptr; a.tag = (int)adam_step;
    a.p_drop = dropout_p; a.inv_denom = inv_denom; a.seed = seed; a.offset = offset;
    // the temporally blocked sweep (rows_device.h) applies to 16-byte aligned tables; without it the chunk tags stay unused
    const bool tb_on = m && D > 0 && (((p[0] | p[1] | m[0] | m[1] | v[0] | v[1]) & 15) == 0);
    MfTimeBlock tb{};
    if (tb_on) {
        a.ctag_u = w.ctag_u; a.ctag_i = w.ctag_i;
        tb.rlast_u = w.rlast_u; tb.rlast_i = w.rlast_i; tb.err = w.tb_err; tb.base = (int)sweep_base;
        tb.period = sweep_period; tb.flush = (sweep_all || sweep_period == 1) ? 1 : 0; tb.inc = 1;
        mf_time_block_scalars(tb, lr, beta1, beta2, eps, weight_decay, adam_step);
        a.tb = tb; a.now = (int)adam_step;
        a.sc0 = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    }
    mf_fwd_bwd_kernel<<<(unsigned)cdiv(B, 4), 256, 0, st>>>(a);
    if (!m) return check_launch("mf_step(forward)");
    MfSweep sw;
    sw.p0 = a.p[0]; sw.p1 = a.p[1]; sw.p2 = a.p[2]; sw.p3 = a.p[3]; sw.p4 = a.p[4];
    sw.m0 = a.m[0]; sw.m1 = a.m[1]; sw.m2 = a.m[2]; sw.m3 = a.m[3]; sw.m4 = a.m[4];
    sw.v0 = a.v[0]; sw.v1 = a.v[1]; sw.v2 = a.v[2]; sw.v3 = a.v[3]; sw.v4 = a.v[4];
    int64_t numel[4], begin[4], chunks = 0;
    for (int k = 0; k < 4; ++k) {                           // sweep workgroups: tables (the due chunks), bias vectors
        numel[k] = rows[k] * width[k];
        begin[k] = chunks;
        chunks += k < 2 ? mf_sweep_wgs(numel[k], mf_chunk(k), tb) : cdiv(numel[k], mf_chunk(k));
    }
    sw.n0 = numel[0]; sw.n1 = numel[1]; sw.n2 = numel[2]; sw.n3 = numel[3];
    sw.cb1 = (int)begin[1]; sw.cb2 = (int)begin[2]; sw.cb3 = (int)begin[3];
    sw.cb_global = (int)chunks;                             // one workgroup: global bias + running SE
    chunks += 1;
    sw.cb_entries = (int)chunks;                            // entry waves: 4 per workgroup, per side
    // election slots (no LDS staging to amortise): a wave takes 256 / D ratings in the wide form, else one
    const bool light = mf_light(D, B);                      // (one rating per entry wave, the generic form)
    sw.epw = (!light && D > 0 && mf_wide(D, sw.p0, sw.m0, sw.v0) && mf_wide(D, sw.p1, sw.m1, sw.v1)) ? 256 / D : 1;
    sw.n_entry_wgs = (int)(2 * cdiv(B, 4 * sw.epw));
    chunks += sw.n_entry_wgs;
    sw.first_u = w.first_u; sw.first_i = w.first_i; sw.last_u = w.last_u; sw.last_i = w.last_i;
    sw.uid32 = w.uid32; sw.iid32 = w.iid32;
    R4R_REQUIRE(chunks < (1ll << 31), "mf_step: too many workgroups");
    sw.uid = uid; sw.iid = iid; sw.gu = w.gu; sw.gi = w.gi; sw.g = w.g; sw.se = se; sw.sse_accum = sse_accum;
    sw.tag_u = w.tag_u; sw.tag_i = w.tag_i; sw.B = B; sw.D = D; sw.now = (int)adam_step;
    sw.s = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    if (tb_on) {
        sw.ctag_u = w.ctag_u; sw.ctag_i = w.ctag_i;
        sw.tb = tb;
        sw.nt = mf_sweep_nt(sw.n0 + sw.n1);
    }
    {
        // more loads in flight per entry wave (and fewer waves per SIMD: 156 vs 116 VGPRs) once rows can
        // have hundreds of ratings
        if (B > 2048) launch_timed(R4R_TIMING_ADAM, mf_adam_kernel<8>, dim3((unsigned)chunks), dim3(MF_THREADS), 0, st, sw);
        else if (light) launch_timed(R4R_TIMING_ADAM, mf_adam_kernel<4, 1, false>, dim3((unsigned)chunks), dim3(MF_THREADS), 0, st, sw);
        else launch_timed(R4R_TIMING_ADAM, mf_adam_kernel<4>, dim3((unsigned)chunks), dim3(MF_THREADS), 0, st, sw);
    }
    return check_launch("mf_step");
}

// What the temporally blocked sweep left pending (r4r_mf_step with sweep_period > 1): every chunk of the two ID tables
// takes its pending updates now.  adam_step = the LAST COMPLETED step; (sweep_period, sweep_base): the schedule in force
// since sweep_base.  The caller's base becomes adam_step.
extern "C" int r4r_mf_rows_flush(const uint64_t *p, const uint64_t *m, const uint64_t *v,
                                 int64_t n_users, int64_t n_items, int D, void *ws, size_t ws_bytes, int64_t B,
                                 int sweep_period, int64_t sweep_base,
                                 float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                                 void *stream) {
    R4R_REQUIRE(p && m && v && ws, "mf_rows_flush: null pointer");
    R4R_REQUIRE(n_users > 0 && n_items > 0 && B >= 0 && D >= 0 && D <= MF_MAX_D, "mf_rows_flush: bad sizes");
    R4R_REQUIRE(adam_step >= 0 && adam_step < (1ll << 31), "mf_rows_flush: bad adam_step");
    R4R_REQUIRE(sweep_period >= 1 && sweep_period <= MF_TB_MAX && sweep_base >= 0 && sweep_base <= adam_step,
                "mf_rows_flush: sweep_period outside 1..%d or sweep_base outside 0..adam_step", MF_TB_MAX);
    if (ws_bytes < r4r_mf_ws_bytes(B, D, n_users, n_items)) {
        set_error("mf_rows_flush: workspace %zu < %zu bytes", ws_bytes, r4r_mf_ws_bytes(B, D, n_users, n_items));
        return R4R_ERR_WORKSPACE;
    }
    if (adam_step == 0 || D == 0 || sweep_base == adam_step) return R4R_OK;   // no step yet / no tables / nothing can be pending
    for (int k = 0; k < 2; ++k) R4R_REQUIRE(p[k] && m[k] && v[k], "mf_rows_flush: table %d: null pointer", k);
    if ((p[0] | p[1] | m[0] | m[1] | v[0] | v[1]) & 15) return R4R_OK;   // (unaligned tables never defer)
    const MfWs w = mf_carve(ws, B, D, n_users, n_items);
    MfTimeBlock tb{};
    tb.rlast_u = w.rlast_u; tb.rlast_i = w.rlast_i; tb.err = w.tb_err; tb.base = (int)sweep_base;
    tb.period = sweep_period; tb.flush = 1; tb.inc = 0;
    mf_time_block_scalars(tb, lr, beta1, beta2, eps, weight_decay, adam_step);
    const AdamScalars sc = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    auto f = [](uint64_t x) { return reinterpret_cast<float *>(x); };
    return mf_table_rows_launch(f(p[0]), f(m[0]), f(v[0]), f(p[1]), f(m[1]), f(v[1]), n_users, n_items, D, nullptr, nullptr,
                                nullptr, nullptr, w.tag_u, w.tag_i, w.ctag_u, w.ctag_i, 0, (int)adam_step, sc, as_stream(stream), &tb);
}

// offset of the int the temporally blocked sweep sets if more updates were ever pending than a visit applies
extern "C" size_t r4r_mf_ws_flag_offset(int64_t B, int D, int64_t n_users, int64_t n_items) {
    const MfWs w = mf_carve(reinterpret_cast<void *>(256), B, D, n_users, n_items);
    return (size_t)(reinterpret_cast<char *>(w.tb_err) - reinterpret_cast<char *>(256));
}

// ------------------------------------------------------------------ data parallel
// One process per GPU, replicated tables.  Each rank computes the compact gradient rows of ITS
// ratings (r4r_mf_grad) into a packed block; one all_gather moves the blocks; every rank then
// applies the same update from all blocks in rank order (r4r_mf_apply): replicas stay bit-identical,
// and the result equals the single-process step on the concatenated batch (same entries, same
// order).  Block layout (bytes): uid32 [B_pad] | iid32 [B_pad] | g [B_pad] | gu [B_pad, D] | gi [B_pad, D];
// entries past a rank's own count carry id -1 (ragged shards).
namespace r4r {

struct MfRegister {
    const char *blocks;                // [world] packed blocks
    MfBlock k;
    int world, D, now;
    int64_t B_pad;
    int *tag_u, *tag_i, *uid32, *iid32;
    unsigned long long *first_u, *first_i, *last_u, *last_i;
    float *g, *gu, *gi;                // contiguous [world * B_pad] entry arrays for the update kernel
    int *ctag_u = nullptr, *ctag_i = nullptr;   // temporally blocked sweep: chunk tags (NULL: not kept)
};

// one wave per gathered entry: ids, d loss / d pred and the two gradient rows into the contiguous
// entry arrays; row tags and owner election for the real entries
__global__ __launch_bounds__(256) void mf_register_kernel(MfRegister a) {
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= (int64_t)a.world * a.B_pad) return;
    const int64_t r = e / a.B_pad, b = e - r * a.B_pad;
    const char *blk = a.blocks + (size_t)r * a.k.bytes;
    const int u = reinterpret_cast<const int *>(blk + a.k.uid)[b], i = reinterpret_cast<const int *>(blk + a.k.iid)[b];
    if (lane == 0) {
        a.uid32[e] = u; a.iid32[e] = i;
        a.g[e] = reinterpret_cast<const float *>(blk + a.k.g)[b];
        if (u >= 0) {
            a.tag_u[u] = a.now; a.tag_i[i] = a.now;
            atomicMax(a.first_u + u, mf_first_pack(a.now, e));
            atomicMax(a.first_i + i, mf_first_pack(a.now, e));
            const unsigned long long lastv = ((unsigned long long)(unsigned)a.now << 32) | (unsigned long long)e;
            atomicMax(a.last_u + u, lastv);
            atomicMax(a.last_i + i, lastv);
        }
        if (a.ctag_u && a.D > 0 && u >= 0) {                // chunk tags of every rank's ids
            const int64_t D = a.D;
            a.ctag_u[u * D / MF_CHUNK] = a.now; a.ctag_u[(u * D + D - 1) / MF_CHUNK] = a.now;
            a.ctag_i[i * D / MF_CHUNK] = a.now; a.ctag_i[(i * D + D - 1) / MF_CHUNK] = a.now;
        }
    }
    if (u >= 0)
        for (int d = lane; d < a.D; d += 64) {
            a.gu[e * a.D + d] = reinterpret_cast<const float *>(blk + a.k.gu)[b * a.D + d];
            a.gi[e * a.D + d] = reinterpret_cast<const float *>(blk + a.k.gi)[b * a.D + d];
        }
}

}  // namespace r4r

extern "C" size_t r4r_mf_dp_block_bytes(int64_t B_pad, int D) {
    if (B_pad < 0 || D < 0) return 0;
    return mf_block(B_pad, D).bytes;
}

namespace r4r {
struct MfPush {                        // r4r_mf_grad_push: where the block goes (null peer_dst: into `block`)
    const uint64_t *peer_dst = nullptr, *peer_flags = nullptr;
    uint32_t *arrive = nullptr;
    int rank = 0, world = 0;
    uint32_t epoch = 0;
};
}  // namespace r4r

static int mf_grad_impl(const int64_t *uid, const int64_t *iid, const float *y, const uint64_t *p,
                        const uint64_t *m, const uint64_t *v,
                        int64_t n_users, int64_t n_items, int D, float *pred, float *se, void *block, float *mult,
                        int64_t B, int64_t B_pad, float dropout_p, int training, uint64_t seed, uint64_t offset,
                        float inv_denom, void *ws, int sweep_period, int64_t sweep_base,
                        float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                        void *stream, const MfPush &push) {
    R4R_REQUIRE(p && pred && se && (block || push.peer_dst), "mf_grad: null pointer");
    R4R_REQUIRE(!m == !v && !m == !ws, "mf_grad: m, v and the workspace of r4r_mf_apply go together");
    R4R_REQUIRE(!m || (sweep_period >= 1 && sweep_period <= MF_TB_MAX && sweep_base >= 0 && sweep_base < adam_step &&
                       adam_step < (1ll << 31)),
                "mf_grad: sweep_period outside 1..%d, or sweep_base outside 0..adam_step - 1", MF_TB_MAX);
    R4R_REQUIRE(B == 0 || (uid && iid && y), "mf_grad: null ids / ratings");   // (an empty shard has none)
    R4R_REQUIRE(n_users > 0 && n_items > 0 && B >= 0 && B_pad >= B, "mf_grad: bad sizes");
    R4R_REQUIRE(D >= 0 && D <= MF_MAX_D, "mf_grad: latent_size %d outside 0..%d", D, MF_MAX_D);
    R4R_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "mf_grad: dropout %f outside [0,1)", (double)dropout_p);
    if (B_pad == 0) return R4R_OK;
    const MfBlock k = mf_block(B_pad, D);
    char *blk = static_cast<char *>(block);
    MfStep a{};
    a.uid = uid; a.iid = iid; a.y = y;
    for (int s = 0; s < MF_SLOTS; ++s) a.p[s] = reinterpret_cast<float *>(p[s]);
    for (int s = (D > 0 ? 0 : 2); s < MF_SLOTS; ++s) R4R_REQUIRE(a.p[s], "mf_grad: slot %d: null parameter pointer", s);
    if (push.peer_dst) {
        // the entry arrays as offsets (4-byte units) inside a rank's slot; push[r] = this rank's slot in rank r's buffer
        R4R_REQUIRE(push.peer_flags && push.arrive && push.world >= 1 && push.world <= PEER_MAX_WORLD && push.rank >= 0 &&
                    push.rank < push.world, "mf_grad_push: rank %d of %d (<= %d ranks), flags and an arrival counter", push.rank,
                    push.world, PEER_MAX_WORLD);
        for (int r = 0; r < PEER_MAX_WORLD; ++r) {
            const int rr = r < push.world ? r : 0;
            R4R_REQUIRE(push.peer_dst[rr] && push.peer_flags[rr] && (push.peer_dst[rr] & 15) == 0, "mf_grad_push: bad peer buffer %d", rr);
            a.push[r] = reinterpret_cast<char *>(push.peer_dst[rr]) + (size_t)push.rank * k.bytes;
            a.flags[r] = reinterpret_cast<unsigned *>(push.peer_flags[rr]);
        }
        a.arrive = push.arrive; a.rank = push.rank; a.world = push.world; a.epoch = push.epoch;
    }
    // (PUSH: not pointers but the arrays' offsets inside a slot, in 4-byte units -- mf_fwd_bwd_wave's put_*)
    auto field = [&](size_t off) { return push.peer_dst ? static_cast<uintptr_t>(off / 4) : reinterpret_cast<uintptr_t>(blk + off); };
    a.uid32 = reinterpret_cast<int *>(field(k.uid)); a.iid32 = reinterpret_cast<int *>(field(k.iid));
    a.g = reinterpret_cast<float *>(field(k.g)); a.gu = reinterpret_cast<float *>(field(k.gu));
    a.gi = reinterpret_cast<float *>(field(k.gi)); a.mult = mult;
    // the tables may carry pending gradient-zero updates (r4r_mf_apply's scheduled sweep): the rows a rating reads
    // are brought to step adam_step - 1 in registers
    if (m && D > 0 && sweep_period > 1 && (((p[0] | p[1] | m[0] | m[1] | v[0] | v[1]) & 15) == 0)) {
        for (int s = 0; s < 2; ++s) { a.m[s] = reinterpret_cast<float *>(m[s]); a.v[s] = reinterpret_cast<float *>(v[s]); }
        const MfWs w = mf_carve(ws, 0, D, n_users, n_items);
        a.tb.rlast_u = w.rlast_u; a.tb.rlast_i = w.rlast_i; a.tb.err = w.tb_err; a.tb.base = (int)sweep_base;
        a.tb.period = sweep_period; a.tb.inc = 1;
        mf_time_block_scalars(a.tb, lr, beta1, beta2, eps, weight_decay, adam_step);
        a.sc0 = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
        a.now = (int)adam_step;
    }
    a.pred = pred; a.se = se; a.B = B; a.B_pad = B_pad; a.register_rows = 0; a.D = D; a.training = training;
    a.want_grad = 1; a.tag = 0; a.p_drop = dropout_p; a.inv_denom = inv_denom; a.seed = seed; a.offset = offset;
    if (push.peer_dst) mf_fwd_bwd_push_kernel<<<(unsigned)cdiv(B_pad, 4), 256, 0, as_stream(stream)>>>(a);
    else mf_fwd_bwd_kernel<<<(unsigned)cdiv(B_pad, 4), 256, 0, as_stream(stream)>>>(a);
    return check_launch("mf_grad");
}

extern "C" int r4r_mf_grad(const int64_t *uid, const int64_t *iid, const float *y, const uint64_t *p,
                           const uint64_t *m, const uint64_t *v,
                           int64_t n_users, int64_t n_items, int D, float *pred, float *se, void *block, float *mult,
                           int64_t B, int64_t B_pad, float dropout_p, int training, uint64_t seed, uint64_t offset,
                           float inv_denom, void *ws, int sweep_period, int64_t sweep_base,
                           float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                           void *stream) {
    return mf_grad_impl(uid, iid, y, p, m, v, n_users, n_items, D, pred, se, block, mult, B, B_pad, dropout_p, training, seed,
                        offset, inv_denom, ws, sweep_period, sweep_base, lr, beta1, beta2, eps, weight_decay, adam_step, stream,
                        MfPush{});
}

extern "C" int r4r_mf_grad_push(const int64_t *uid, const int64_t *iid, const float *y, const uint64_t *p,
                                const uint64_t *m, const uint64_t *v,
                                int64_t n_users, int64_t n_items, int D, float *pred, float *se, float *mult,
                                int64_t B, int64_t B_pad, float dropout_p, int training, uint64_t seed, uint64_t offset,
                                float inv_denom, void *ws, int sweep_period, int64_t sweep_base,
                                float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                                const uint64_t *peer_dst, const uint64_t *peer_flags, uint32_t *arrive, int rank, int world,
                                uint32_t epoch, void *stream) {
    R4R_REQUIRE(peer_dst && peer_flags && arrive, "mf_grad_push: null pointer");
    MfPush push;
    push.peer_dst = peer_dst; push.peer_flags = peer_flags; push.arrive = arrive; push.rank = rank; push.world = world;
    push.epoch = epoch;
    return mf_grad_impl(uid, iid, y, p, m, v, n_users, n_items, D, pred, se, nullptr, mult, B, B_pad, dropout_p, training, seed,
                        offset, inv_denom, ws, sweep_period, sweep_base, lr, beta1, beta2, eps, weight_decay, adam_step, stream,
                        push);
}

namespace r4r {
struct MfWait {                        // r4r_mf_apply_peer: the flags every workgroup waits for before it reads the blocks
    const uint32_t *flags = nullptr;
    uint32_t *timed_out = nullptr;
    uint32_t epoch = 0;
    double timeout_s = 0.0;
};
}  // namespace r4r

static int mf_apply_impl(const void *blocks, int world, int64_t B_pad, const uint64_t *p, const uint64_t *m,
                         const uint64_t *v, int64_t n_users, int64_t n_items, int D, void *ws, size_t ws_bytes,
                         int sweep_period, int64_t sweep_base, int sweep_all,
                         const float *se, int64_t se_n, float *sse_accum,
                         float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                         void *stream, const MfWait &wait) {
    R4R_REQUIRE(blocks && p && m && v && ws, "mf_apply: null pointer");
    R4R_REQUIRE(!sse_accum || (se && se_n >= 0), "mf_apply: sse_accum needs se [se_n]");
    R4R_REQUIRE(sweep_period >= 1 && sweep_period <= MF_TB_MAX, "mf_apply: sweep_period %d outside 1..%d", sweep_period, MF_TB_MAX);
    R4R_REQUIRE(world >= 1 && B_pad >= 0 && n_users > 0 && n_items > 0, "mf_apply: bad sizes");
    const int64_t B = (int64_t)world * B_pad;
    R4R_REQUIRE(B <= MF_MAX_B_STEP, "mf_apply: %lld gathered entries > %d", (long long)B, MF_MAX_B_STEP);
    R4R_REQUIRE(D >= 0 && D <= MF_MAX_D, "mf_apply: latent_size %d outside 0..%d", D, MF_MAX_D);
    R4R_REQUIRE(adam_step >= 1 && adam_step < (1ll << 31), "mf_apply: bad adam_step");
    R4R_REQUIRE(sweep_base >= 0 && sweep_base < adam_step, "mf_apply: sweep_base outside 0..adam_step - 1");
    if (ws_bytes < r4r_mf_ws_bytes(B, D, n_users, n_items)) {
        set_error("mf_apply: workspace %zu < %zu bytes", ws_bytes, r4r_mf_ws_bytes(B, D, n_users, n_items));
        return R4R_ERR_WORKSPACE;
    }
    if (B == 0) return R4R_OK;
    hipStream_t st = as_stream(stream);
    const MfWs w = mf_carve(ws, B, D, n_users, n_items);
    MfRegister rg;
    rg.blocks = static_cast<const char *>(blocks); rg.k = mf_block(B_pad, D); rg.world = world; rg.D = D;
    rg.now = (int)adam_step; rg.B_pad = B_pad;
    rg.tag_u = w.tag_u; rg.tag_i = w.tag_i; rg.uid32 = w.uid32; rg.iid32 = w.iid32;
    rg.first_u = w.first_u; rg.first_i = w.first_i; rg.last_u = w.last_u; rg.last_i = w.last_i;
    rg.g = w.g; rg.gu = w.gu; rg.gi = w.gi;
    // the temporally blocked sweep (rows_device.h) over the gathered entries: 16-byte aligned tables
    const bool tb_on = D > 0 && (((p[0] | p[1] | m[0] | m[1] | v[0] | v[1]) & 15) == 0);
    if (tb_on) { rg.ctag_u = w.ctag_u; rg.ctag_i = w.ctag_i; }
    // Up to MF_SCAN_MAX_B gathered entries nothing is registered: the update launch's workgroups read the ids out of
    // the blocks and find their rows themselves (one dependent launch less per step; R4R_MF_DP_REGISTER=1 pins the
    // registered form for A/B runs and tests).  Same entries, same order, same sums: the same bits.
    static const bool pin_register = [] { const char *e = getenv("R4R_MF_DP_REGISTER"); return e && e[0] == '1'; }();
    const bool scan = B <= MF_SCAN_MAX_B && rg.k.bytes % 4 == 0 && (!pin_register || wait.flags);
    R4R_REQUIRE(scan || !wait.flags, "mf_apply_peer: %lld gathered entries > %d", (long long)B, MF_SCAN_MAX_B);
    if (!scan) mf_register_kernel<<<(unsigned)cdiv(B, 4), 256, 0, st>>>(rg);
    float *P[MF_SLOTS], *M[MF_SLOTS], *V[MF_SLOTS];
    for (int k = 0; k < MF_SLOTS; ++k) {
        P[k] = reinterpret_cast<float *>(p[k]); M[k] = reinterpret_cast<float *>(m[k]); V[k] = reinterpret_cast<float *>(v[k]);
        R4R_REQUIRE(k < (D > 0 ? 0 : 2) || (P[k] && M[k] && V[k]), "mf_apply: slot %d: null parameter / moment pointer", k);
    }
    MfSweep sw{};
    sw.p0 = P[0]; sw.p1 = P[1]; sw.p2 = P[2]; sw.p3 = P[3]; sw.p4 = P[4];
    sw.m0 = M[0]; sw.m1 = M[1]; sw.m2 = M[2]; sw.m3 = M[3]; sw.m4 = M[4];
    sw.v0 = V[0]; sw.v1 = V[1]; sw.v2 = V[2]; sw.v3 = V[3]; sw.v4 = V[4];
    const int64_t rows[4] = {n_users, n_items, n_users, n_items};
    const int width[4] = {D, D, 1, 1};
    MfTimeBlock tb{};
    if (tb_on) {
        tb.rlast_u = w.rlast_u; tb.rlast_i = w.rlast_i; tb.err = w.tb_err; tb.base = (int)sweep_base;
        tb.period = sweep_period; tb.flush = (sweep_all || sweep_period == 1) ? 1 : 0; tb.inc = 1;
        mf_time_block_scalars(tb, lr, beta1, beta2, eps, weight_decay, adam_step);
    }
    int64_t numel[4], begin[4], chunks = 0;
    for (int k = 0; k < 4; ++k) {
        numel[k] = rows[k] * width[k];
        begin[k] = chunks;
        chunks += k < 2 ? mf_sweep_wgs(numel[k], mf_chunk(k), tb) : cdiv(numel[k], mf_chunk(k));
    }
    sw.n0 = numel[0]; sw.n1 = numel[1]; sw.n2 = numel[2]; sw.n3 = numel[3];
    sw.cb1 = (int)begin[1]; sw.cb2 = (int)begin[2]; sw.cb3 = (int)begin[3];
    sw.cb_global = (int)chunks;
    chunks += 1;
    sw.cb_entries = (int)chunks;
    const bool light = mf_light(D, B);                      // (one rating per entry wave, the generic form)
    if (scan) sw.epw = mf_epw(B);
    else sw.epw = (!light && D > 0 && mf_wide(D, sw.p0, sw.m0, sw.v0) && mf_wide(D, sw.p1, sw.m1, sw.v1)) ? 256 / D : 1;
    sw.n_entry_wgs = (int)(2 * cdiv(B, 4 * sw.epw));
    chunks += sw.n_entry_wgs;
    R4R_REQUIRE(chunks < (1ll << 31), "mf_apply: too many workgroups");
    sw.uid = nullptr; sw.iid = nullptr;
    if (scan) {                                             // entries straight out of the gathered blocks
        const char *b0 = rg.blocks;
        sw.uid32 = reinterpret_cast<const int *>(b0 + rg.k.uid); sw.iid32 = reinterpret_cast<const int *>(b0 + rg.k.iid);
        sw.g = reinterpret_cast<const float *>(b0 + rg.k.g);
        sw.gu = reinterpret_cast<const float *>(b0 + rg.k.gu); sw.gi = reinterpret_cast<const float *>(b0 + rg.k.gi);
        sw.B_pad = B_pad; sw.blk_units = (int64_t)(rg.k.bytes / 4);
        if (tb_on) { sw.ctag_wu = w.ctag_u; sw.ctag_wi = w.ctag_i; }
        if (wait.flags) {
            sw.wait_flags = wait.flags; sw.timed_out = wait.timed_out; sw.wait_epoch = wait.epoch; sw.world = world;
            sw.max_ticks = (unsigned long long)(wait.timeout_s * 1e8);
        }
    } else {
        sw.first_u = w.first_u; sw.first_i = w.first_i; sw.last_u = w.last_u; sw.last_i = w.last_i;
        sw.uid32 = w.uid32; sw.iid32 = w.iid32;
        sw.gu = w.gu; sw.gi = w.gi; sw.g = w.g;
    }
    sw.se = sse_accum ? se : nullptr; sw.se_n = se_n; sw.sse_accum = sse_accum;   // this rank's share of the running metric rides on the global-bias workgroup
    sw.tag_u = w.tag_u; sw.tag_i = w.tag_i; sw.B = B; sw.D = D; sw.now = (int)adam_step;
    sw.s = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, adam_step, nullptr);
    if (tb_on) {
        sw.ctag_u = w.ctag_u; sw.ctag_i = w.ctag_i;
        sw.tb = tb;
        sw.nt = mf_sweep_nt(sw.n0 + sw.n1);
    }
    if (scan) {
        const size_t lds = (size_t)B * sizeof(int);         // the entry waves' ids (<= 8 KB)
        if (light) mf_adam_kernel<4, 1, false, true><<<(unsigned)chunks, MF_THREADS, lds, st>>>(sw);
        else mf_adam_kernel<4, 4, true, true><<<(unsigned)chunks, MF_THREADS, lds, st>>>(sw);
    }
    else if (B > 2048) mf_adam_kernel<8><<<(unsigned)chunks, MF_THREADS, 0, st>>>(sw);
    else if (light) mf_adam_kernel<4, 1, false><<<(unsigned)chunks, MF_THREADS, 0, st>>>(sw);
    else mf_adam_kernel<4><<<(unsigned)chunks, MF_THREADS, 0, st>>>(sw);
    return check_launch("mf_apply");
}

extern "C" int r4r_mf_apply(const void *blocks, int world, int64_t B_pad, const uint64_t *p, const uint64_t *m,
                            const uint64_t *v, int64_t n_users, int64_t n_items, int D, void *ws, size_t ws_bytes,
                            int sweep_period, int64_t sweep_base, int sweep_all,
                            const float *se, int64_t se_n, float *sse_accum,
                            float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                            void *stream) {
    return mf_apply_impl(blocks, world, B_pad, p, m, v, n_users, n_items, D, ws, ws_bytes, sweep_period, sweep_base, sweep_all,
                         se, se_n, sse_accum, lr, beta1, beta2, eps, weight_decay, adam_step, stream, MfWait{});
}

extern "C" int r4r_mf_apply_peer(const void *blocks, int world, int64_t B_pad, const uint64_t *p, const uint64_t *m,
                                 const uint64_t *v, int64_t n_users, int64_t n_items, int D, void *ws, size_t ws_bytes,
                                 int sweep_period, int64_t sweep_base, int sweep_all,
                                 const float *se, int64_t se_n, float *sse_accum,
                                 float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                                 const uint32_t *wait_flags, uint32_t epoch, uint32_t *timed_out, double timeout_s,
                                 void *stream) {
    R4R_REQUIRE(wait_flags && timed_out, "mf_apply_peer: null pointer");
    R4R_REQUIRE(world >= 1 && world <= PEER_MAX_WORLD, "mf_apply_peer: %d ranks (<= %d)", world, PEER_MAX_WORLD);
    R4R_REQUIRE(timeout_s > 0 && timeout_s <= 60, "mf_apply_peer: timeout %.3f s outside (0, 60]", timeout_s);
    MfWait wait;
    wait.flags = wait_flags; wait.timed_out = timed_out; wait.epoch = epoch; wait.timeout_s = timeout_s;
    return mf_apply_impl(blocks, world, B_pad, p, m, v, n_users, n_items, D, ws, ws_bytes, sweep_period, sweep_base, sweep_all,
                         se, se_n, sse_accum, lr, beta1, beta2, eps, weight_decay, adam_step, stream, wait);
}
