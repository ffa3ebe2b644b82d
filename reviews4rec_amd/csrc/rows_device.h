// Launchers shared between the fused model steps (defined in mf_engine.hip).
#pragma once
#include "adam_device.h"

namespace r4r {

// elements of a table per sweep workgroup (and per chunk tag).  (4096 until the sweep went on a schedule: only due chunks
// are launched now, and four times the waves share the 8 updates a visit applies -- profiles/r04d_chunk_ab.txt)
constexpr int MF_CHUNK = 1024;

int mf_bias_rows_launch(float *ub, float *ub_m, float *ub_v, float *ib, float *ib_m, float *ib_v,
                        int64_t n_users, int64_t n_items, const int64_t *uid, const int64_t *iid, const float *g,
                        const int *tag_u, const int *tag_i, int64_t B, int now, const AdamScalars &sc, hipStream_t st);

// The sweep, temporally blocked on a schedule.  Dense Adam moves every element every step (weight decay, decaying
// moments), but an element no rating names goes through the SAME gradient-zero update whether it is applied now or
// together with the next ones: per element the update reads nothing but the element's own (p, m, v) and the step's two
// bias corrections.  So chunk c of MF_CHUNK table elements (run r = c / MF_TB_RUN) is visited only at the steps s with
// (r % period + r / period + s) % period == 0 -- one run of every block of `period` consecutive runs per step, on a
// diagonal, so a launch covers exactly the due chunks and their addresses are not a power-of-two stride apart -- and
// then takes all its pending updates at once, in order, each with its own step's scalars: the fp32 operations per
// element are those of the plain sweep, the traffic a period-th.  Nothing is announced and no per-chunk state is
// kept: the step of a chunk's LAST visit is a function of (c, now) (tb_prev_visit), an element is current through
// max(base, last visit of its chunk, rlast[its row]), and whoever needs it newer applies the missing gradient-zero
// updates on the way: the forward kernels in registers (nothing written), the entry waves before their gradient
// update (they then set rlast[row] = now), the sweep at its visit.  `base`: a step through which EVERYTHING is current
// (the last flush or all-chunks step); the caller keeps (base, period) and changes the period only at an all-chunks
// launch (`flush` = 1), which uses the OLD period.  (Until round 4 the sweep kept a pending count per chunk and the
// caller ANNOUNCED the next batch so that its rows could be brought up to date a step ahead: half of the traffic
// at cfg2 was chunks some rating named, and a kept promise was part of the contract.)
constexpr int MF_TB_MAX = 8;  // pending updates an element may carry (period <= this)
struct MfTimeBlock {
    int *err;                          // *err = 2 if more than MF_TB_MAX updates were ever pending somewhere
    int period, flush;                 // flush: visit every chunk (applying what is pending under `period`)
    int inc;                           // 1: this launch is step `now` itself; 0: flush only, `now` = the last completed step
    float lr_bc1[MF_TB_MAX], isb2[MF_TB_MAX];   // AdamScalars::lr_over_bc1 / inv_sqrt_bc2 of steps now - 7 .. now
    int *rlast_u = nullptr, *rlast_i = nullptr;   // [rows]: the last step an entry wave updated the row (0: never); NULL: off
    int base = 0;
};

// The schedule works on RUNS of MF_TB_RUN consecutive chunks (one phase per run): what a launch visits is then runs
// of MF_TB_RUN * MF_CHUNK contiguous elements per array -- whole DRAM pages instead of 4 KB pieces
// (measured, profiles/r04d_run_ab.txt: runs of 1, 4, 16, 64 chunks within 1 % of each other at cfg2 and cfg5 -- the sweep is not bound by DRAM page locality; the default stays 1).
constexpr int MF_TB_RUN = 1;
// the last step <= t at which the schedule visits chunk c (may lie before `base`: the caller takes the max)
__host__ __device__ inline int tb_prev_visit(int64_t c, int t, int period) {
    const unsigned cu = (unsigned)(c / MF_TB_RUN), pu = (unsigned)period;   // (chunk numbers fit 31 bits: the launch is one workgroup per chunk)
    const unsigned ph = (cu % pu + cu / pu) % pu;
    return t - (int)((ph + (unsigned)t) % pu);
}
// the chunk that workgroup q of a table's sweep visits at step `now`: chunk q % MF_TB_RUN of the one due run of the
// block of `period` runs number q / MF_TB_RUN
__host__ __device__ inline int64_t tb_due_chunk(int64_t q, int now, int period) {
    const unsigned pu = (unsigned)period;
    const int64_t blk = q / MF_TB_RUN;
    const int64_t run = blk * period + (int)((pu - ((unsigned)blk + (unsigned)now) % pu) % pu);
    return run * MF_TB_RUN + q % MF_TB_RUN;
}
// workgroups of a table's sweep when only the due chunks are visited
__host__ __device__ inline int64_t tb_due_wgs(int64_t nchunks, int period) {
    const int64_t runs = (nchunks + MF_TB_RUN - 1) / MF_TB_RUN;
    return ((runs + period - 1) / period) * MF_TB_RUN;
}
// the gradient-zero updates of steps cur + 1 .. upto on one element (upto <= now, upto - cur <= MF_TB_MAX; per lane)
__device__ __forceinline__ void tb_catch_up(float &P, float &M, float &V, int cur, int upto, int now, const AdamScalars &sc0,
                                            const MfTimeBlock &tb) {
#pragma unroll
    for (int j = 0; j < MF_TB_MAX; ++j) {
        const int s = now - (MF_TB_MAX - 1 - j);
        if (s > cur && s <= upto) {
            AdamScalars sc = sc0;
            sc.lr_over_bc1 = tb.lr_bc1[j];
            sc.inv_sqrt_bc2 = tb.isb2[j];
            adam_elem_fast(P, 0.f, M, V, sc);
        }
    }
}
// the same on N float4 of (p, m, v), float4 u current through cur[u] (per lane): the step loop is the OUTER one, so
// a step's scalars are fetched once for everything a lane holds (they are kernel arguments; a loop per element made
// hipcc re-load them from the argument segment in front of every update)
typedef float tb_f32x4 __attribute__((ext_vector_type(4)));
template <int N>
__device__ __forceinline__ void tb_catch_up_v(tb_f32x4 (&P)[N], tb_f32x4 (&M)[N], tb_f32x4 (&V)[N], const int (&cur)[N], int upto,
                                              int now, const AdamScalars &sc0, const MfTimeBlock &tb) {
    // The sixteen per-step scalars live in VECTOR registers here (the same value in every lane): as kernel arguments they
    // are scalar registers, the sweep kernels have none to spare (128 scalar spills), and hipcc kept them spilled in the
    // lanes of a vector register -- 136 v_readlane restores among the 208 arithmetic instructions of this loop, in a
    // launch its arithmetic bounds.
    float lrv[MF_TB_MAX], isv[MF_TB_MAX];
#pragma unroll
    for (int j = 0; j < MF_TB_MAX; ++j) {
        lrv[j] = tb.lr_bc1[j]; isv[j] = tb.isb2[j];
        asm volatile("" : "+v"(lrv[j]), "+v"(isv[j]));
    }
    // ... and the six step-invariant ones are pinned in scalar registers: read through the argument pointer (common.h:
    // kernel_args) they are loads hipcc is free to SINK into the blocks that use them -- an s_load_dwordx4 and a full
    // scalar wait in front of each of the eight predicated updates.
    AdamScalars scp = sc0;
    asm volatile("" : "+s"(scp.beta1), "+s"(scp.beta2), "+s"(scp.eps), "+s"(scp.wd), "+s"(scp.omb1), "+s"(scp.omb2));
#pragma unroll
    for (int j = 0; j < MF_TB_MAX; ++j) {
        const int s = now - (MF_TB_MAX - 1 - j);
        if (s > upto) continue;                             // uniform
        AdamScalars sc = scp;
        sc.lr_over_bc1 = lrv[j];
        sc.inv_sqrt_bc2 = isv[j];
#pragma unroll
        for (int u = 0; u < N; ++u)
            if (s > cur[u]) {                               // (two elements per packed instruction: adam_pair_fast)
                const adam_f32x2 z = {0.f, 0.f};
                adam_f32x2 pa = {P[u][0], P[u][1]}, ma = {M[u][0], M[u][1]}, va = {V[u][0], V[u][1]};
                adam_f32x2 pb = {P[u][2], P[u][3]}, mb = {M[u][2], M[u][3]}, vb = {V[u][2], V[u][3]};
                adam_pair_fast(pa, z, ma, va, sc);
                adam_pair_fast(pb, z, mb, vb, sc);
                P[u] = (tb_f32x4){pa.x, pa.y, pb.x, pb.y}; M[u] = (tb_f32x4){ma.x, ma.y, mb.x, mb.y};
                V[u] = (tb_f32x4){va.x, va.y, vb.x, vb.y};
            }
    }
}
// through which step is element e of a table (row r) current before step `now`'s own update?
__device__ __forceinline__ int tb_current(const MfTimeBlock &tb, const int *rlast, int64_t e, int64_t r, int now) {
    int cur = tb_prev_visit(e / MF_CHUNK, now - 1, tb.period);
    if (cur < tb.base) cur = tb.base;
    const int rl = rlast[r];
    return cur < rl ? rl : cur;
}
void mf_time_block_scalars(MfTimeBlock &tb, float lr, double beta1, double beta2, float eps, float weight_decay, int64_t now);

int mf_table_rows_launch(float *ut, float *ut_m, float *ut_v, float *it, float *it_m, float *it_v,
                         int64_t n_users, int64_t n_items, int D, const int64_t *uid, const int64_t *iid,
                         const float *gu, const float *gi, const int *tag_u, const int *tag_i,
                         const int *ctag_u, const int *ctag_i,   // [ceil(rows * D / MF_CHUNK)] chunk tags, or NULL
                         int64_t B, int now, const AdamScalars &sc, hipStream_t st,
                         const MfTimeBlock *tb = nullptr);       // (needs the chunk tags and 16-byte aligned tables)

// Data parallel: a rank's compact entries as one packed block (bytes): uid32 [B_pad] | iid32 [B_pad] | g [B_pad] |
// gu [B_pad, D] | gi [B_pad, D], every field 256-byte aligned; entries past the rank's own count carry id -1.
struct MfBlock { size_t uid, iid, g, gu, gi, bytes; };
inline MfBlock mf_block(int64_t B_pad, int D) {
    MfBlock k;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 255) & ~(size_t)255; return r; };
    k.uid = take((size_t)B_pad * 4); k.iid = take((size_t)B_pad * 4); k.g = take((size_t)B_pad * 4);
    k.gu = take((size_t)B_pad * D * 4); k.gi = take((size_t)B_pad * D * 4);
    k.bytes = o;
    return k;
}
int mf_table_rows_blocks_launch(float *ut, float *ut_m, float *ut_v, float *it, float *it_m, float *it_v,
                                int64_t n_users, int64_t n_items, int D, const void *blocks, int world, int64_t B_pad,
                                int *ctag_u, int *ctag_i, int now, const AdamScalars &sc, hipStream_t st,
                                const MfTimeBlock *tb = nullptr);

int mf_table_bias_rows_launch(float *ut, float *ut_m, float *ut_v, float *it, float *it_m, float *it_v,
                              float *ub, float *ub_m, float *ub_v, float *ib, float *ib_m, float *ib_v,
                              int64_t n_users, int64_t n_items, int D, const int64_t *uid, const int64_t *iid,
                              const float *gu, const float *gi, const float *g, const int *tag_u, const int *tag_i,
                              int64_t B, int now, const AdamScalars &sc, hipStream_t st,
                              const int *ctag_u = nullptr, const int *ctag_i = nullptr, const MfTimeBlock *tb = nullptr);

}  // namespace r4r
