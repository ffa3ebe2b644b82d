// Launchers shared between the fused model steps (defined in mf_engine.hip).
#pragma once
#include "adam_device.h"

namespace r4r {

#ifndef R4R_MF_CHUNK
#define R4R_MF_CHUNK 4096                  // (8192 until late round 3; A/B in profiles/r03f_chunk_ab.txt: cfg2 +12 % at B = 128, +6 % at 8,192; cfg5 within 1 %; 2048: cfg5 -10 %)
#endif
constexpr int MF_CHUNK = R4R_MF_CHUNK;         // elements of a table per sweep workgroup (and per chunk tag)

int mf_bias_rows_launch(float *ub, float *ub_m, float *ub_v, float *ib, float *ib_m, float *ib_v,
                        int64_t n_users, int64_t n_items, const int64_t *uid, const int64_t *iid, const float *g,
                        const int *tag_u, const int *tag_i, int64_t B, int now, const AdamScalars &sc, hipStream_t st);

// The sweep, temporally blocked.  Dense Adam moves every element every step (weight decay, decaying moments), but
// an element no rating names goes through the SAME gradient-zero update whether it is applied now or together with
// the next ones: per element the update reads nothing but the element's own (p, m, v) and the step's two bias
// corrections.  So a chunk of MF_CHUNK elements that neither this batch nor the ANNOUNCED next batch names is
// visited every `period`-th step only and then takes all its pending updates at once -- in order, each with its own
// step's scalars: the fp32 operations per element are those of the plain sweep, the traffic a period-th.  Chunks this
// batch names are visited now (they were brought up to date a step ago, when this batch was the announced one);
// chunks the next batch names are brought up to date now.  `lag` = pending updates per chunk (0 = current; zeroed
// workspace = nothing pending).  A caller that announces nothing flushes (period 1 semantics for this step).
constexpr int MF_TB_MAX = 8;           // pending updates a chunk may carry (period <= this)
struct MfTimeBlock {
    int *lag_u, *lag_i;                // [chunks of the table]
    const int *ntag_u, *ntag_i;        // [chunks]: == now where the announced next batch names a row (ignored when flushing)
    int *err;                          // *err = 1 if a chunk this batch names was NOT current (a broken announcement)
    int period, flush;
    int inc;                           // 1: this launch is step `now` itself; 0: flush only, `now` = the last completed step
    float lr_bc1[MF_TB_MAX], isb2[MF_TB_MAX];   // AdamScalars::lr_over_bc1 / inv_sqrt_bc2 of steps now - 7 .. now
};
void mf_time_block_scalars(MfTimeBlock &tb, float lr, double beta1, double beta2, float eps, float weight_decay, int64_t now);

int mf_table_rows_launch(float *ut, float *ut_m, float *ut_v, float *it, float *it_m, float *it_v,
                         int64_t n_users, int64_t n_items, int D, const int64_t *uid, const int64_t *iid,
                         const float *gu, const float *gi, const int *tag_u, const int *tag_i,
                         const int *ctag_u, const int *ctag_i,   // [ceil(rows * D / MF_CHUNK)] chunk tags, or NULL
                         int64_t B, int now, const AdamScalars &sc, hipStream_t st,
                         const MfTimeBlock *tb = nullptr);       // (needs the chunk tags and 16-byte aligned tables)

int mf_table_bias_rows_launch(float *ut, float *ut_m, float *ut_v, float *it, float *it_m, float *it_v,
                              float *ub, float *ub_m, float *ub_v, float *ib, float *ib_m, float *ib_v,
                              int64_t n_users, int64_t n_items, int D, const int64_t *uid, const int64_t *iid,
                              const float *gu, const float *gi, const float *g, const int *tag_u, const int *tag_i,
                              int64_t B, int now, const AdamScalars &sc, hipStream_t st,
                              const int *ctag_u = nullptr, const int *ctag_i = nullptr, const MfTimeBlock *tb = nullptr);

}  // namespace r4r
