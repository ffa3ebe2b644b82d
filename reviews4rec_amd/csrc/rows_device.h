// Launchers shared between the fused model steps (defined in mf_engine.hip).
#pragma once
#include "adam_device.h"

namespace r4r {

constexpr int MF_CHUNK = 8192;         // elements of a table per sweep workgroup (and per chunk tag)

int mf_bias_rows_launch(float *ub, float *ub_m, float *ub_v, float *ib, float *ib_m, float *ib_v,
                        int64_t n_users, int64_t n_items, const int64_t *uid, const int64_t *iid, const float *g,
                        const int *tag_u, const int *tag_i, int64_t B, int now, const AdamScalars &sc, hipStream_t st);

int mf_table_rows_launch(float *ut, float *ut_m, float *ut_v, float *it, float *it_m, float *it_v,
                         int64_t n_users, int64_t n_items, int D, const int64_t *uid, const int64_t *iid,
                         const float *gu, const float *gi, const int *tag_u, const int *tag_i,
                         const int *ctag_u, const int *ctag_i,   // [ceil(rows * D / MF_CHUNK)] chunk tags, or NULL
                         int64_t B, int now, const AdamScalars &sc, hipStream_t st);

int mf_table_bias_rows_launch(float *ut, float *ut_m, float *ut_v, float *it, float *it_m, float *it_v,
                              float *ub, float *ub_m, float *ub_v, float *ib, float *ib_m, float *ib_v,
                              int64_t n_users, int64_t n_items, int D, const int64_t *uid, const int64_t *iid,
                              const float *gu, const float *gi, const float *g, const int *tag_u, const int *tag_i,
                              int64_t B, int now, const AdamScalars &sc, hipStream_t st);

}  // namespace r4r
