// Launchers shared between the fused model steps (defined in mf_engine.hip).
#pragma once
#include "adam_device.h"

namespace r4r {

int mf_bias_rows_launch(float *ub, float *ub_m, float *ub_v, float *ib, float *ib_m, float *ib_v,
                        int64_t n_users, int64_t n_items, const int64_t *uid, const int64_t *iid, const float *g,
                        const int *tag_u, const int *tag_i, int64_t B, int now, const AdamScalars &sc, hipStream_t st);

}  // namespace r4r
