// Adam update of one element + the per-step scalars, shared by the stand-alone multi-tensor
// kernel (adam.hip) and the fused DeepCoNN step (engine.hip), so both produce the same bits.
// Reference behaviour restated: torch.optim.Adam(lr, weight_decay).step() as used at
// main.py:94-96,60 -- betas (0.9, 0.999), eps 1e-8, L2 weight decay added to the gradient,
// bias-corrected, NOT amsgrad.
#pragma once
#include <math.h>
#include <stdint.h>

#include "common.h"

namespace r4r {

struct AdamScalars {
    float lr_over_bc1;      // lr / (1 - beta1^t)
    float inv_sqrt_bc2;     // 1 / sqrt(1 - beta2^t)
    float lr;
    double b1d, b2d;        // betas in double, for the device-side bias correction
    const int64_t *step_dev;   // optional: completed-step counter in device memory (graph replay)
    float beta1, beta2, eps, wd;
    float omb1, omb2;       // 1 - beta, rounded from double like torch's `value=1 - beta2`
};

// The square root and the division are the hardware's (v_sqrt_f32, v_rcp_f32: 1 ulp each) instead of the IEEE-exact
// expansions hipcc emits for sqrtf() and `/` (27 vector instructions of the update's 35): the update term carries a
// relative error of ~2e-7 either way -- the reference's own CPU arithmetic is no closer to the real number -- and the
// temporally blocked sweeps, which apply several updates per byte moved, stop being bound by this arithmetic (cfg5's
// sweep 92 -> see DESIGN.md 4.5).  R4R_ADAM_IEEE=1 at build time restores the expansions (A/B runs).  A denormal
// second moment reads as zero here: sqrt(v) < 1e-19 against eps = 1e-8 either way.
#ifndef R4R_ADAM_IEEE
#define R4R_ADAM_IEEE 0
#endif
__device__ __forceinline__ void adam_elem(float &p, float g, float &m, float &v, const AdamScalars &s) {
    g = fmaf(s.wd, p, g);
    m = fmaf(s.beta1, m, s.omb1 * g);
    v = fmaf(s.beta2, v, s.omb2 * g * g);
#if R4R_ADAM_IEEE
    const float denom = sqrtf(v) * s.inv_sqrt_bc2 + s.eps;
    p -= s.lr_over_bc1 * (m / denom);
#else
    const float denom = __builtin_amdgcn_sqrtf(v) * s.inv_sqrt_bc2 + s.eps;
    p -= s.lr_over_bc1 * (m * __builtin_amdgcn_rcpf(denom));
#endif
}

// step >= 1: the 1-based count of this update
static inline AdamScalars adam_make_scalars(float lr, double beta1, double beta2, float eps, float weight_decay,
                                            int64_t step, const int64_t *step_dev) {
    AdamScalars s;
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    s.lr_over_bc1 = (float)((double)lr / bc1);
    s.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    s.beta1 = (float)beta1; s.beta2 = (float)beta2; s.eps = eps; s.wd = weight_decay;
    s.omb1 = (float)(1.0 - beta1); s.omb2 = (float)(1.0 - beta2);
    s.lr = lr; s.b1d = beta1; s.b2d = beta2; s.step_dev = step_dev;
    return s;
}

}  // namespace r4r
