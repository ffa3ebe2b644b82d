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

// Two forms of the same update.
//   adam_elem       -- IEEE-exact sqrtf() and `/` (hipcc's expansions: 27 of the update's 35 vector instructions).
//                      The DEFAULT: the flat dense-parameter buffers (adam.hip, the fused reduce launches) are
//                      latency- or HBM-bound and pay nothing for it, and their update is torch.optim.Adam's to the bit.
//   adam_elem_fast  -- the hardware's v_sqrt_f32 / v_rcp_f32 (1 ulp each; m * rcp(denom) for m / denom; a denormal
//                      second moment reads as zero: sqrt(v) < 1e-19 against eps = 1e-8 either way).  Only for the
//                      ID-table sweeps (mf_engine.hip, rows_device.h / step_device.h, idnet_engine.hip): temporally
//                      blocked, they apply several updates per byte moved and were bound by this arithmetic (DESIGN.md
//                      4.5); every path that can touch a table row uses this form, so blocked and dense sweeps agree to
//                      the bit.  R4R_ADAM_IEEE=1 at build time makes both forms exact (A/B runs).
#ifndef R4R_ADAM_IEEE
#define R4R_ADAM_IEEE 0
#endif
template <bool FAST>
__device__ __forceinline__ void adam_elem_form(float &p, float g, float &m, float &v, const AdamScalars &s) {
    g = fmaf(s.wd, p, g);
    m = fmaf(s.beta1, m, s.omb1 * g);
    v = fmaf(s.beta2, v, s.omb2 * g * g);
    if constexpr (FAST && !R4R_ADAM_IEEE) {
        const float denom = __builtin_amdgcn_sqrtf(v) * s.inv_sqrt_bc2 + s.eps;
        p -= s.lr_over_bc1 * (m * __builtin_amdgcn_rcpf(denom));
    } else {
        const float denom = sqrtf(v) * s.inv_sqrt_bc2 + s.eps;
        p -= s.lr_over_bc1 * (m / denom);
    }
}
__device__ __forceinline__ void adam_elem(float &p, float g, float &m, float &v, const AdamScalars &s) {
    adam_elem_form<false>(p, g, m, v, s);
}
__device__ __forceinline__ void adam_elem_fast(float &p, float g, float &m, float &v, const AdamScalars &s) {
    adam_elem_form<true>(p, g, m, v, s);
}

// adam_elem_fast on TWO elements at once: the same IEEE operations per element (fma, mul, v_sqrt_f32, v_rcp_f32 -- the
// results are bit-identical to two adam_elem_fast calls), written on 2-vectors so that hipcc emits the packed fp32
// instructions (v_pk_fma_f32 / v_pk_mul_f32: nine of them + four transcendentals for two elements instead of 2 x (8 + 2)).
// The blocked sweeps apply several updates per byte moved and are bound by this arithmetic.
typedef float adam_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void adam_pair_fast(adam_f32x2 &p, adam_f32x2 g, adam_f32x2 &m, adam_f32x2 &v, const AdamScalars &s) {
#if R4R_ADAM_IEEE
    float p0 = p.x, p1 = p.y, m0 = m.x, m1 = m.y, v0 = v.x, v1 = v.y;
    adam_elem(p0, g.x, m0, v0, s); adam_elem(p1, g.y, m1, v1, s);
    p = (adam_f32x2){p0, p1}; m = (adam_f32x2){m0, m1}; v = (adam_f32x2){v0, v1};
#else
    g = __builtin_elementwise_fma((adam_f32x2){s.wd, s.wd}, p, g);
    m = __builtin_elementwise_fma((adam_f32x2){s.beta1, s.beta1}, m, s.omb1 * g);
    v = __builtin_elementwise_fma((adam_f32x2){s.beta2, s.beta2}, v, (s.omb2 * g) * g);
    const adam_f32x2 sq = {__builtin_amdgcn_sqrtf(v.x), __builtin_amdgcn_sqrtf(v.y)};
    const adam_f32x2 den = sq * s.inv_sqrt_bc2 + s.eps;
    const adam_f32x2 rc = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    p -= s.lr_over_bc1 * (m * rc);
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
