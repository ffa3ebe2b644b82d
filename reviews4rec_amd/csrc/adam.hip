// Dense fused Adam for gfx950: one streaming pass over every element of every
// listed tensor (read p, g, m, v; write p, m, v = 28 B/element, 24 B when the
// gradient is known to be zero).  HBM-bound by construction: float4 accesses,
// one 8192-element chunk per workgroup so a 16.6 M-parameter table is ~2000
// workgroups.
//
// Reference behaviour restated: torch.optim.Adam(lr, weight_decay).step() as
// used at main.py:94-96,60 -- betas (0.9, 0.999), eps 1e-8, L2 weight decay
// added to the gradient, bias-corrected, NOT amsgrad.  Every element moves
// every step, including embedding rows no example touched.
#include "adam_device.h"

namespace r4r {

constexpr int ADAM_CHUNK = 8192;      // elements per workgroup for large tensor lists
constexpr int ADAM_CHUNK_SMALL = 1024; // ... when the whole list is small (latency-bound): more, shorter workgroups
constexpr int ADAM_THREADS = 256;

constexpr int ADAM_BATCH = 16;     // tensors described by value in one launch's kernel arguments

struct AdamBatch {
    float *p[ADAM_BATCH];
    const float *g[ADAM_BATCH];
    float *m[ADAM_BATCH];
    float *v[ADAM_BATCH];
    int64_t numel[ADAM_BATCH];
    int chunk_begin[ADAM_BATCH + 1];   // prefix sum of per-tensor chunk counts
    int ntensor;
    int chunk;                         // elements per workgroup
};

__global__ __launch_bounds__(ADAM_THREADS) void adam_multi_kernel(AdamBatch tb, AdamScalars s) {
    if (s.step_dev) {
        // t = completed steps + 1, bias corrections computed here so that a captured graph
        // replays with the right step (kernel arguments are frozen at capture time)
        const double t_ = (double)(s.step_dev[0] + 1);
        s.lr_over_bc1 = (float)((double)s.lr / (1.0 - pow(s.b1d, t_)));
        s.inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow(s.b2d, t_)));
    }
    int t = 0;
#pragma unroll
    for (int k = 1; k < ADAM_BATCH; ++k)
        if (k < tb.ntensor && (int)blockIdx.x >= tb.chunk_begin[k]) t = k;
    const int64_t start = (int64_t)((int)blockIdx.x - tb.chunk_begin[t]) * tb.chunk;
    float *p = tb.p[t] + start;
    const float *g = tb.g[t] ? tb.g[t] + start : nullptr;
    float *m = tb.m[t] + start;
    float *v = tb.v[t] + start;
    const int64_t *numel = tb.numel;
    int64_t cnt = numel[t] - start;
    if (cnt > tb.chunk) cnt = tb.chunk;
    const bool aligned = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) |
                           reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(g)) & 15) == 0;
    if (aligned) {
        const int64_t nvec = cnt >> 2;
        for (int64_t i = threadIdx.x; i < nvec; i += ADAM_THREADS) {
            float4 P = reinterpret_cast<float4 *>(p)[i];
            float4 M = reinterpret_cast<float4 *>(m)[i];
            float4 V = reinterpret_cast<float4 *>(v)[i];
            float4 G = g ? reinterpret_cast<const float4 *>(g)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            adam_elem(P.x, G.x, M.x, V.x, s);
            adam_elem(P.y, G.y, M.y, V.y, s);
            adam_elem(P.z, G.z, M.z, V.z, s);
            adam_elem(P.w, G.w, M.w, V.w, s);
            reinterpret_cast<float4 *>(p)[i] = P;
            reinterpret_cast<float4 *>(m)[i] = M;
            reinterpret_cast<float4 *>(v)[i] = V;
        }
        for (int64_t i = (nvec << 2) + threadIdx.x; i < cnt; i += ADAM_THREADS) {
            float P = p[i], M = m[i], V = v[i];
            adam_elem(P, g ? g[i] : 0.f, M, V, s);
            p[i] = P; m[i] = M; v[i] = V;
        }
    } else {
        for (int64_t i = threadIdx.x; i < cnt; i += ADAM_THREADS) {
            float P = p[i], M = m[i], V = v[i];
            adam_elem(P, g ? g[i] : 0.f, M, V, s);
            p[i] = P; m[i] = M; v[i] = V;
        }
    }
}

// Data-parallel form: the gradient is the sum of `world` per-rank buffers that an all_gather
// laid out back to back; they are added in rank order (every rank gets the same bits) and the
// element is updated at once -- the exchange needs one collective phase instead of the
// reduce-scatter + all-gather of an all-reduce, and no separate summation launch.
__global__ __launch_bounds__(ADAM_THREADS) void adam_gathered_kernel(float *__restrict__ p, const float *__restrict__ gathered,
                                                                     int world, float *__restrict__ g_sum,
                                                                     float *__restrict__ m, float *__restrict__ v,
                                                                     int64_t numel, AdamScalars s,
                                                                     const uint32_t *__restrict__ abort_word) {
    // a peer exchange whose wait timed out left stale or partial slots behind (peer.hip sets the word and ends):
    // summing them would give every rank a different gradient -- the update is skipped and the host raises
    if (abort_word && *abort_word != 0) return;
    const int64_t nvec = numel >> 2;
    const int64_t i = (int64_t)blockIdx.x * ADAM_THREADS + threadIdx.x;
    if (i < nvec) {
        float4 G = reinterpret_cast<const float4 *>(gathered)[i];
        for (int w = 1; w < world; ++w) {
            const float4 t = reinterpret_cast<const float4 *>(gathered + (size_t)w * numel)[i];
            G.x += t.x; G.y += t.y; G.z += t.z; G.w += t.w;
        }
        float4 P = reinterpret_cast<float4 *>(p)[i];
        float4 M = reinterpret_cast<float4 *>(m)[i];
        float4 V = reinterpret_cast<float4 *>(v)[i];
        adam_elem(P.x, G.x, M.x, V.x, s);
        adam_elem(P.y, G.y, M.y, V.y, s);
        adam_elem(P.z, G.z, M.z, V.z, s);
        adam_elem(P.w, G.w, M.w, V.w, s);
        reinterpret_cast<float4 *>(p)[i] = P;
        reinterpret_cast<float4 *>(m)[i] = M;
        reinterpret_cast<float4 *>(v)[i] = V;
        if (g_sum) reinterpret_cast<float4 *>(g_sum)[i] = G;
    }
}

}  // namespace r4r

using namespace r4r;

extern "C" int r4r_adam_gathered_guarded(float *p, const float *gathered, int world, float *g_sum, float *m, float *v,
                                         int64_t numel, float lr, double beta1, double beta2, float eps,
                                         float weight_decay, int64_t step, const uint32_t *abort_word, void *stream);

extern "C" int r4r_adam_gathered(float *p, const float *gathered, int world, float *g_sum, float *m, float *v,
                                 int64_t numel, float lr, double beta1, double beta2, float eps,
                                 float weight_decay, int64_t step, void *stream) {
    return r4r_adam_gathered_guarded(p, gathered, world, g_sum, m, v, numel, lr, beta1, beta2, eps, weight_decay, step,
                                     nullptr, stream);
}

extern "C" int r4r_adam_gathered_guarded(float *p, const float *gathered, int world, float *g_sum, float *m, float *v,
                                         int64_t numel, float lr, double beta1, double beta2, float eps,
                                         float weight_decay, int64_t step, const uint32_t *abort_word, void *stream) {
    R4R_REQUIRE(p && gathered && m && v, "adam_gathered: null pointer");
    R4R_REQUIRE(world >= 1 && numel >= 0 && step >= 1, "adam_gathered: bad world / numel / step");
    R4R_REQUIRE(numel % 4 == 0, "adam_gathered: numel %lld must be a multiple of 4 (flat buffers are)", (long long)numel);
    R4R_REQUIRE(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(gathered) | reinterpret_cast<uintptr_t>(m) |
                  reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(g_sum)) & 15) == 0,
                "adam_gathered: buffers must be 16-byte aligned");
    if (numel == 0) return R4R_OK;
    const AdamScalars s = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, step, nullptr);
    ScopedTiming tm(R4R_TIMING_ADAM, as_stream(stream));
    adam_gathered_kernel<<<(unsigned)cdiv(numel / 4, ADAM_THREADS), ADAM_THREADS, 0, as_stream(stream)>>>(
        p, gathered, world, g_sum, m, v, numel, s, abort_word);
    return check_launch("adam_gathered");
}

extern "C" int r4r_adam_chunk_elems(void) { return ADAM_CHUNK; }

extern "C" int r4r_adam_multi(int ntensor, const uint64_t *p, const uint64_t *g, const uint64_t *m,
                              const uint64_t *v, const int64_t *numel,
                              float lr, double beta1, double beta2, float eps,
                              float weight_decay, int64_t step, const int64_t *step_dev, void *stream) {
    R4R_REQUIRE(ntensor >= 0 && (ntensor == 0 || (p && g && m && v && numel)), "adam_multi: null pointer");
    R4R_REQUIRE(step_dev || step >= 1, "adam_multi: step must be >= 1");
    if (step_dev) step = 1;            // placeholder: the kernel derives the corrections from *step_dev
    const AdamScalars s = adam_make_scalars(lr, beta1, beta2, eps, weight_decay, step, step_dev);
    int64_t total = 0;
    for (int k = 0; k < ntensor; ++k) total += numel[k];
    const int chunk = total >= (4ll << 20) ? ADAM_CHUNK : ADAM_CHUNK_SMALL;
    for (int base = 0; base < ntensor; base += ADAM_BATCH) {
        AdamBatch tb;
        tb.chunk = chunk;
        tb.ntensor = ntensor - base < ADAM_BATCH ? ntensor - base : ADAM_BATCH;
        int64_t chunks = 0;
        for (int k = 0; k < ADAM_BATCH; ++k) {
            const bool live = k < tb.ntensor;
            tb.p[k] = live ? reinterpret_cast<float *>(p[base + k]) : nullptr;
            tb.g[k] = live ? reinterpret_cast<const float *>(g[base + k]) : nullptr;
            tb.m[k] = live ? reinterpret_cast<float *>(m[base + k]) : nullptr;
            tb.v[k] = live ? reinterpret_cast<float *>(v[base + k]) : nullptr;
            tb.numel[k] = live ? numel[base + k] : 0;
            R4R_REQUIRE(!live || (tb.p[k] && tb.m[k] && tb.v[k] && tb.numel[k] >= 0), "adam_multi: tensor %d: null "
                        "p/m/v or negative size", base + k);
            tb.chunk_begin[k] = (int)chunks;
            chunks += cdiv(tb.numel[k], chunk);
            R4R_REQUIRE(chunks < (1ll << 31), "adam_multi: too many chunks");
        }
        tb.chunk_begin[ADAM_BATCH] = (int)chunks;
        if (chunks == 0) continue;
        ScopedTiming tm(R4R_TIMING_ADAM, as_stream(stream));
        adam_multi_kernel<<<(unsigned)chunks, ADAM_THREADS, 0, as_stream(stream)>>>(tb, s);
    }
    return check_launch("adam_multi");
}
