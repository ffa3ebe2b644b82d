/* r4r.h -- C ABI of libr4r_hip.so, the MI355X (gfx950) rating-prediction hot path.
 *
 * The reference (noveens/reviews4rec) has no FFI: its hot path is the list of
 * ATen ops that pytorch_models/{common_pytorch_models,MF,DeepCoNN,NARRE,TransNet}.py
 * and torch.optim.Adam dispatch per training step.  Each entry point below
 * replaces one of those call sites (cited as file:line under the reference
 * root) and is what a binding written against the reference would bind.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch owns all
 *     memory); sizes are element counts unless a name ends in _bytes
 *   - fp32 data, int64 indices (the reference's LongTensor batches,
 *     data_fast.py:101-108), int32 argmax
 *   - `stream` is a hipStream_t; nothing synchronises internally, nothing
 *     allocates, every call is asynchronous and graph-capturable
 *   - return 0 on success, <0 on error; r4r_last_error() describes the last
 *     failure on the calling thread; nothing throws or aborts
 *   - thread-compatible: no hidden global state besides the error string
 */
#ifndef R4R_H
#define R4R_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define R4R_OK 0
#define R4R_ERR_ARG (-1)      /* bad shape / null pointer / unsupported size */
#define R4R_ERR_LAUNCH (-2)   /* HIP runtime refused the launch */
#define R4R_ERR_WORKSPACE (-3)/* workspace too small */

int r4r_version(void);
const char *r4r_last_error(void);

/* ------------------------------------------------------------------------
 * TextCNN tower: word gather -> conv(3 x E, pad 2) -> relu -> global max-pool
 * Replaces  nn.Embedding lookup      DeepCoNN.py:53-54, NARRE.py:95-96, TransNet.py:100-102
 *           Conv2d + relu + max_pool common_pytorch_models.py:29-31
 * The [N,T,E] gathered activations and the [N,F,T+2] conv output are never
 * written to HBM.  Two algorithms compute the same function (selection: R4R_CONV_AUTO rule in
 * DESIGN.md 4.1b; env R4R_CONV_ALGO=direct|project pins one): the direct gather-fused MFMA conv
 * over every position, or project-then-gather (projection GEMM over the batch's distinct tokens
 * + gather-add-max over positions).
 *   table  [V, E]   frozen word vectors (E % 4 == 0)
 *   idx    [N, T]   token ids in [0, V)
 *   conv_w [F, 3, E] (= Conv2d weight [F,1,3,E]),  conv_b [F],  F <= 112
 *   pooled [N, F]   max_p relu(conv)          argmax [N, F]  position p in [0,T+2) of the
 *                                              max, or -1 where pooled == 0 (no gradient)
 * ---------------------------------------------------------------------- */
size_t r4r_textcnn_ws_bytes(int64_t N, int T, int E, int F, int64_t V);
int r4r_textcnn_fwd(const float *table, int64_t V, const int64_t *idx,
                    const float *conv_w, const float *conv_b,
                    float *pooled, int32_t *argmax,
                    void *ws, size_t ws_bytes,
                    int64_t N, int T, int E, int F, void *stream);

/* The algorithm a request resolves to: `requested` R4R_CONV_AUTO / _DIRECT / _PROJECT for N documents (per
 * tower) of T words, embed size E, F filters -> R4R_CONV_DIRECT or R4R_CONV_PROJECT (the static
 * R4R_CONV_AUTO rule and the R4R_CONV_ALGO environment pin applied). */
int r4r_conv_algo(int requested, int64_t N, int T, int E, int F);

/* The MEASURED rule behind the engines' automatic choice: project-then-gather's work follows the batch's
 * DISTINCT tokens, the direct conv's its positions.  `rows`: distinct tokens of a batch summed over its
 * towers (left by the gather kernel next to each tower's token counter: the int after it), `docs`:
 * documents summed over the towers, V: vocabulary.  Returns R4R_CONV_PROJECT or R4R_CONV_DIRECT by a cost
 * model fitted to MI355X measurements.  Host-side arithmetic, no device work. */
int r4r_conv_pick(int E, int T, int64_t docs, int64_t rows, int64_t V);

/* Arithmetic of the projection GEMM inside project-then-gather.  0 (default, the fp32 results every parity claim
 * refers to): fp32 operands on the fp32 MFMA.  1 (opt-in experiment): every operand split exactly
 * into two fp16 numbers (hi + lo), three f16-MFMA products per fp32 product into ONE fp32 accumulator, exact
 * power-of-two scaling from `table_maxabs` = max |word table| (frozen: computed once by the host) and
 * `weight_maxabs` = max |conv weights| over the towers (the host re-reads it every few steps; the scale keeps two
 * binades of headroom): ~2^-21 relative error per product at 16 / 3 the fp32-MFMA rate.  Mode 1 applies to the fused
 * steps (their host keeps the scales current); mode 2 also to r4r_textcnn_fwd (the caller vouches that the scales
 * describe the table and weights it passes).  Process-wide, like r4r_gemm_form. */
int r4r_gemm_math(int mode, float table_maxabs, float weight_maxabs);

/* Form of the projection GEMM inside project-then-gather (A/B runs and tests; results are bit-identical):
 * 1 = the balanced 7-row-tile form wherever its plan applies, 3 = the same plan in its A-resident form (the
 * workgroup's table rows stay in LDS for all of K, weight fragments go global -> registers, the K sweep runs once per
 * group of row tiles so that finished rows leave the chip while the rest is computed; word_embed_size <= 320, else
 * form 1), 4 = the weight-resident form (the tower's
 * 300 x E weights stay in LDS, table fragments go global -> registers, units of 16 rows x 64 columns round-robin
 * over the waves; word_embed_size <= 128, else form 3), 0 = always the 128-row tile form (its last, partial round
 * cut into column parts where that fills idle workgroups), 2 = the tile form with whole tiles only, -1 = take it
 * from the environment again (R4R_GEMM=tile / whole / balanced / ares / resident-weights pin 0 / 2 / 1 / 3 / 4;
 * unset: form 4 for word_embed_size <= 64, form 3 above). */
int r4r_gemm_form(int balanced);

/* Backward of the tower w.r.t. conv weight and bias (the word table is frozen:
 * Embedding.from_pretrained, DeepCoNN.py:15 -- no dgrad exists).
 * Replaces  convolution_backward / max_pool2d_with_indices_backward /
 *           threshold_backward behind common_pytorch_models.py:29-31.
 * Only the argmax window of each (n, f) carries gradient:
 *   d_conv_w[f, j, :] = sum_n g_pooled[n, f] * table[idx[n, argmax[n,f] - 2 + j], :]
 *   d_conv_b[f]       = sum_n g_pooled[n, f]            (terms with argmax < 0 dropped)
 * Outputs are overwritten, not accumulated. */
int r4r_textcnn_wgrad(const float *table, int64_t V, const int64_t *idx,
                      const float *g_pooled, const int32_t *argmax,
                      float *d_conv_w, float *d_conv_b,
                      void *ws, size_t ws_bytes,
                      int64_t N, int T, int E, int F, void *stream);

/* ------------------------------------------------------------------------
 * Small dense layers (nn.Linear with in/out <= 256), optional fused ReLU.
 * Replaces  TextCNN.fc                 common_pytorch_models.py:37
 *           DeepCoNN.final             DeepCoNN.py:21-26,69
 *           MF.projection              MF.py:26-31,62
 *           NARRE.final                NARRE.py:38-43,123
 *           Source.project             TransNet.py:17-21,33
 *   x [N, n_in], w [n_out, n_in], b [n_out], y [N, n_out]
 * bwd: g_x may be NULL; g_w / g_b are overwritten.  With relu != 0, `y` is the
 * post-activation output saved by fwd. */
int r4r_linear_fwd(const float *x, const float *w, const float *b, float *y,
                   int64_t N, int n_in, int n_out, int relu, void *stream);
size_t r4r_linear_bwd_ws_bytes(int64_t N, int n_in, int n_out);   /* 0 for small N (ws may be NULL) */
int r4r_linear_bwd(const float *x, const float *w, const float *y, const float *g_y,
                   float *g_x, float *g_w, float *g_b, void *ws, size_t ws_bytes,
                   int64_t N, int n_in, int n_out, int relu, void *stream);

/* ------------------------------------------------------------------------
 * Dropout (nn.Dropout in train mode).  Replaces common_pytorch_models.py:37,
 * DeepCoNN.py:24, MF.py:52-53, NARRE.py:27,34,39,115-116, TransNet.py:34,58,108-109.
 * mult[i] is 0 or 1/(1-p) drawn from Philox4x32-10(seed, offset + i/4); the
 * multiplier tensor is kept for backward (g_x = g_y * mult) and so a test can
 * inject the same mask into the CPU oracle.  The reference draws from torch's
 * unseeded global RNG, so streams are not comparable. */
int r4r_dropout_fwd(const float *x, float *y, float *mult, int64_t n, float p,
                    uint64_t seed, uint64_t offset, uint64_t *offset_dev, void *stream);
/* offset_dev (nullable, DEVICE): the stream position lives in device memory: the kernel uses
 * offset + *offset_dev and *offset_dev is advanced by ceil(n/4) afterwards on the same stream,
 * so a captured hipGraph draws fresh masks on every replay. */
int r4r_counter_add(uint64_t *counter, uint64_t delta, void *stream);
int r4r_mul(const float *a, const float *b, float *out, int64_t n, void *stream);
int r4r_add(const float *a, const float *b, float *out, int64_t n, void *stream);

/* ------------------------------------------------------------------------
 * Factorisation machine head (no global bias).
 * Replaces  TorchFM.forward            common_pytorch_models.py:49-57
 *   out[b] = 0.5 * (sum_k (x V)_k^2 - sum_k (x^2 V^2)_k) + lin_w . x + lin_b
 *   x [N, n], V [n, k], lin_w [n], lin_b [1], out [N]     (n <= 512: the reference bounds neither latent_size nor
 *   the FM width -- hyper_params.py:63, DeepCoNN.py:32 reads 2 x latent_size inputs; k <= 4096)
 * bwd overwrites g_x [N,n], g_V [n,k], g_lin_w [n], g_lin_b [1]. */
int r4r_fm_fwd(const float *x, const float *V, const float *lin_w, const float *lin_b,
               float *out, int64_t N, int n, int k, void *stream);
int r4r_fm_bwd(const float *x, const float *V, const float *lin_w, const float *g_out,
               float *g_x, float *g_V, float *g_lin_w, float *g_lin_b,
               int64_t N, int n, int k, void *stream);

/* ------------------------------------------------------------------------
 * ID-embedding / bias-vector gathers and their dense gradients.
 * Replaces  nn.Embedding(user/item)    MF.py:52-53, NARRE.py:110,112,115-116, TransNet.py:108-109
 *           Parameter.gather           MF.py:45-46, DeepCoNN.py:70-71, NARRE.py:87-88
 *           embedding_dense_backward / scatter_add behind them
 *   table [R, D] (D = 1 for a bias vector), idx [n], out [n, D]
 * scatter_add zero-fills g_table [R, D] and adds every g_out row into it (the
 * reference's gradients for these tables are DENSE). */
int r4r_embed_gather(const float *table, const int64_t *idx, float *out,
                     int64_t R, int D, int64_t n, void *stream);
int r4r_embed_scatter_add(const float *g_out, const int64_t *idx, float *g_table,
                          int64_t R, int D, int64_t n, void *stream);
/* Same result, fixed summation order (the first occurrence of a row adds its later duplicates in
 * ascending entry order; no atomics): bit-identical on every run and every rank.  Used to rebuild
 * the dense gradient from the all-gathered compact (row-id, grad-row) lists under data parallelism
 * (no reference counterpart: the reference is single-process). */
int r4r_embed_scatter_add_ordered(const float *g_out, const int64_t *idx, float *g_table,
                                  int64_t R, int D, int64_t n, void *stream);

/* ------------------------------------------------------------------------
 * Rating head pieces.
 * rowdot: out[b] = sum_d a[b,d] * c[b,d]          torch.sum(user*item,-1), MF.py:57
 * bias_head: out[b] = (r ? r[b] : 0) + user_bias[uid[b]] + item_bias[iid[b]] + global_bias[0]
 *           MF.py:45-49,58,68; DeepCoNN.py:70-72; NARRE.py:87-88,124
 *           user_bias == NULL (then item_bias/uid/iid are ignored): out[b] = r[b] + global_bias[0],
 *           the 'deepconn' head  self.global_bias + fm(cat)[:, 0]  (DeepCoNN.py:65)
 * bias_head_bwd overwrites the DENSE g_user_bias [RU], g_item_bias [RI] (skipped when
 * g_user_bias == NULL) and g_global [1]  (g_r == g_out, the caller aliases it). */
int r4r_rowdot_fwd(const float *a, const float *c, float *out, int64_t N, int D, void *stream);
int r4r_rowdot_bwd(const float *a, const float *c, const float *g_out, float *g_a, float *g_c,
                   int64_t N, int D, void *stream);
int r4r_bias_head_fwd(const float *r, const float *user_bias, const float *item_bias,
                      const float *global_bias, const int64_t *uid, const int64_t *iid,
                      float *out, int64_t N, void *stream);
int r4r_bias_head_bwd(const float *g_out, const int64_t *uid, const int64_t *iid,
                      float *g_user_bias, float *g_item_bias, float *g_global,
                      int64_t RU, int64_t RI, int64_t N, void *stream);

/* ------------------------------------------------------------------------
 * NARRE review-level attention.  Replaces NARRE.attention, NARRE.py:53-64 with
 * scorer = Linear(2L->L), ReLU, Dropout, Linear(L->1) (NARRE.py:24-36):
 *   h      = relu(W0 [x ; other] + b0) * mult          (mult NULL in eval / p = 0)
 *   a      = softmax_R(w3 . h + b3)                    (padded reviews NOT masked)
 *   out[n] = sum_r a[n,r] * x[n,r,:]
 *   x, other [N, R, L]; W0 [L, 2L]; b0 [L]; w3 [L]; b3 [1]; out [N, L]
 * fwd saves h [N,R,L] (post relu, post dropout) and a [N,R] for backward.
 * bwd overwrites g_x, g_other [N,R,L], g_W0, g_b0, g_w3, g_b3; ws holds the
 * per-row scorer gradients it reduces over.   L <= 32, R <= 32. */
size_t r4r_narre_attn_ws_bytes(int64_t N, int R, int L);
int r4r_narre_attn_fwd(const float *x, const float *other, const float *W0, const float *b0,
                       const float *w3, const float *b3, const float *mult,
                       float *out, float *h_save, float *a_save,
                       int64_t N, int R, int L, void *stream);
int r4r_narre_attn_bwd(const float *x, const float *other, const float *W0, const float *w3,
                       const float *mult, const float *h_save, const float *a_save,
                       const float *g_out,
                       float *g_x, float *g_other, float *g_W0, float *g_b0, float *g_w3, float *g_b3,
                       void *ws, size_t ws_bytes,
                       int64_t N, int R, int L, void *stream);

/* ------------------------------------------------------------------------
 * Per-example squared error + its gradient for mean(SE) over `denom` examples.
 * Replaces  MSELoss.forward + torch.mean   loss.py:7-11, main.py:56-58
 *   se[b] = (out[b]-y[b])^2 ; g_out[b] = 2 (out[b]-y[b]) / denom   (g_out may be NULL) */
int r4r_mse_fwd_bwd(const float *out, const float *y, float *se, float *g_out,
                    int64_t N, float denom, void *stream);

/* TransNet transform loss  mean_n sum_l (a[n,l] - b[n,l])^2  (a = source.ir, b = target.ir).
 * Replaces  torch.mean(torch.sum(torch.pow(.., 2), -1))   TransNet.py:121
 *   a, b [N, L]; out [1]; bwd: g_out [1] -> g_a, g_b [N, L] */
int r4r_sqdist_mean_fwd(const float *a, const float *b, float *out, int64_t N, int L, void *stream);
int r4r_sqdist_mean_bwd(const float *a, const float *b, const float *g_out, float *g_a, float *g_b,
                        int64_t N, int L, void *stream);

/* ------------------------------------------------------------------------
 * Dense fused Adam over many tensors in one launch.
 * Replaces  torch.optim.Adam(lr, weight_decay).step()   main.py:94-96,60
 * (defaults betas (0.9, 0.999), eps 1e-8, L2 decay added to the gradient,
 * bias correction; every element of every listed tensor moves each step,
 * SURVEY.md fact 4).
 *   p/g/m/v : HOST arrays of `ntensor` DEVICE pointers, numel : HOST int64[ntensor] (the one
 *   exception to "every pointer is a device pointer": the tensor list is passed by value in
 *   the kernel arguments, 16 tensors per launch, so no descriptor table lives in HBM and no
 *   H2D copy happens per step).  One workgroup per r4r_adam_chunk_elems()-element chunk (1024
 *   when the whole list is under 4 M elements: latency-bound, more and shorter workgroups).
 *   g[i] == 0 means "gradient is zero" (weight decay still applies).
 *   step = 1-based step count shared by all listed tensors; or step_dev (nullable, DEVICE):
 *   a counter of COMPLETED steps in device memory -- the kernel then uses t = *step_dev + 1 and
 *   derives the bias corrections itself (for hipGraph replay; the caller advances the counter
 *   with r4r_counter_add after the launch). */
int r4r_adam_chunk_elems(void);
int r4r_adam_multi(int ntensor, const uint64_t *p, const uint64_t *g, const uint64_t *m, const uint64_t *v,
                   const int64_t *numel, float lr, double beta1, double beta2, float eps,
                   float weight_decay, int64_t step, const int64_t *step_dev, void *stream);
/* Data-parallel Adam on one flat buffer: gradient = sum over the `world` per-rank buffers of an
 * all_gather (`gathered` [world][numel], added in rank order: identical bits on every rank),
 * update as above in the same pass; g_sum (nullable) receives the summed gradient.  numel % 4 == 0,
 * 16-byte aligned buffers (the flat buffers of r4r_deepconn_layout are). */
int r4r_adam_gathered(float *p, const float *gathered, int world, float *g_sum, float *m, float *v,
                      int64_t numel, float lr, double beta1, double beta2, float eps,
                      float weight_decay, int64_t step, void *stream);
/* The same behind a device-side guard: `abort_word` (nullable, DEVICE uint32) != 0 makes the launch a no-op.  The
 * peer exchange below passes its timed_out word: slots a missing rank never filled must not be summed. */
int r4r_adam_gathered_guarded(float *p, const float *gathered, int world, float *g_sum, float *m, float *v,
                              int64_t numel, float lr, double beta1, double beta2, float eps,
                              float weight_decay, int64_t step, const uint32_t *abort_word, void *stream);

/* Device-side form of that all_gather over peer-mapped buffers (the exchange step of the data-parallel form of
 * main.py:56-60; the reference itself is single-process, main.py:407).  Every rank owns two gathered buffers
 * [world][numel] (alternating by step parity) and a flag array [world] of uint32, all zero-initialised, and has
 * them mapped by its peers:
 *   r4r_peer_segment_create: one zero-filled FINE-GRAINED device allocation of `bytes` on the current device (the
 *     flags are polled while peers write them: coarse-grained memory promises visibility at kernel boundaries only)
 *     and its 64-byte IPC handle, which the host ships to the peers by any means it has;
 *     r4r_peer_segment_open maps a peer's handle (-> *ptr), _close unmaps it, _destroy frees one's own segment.
 *   r4r_peer_push: copies `src` [numel] into slot `rank` of every rank's gathered buffer (peer_dst[r] = rank r's
 *     buffer of this step's parity, device pointers as uint64), then -- once every workgroup's stores are fenced at
 *     system scope -- writes `epoch` (the 1-based step) to element `rank` of every rank's flag array
 *     (peer_flags[r]).  `arrive`: one zero-initialised uint32 in this rank's memory.  numel % 4 == 0, 16-byte aligned.
 *     With `wait_flags` (MY flag array) the same launch then waits as r4r_peer_wait does: the exchange is one launch.
 *   r4r_peer_wait: one wave spins until all `world` elements of MY flag array have reached `epoch`, at most
 *     `timeout_s` seconds (then *timed_out = 1 + the missing rank, and the launch ends: the caller checks it).
 * r4r_adam_gathered on the buffer then sums the slots in rank order: identical bits on every rank.  world <= 16. */
int r4r_peer_segment_create(int64_t bytes, void **ptr, uint8_t *handle);
int r4r_peer_segment_open(const uint8_t *handle, void **ptr);
int r4r_peer_segment_close(void *ptr);
int r4r_peer_segment_destroy(void *ptr);
int r4r_peer_push(const float *src, int64_t numel, const uint64_t *peer_dst, const uint64_t *peer_flags,
                  uint32_t *arrive, int rank, int world, uint32_t epoch, const uint32_t *wait_flags,
                  uint32_t *timed_out, double timeout_s, void *stream);
int r4r_peer_wait(const uint32_t *flags, int world, uint32_t epoch, uint32_t *timed_out, double timeout_s, void *stream);


/* ------------------------------------------------------------------------
 * Fused DeepCoNN step ('deepconn' mode): the whole of DeepCoNN.forward
 * (DeepCoNN.py:37-66) + MSELoss/mean (loss.py:7-11, main.py:56-58) + backward in
 * ONE call that enqueues 6 kernels (both towers share the conv launch).
 * Replaces, per training step, every ATen op main.py:32,56-59 dispatches.
 *
 * The trainable parameters of this mode live in ONE flat fp32 buffer; slot i of
 * r4r_deepconn_layout() (HOST out-arrays of r4r_deepconn_nparam() entries, offsets
 * in floats, 16-byte aligned) is, in order:
 *   user_conv.convs.0.weight, user_conv.convs.0.bias, user_conv.fc.weight,
 *   user_conv.fc.bias, item_conv.(same four), fm.V, fm.lin.weight, fm.lin.bias,
 *   global_bias            (`final`, user_bias, item_bias are unused in this mode,
 *                           DeepCoNN.py:64-66, and are not part of the buffer)
 * flat_g has the same layout and is OVERWRITTEN with d mean(SE)/d param, where the
 * mean is over 1/inv_denom examples (pass 1/B_global under data parallelism).
 *   flat_g == NULL : forward only (eval, or y == NULL for pure prediction)
 *   training != 0  : dropout(p) on the FC outputs, Philox4x32-10(seed, offset + b*2L + i)
 *   conv_algo      : R4R_CONV_AUTO | R4R_CONV_DIRECT (gather-fused MFMA conv over every position)
 *                    | R4R_CONV_PROJECT (projection GEMM over the batch's distinct tokens, then a
 *                    gather-add-max over positions; same function, different summation order)
 *   token_buffer   : which of the two token-state buffers of the workspace this step uses (0/1)
 *   tokens_ready   : != 0 when that buffer already holds this batch's compacted tokens
 *                    (prepared by r4r_deepconn_tokens or by the previous step, below)
 *   next_user_idx / next_item_idx (both or neither; training steps only): the NEXT batch of the
 *                    same B and T.  Its token marks ride on this step's backward launch and their
 *                    compaction on the gradient-reduce launch, into buffer token_buffer ^ 1; call
 *                    the next step with that buffer and tokens_ready = 1.
 *   flat_m / flat_v (both or neither; layout of flat_p): when given, the step is also the
 *                    optimiser -- torch.optim.Adam(lr, betas, eps, weight_decay) update number
 *                    adam_step (1-based) of flat_p in place (main.py:94-96,60), done by the launch
 *                    that finishes the gradients; flat_g still receives the gradients.  Leave NULL
 *                    under data parallelism (all-reduce flat_g, then r4r_adam_multi).
 *   pred [B], se [B] (se required when y != NULL); sse_accum (nullable device scalar)
 *   is incremented by sum_b se[b] -- the host's running metric (main.py:57) without a
 *   per-step device->host sync. */
#define R4R_CONV_AUTO 0
#define R4R_CONV_DIRECT 1
#define R4R_CONV_PROJECT 2
int r4r_deepconn_nparam(void);
int r4r_deepconn_layout(int E, int L, int64_t *offsets, int64_t *sizes, int64_t *total);
size_t r4r_deepconn_ws_bytes(int64_t B, int T, int E, int L, int64_t V);
size_t r4r_deepconn_ws_mult_offset(int64_t B, int T, int E, int L, int64_t V);   /* [B,2L] dropout multipliers (tests) */
size_t r4r_deepconn_ws_count_offset(int64_t B, int T, int E, int L, int64_t V, int tower, int buffer);   /* token counter of (tower, token buffer): int[0] live count, int[1] the count the last gather consumed */
int r4r_deepconn_step(const float *table, int64_t V, const int64_t *user_idx, const int64_t *item_idx,
                      const float *y, float *flat_p, float *flat_g,
                      float *pred, float *se, float *sse_accum,
                      void *ws, size_t ws_bytes,
                      int64_t B, int T, int E, int L,
                      float dropout_p, int training, uint64_t seed, uint64_t offset,
                      float inv_denom, int conv_algo, int token_buffer, int tokens_ready,
                      const int64_t *next_user_idx, const int64_t *next_item_idx,
                      float *flat_m, float *flat_v, float lr, double beta1, double beta2, float eps,
                      float weight_decay, int64_t adam_step, void *stream);
/* Token compaction of a batch (distinct tokens -> dense rows) into token-state buffer 0 or 1 of the
 * workspace.  It depends only on the indices, so the caller may run it for batch k+1 on another
 * stream while step k computes, and then pass token_buffer / tokens_ready = 1 to step k+1
 * (otherwise the step does it itself).  The workspace must be zero-filled when created; the
 * kernels keep the flag / count words zero between calls. */
int r4r_deepconn_tokens(const int64_t *user_idx, const int64_t *item_idx, void *ws, size_t ws_bytes,
                        int64_t B, int T, int E, int L, int64_t V, int conv_algo, int token_buffer,
                        int discard, void *stream);   /* discard != 0: reset a prepared but unused buffer */

/* ------------------------------------------------------------------------
 * Live kernel timing for bench.py's roofline leg (no reference counterpart: the
 * reference has no profiler hooks).  When enabled, each
 * instrumented entry point brackets its dominant kernel -- only that kernel --
 * with hipEvents (recycled from a pool) on the launch stream; read() synchronises and
 * accumulates.  Each instrumented launch costs two event records, so bench.py instruments only
 * the kernels its roofline needs.
 * `total_ms` and `count` are HOST pointers. */
#define R4R_TIMING_TEXTCNN_FWD 0    /* textcnn_fwd_kernel (the MFMA conv tile kernel) */
#define R4R_TIMING_TEXTCNN_WGRAD 1  /* textcnn_wgrad_kernel */
#define R4R_TIMING_ADAM 2           /* adam_multi_kernel */
#define R4R_TIMING_PROJ_GEMM 3      /* proj_gemm_kernel (projection of the batch's distinct tokens) */
#define R4R_TIMING_PROJ_GATHER 4    /* proj_gather_max_kernel (gather-add-max over positions) */
#define R4R_TIMING_SLOTS 8
int r4r_timing_enable(int slot_mask);   /* bit i instruments slot i; 0 switches timing off */
int r4r_timing_read(int slot, double *total_ms, int64_t *count, int reset);

/* ------------------------------------------------------------------------
 * Fused native step for the ID-only recommenders: model_type 'MF_dot' and 'bias_only'
 * Replaces, per training step, MF.forward (MF.py:39-58: user/item bias gathers, the two
 * ID-embedding gathers + dropout, the row dot product), MSELoss (loss.py:7-11), loss.backward()
 * and torch.optim.Adam.step() (main.py:56-60,94-96) by TWO launches.  The dense gradient of an
 * ID table is never materialised: gradient rows stay compact ([B, D]) and the Adam sweep treats a
 * row no rating touched as gradient zero (24 B/element instead of 28 + a zero fill); a touched row
 * is updated by the wave of its first rating, which sums the row's entries in a fixed order
 * (deterministic, no atomics).
 *   p / m / v : HOST arrays of 5 DEVICE pointers -- user_embedding.weight [n_users, D],
 *               item_embedding.weight [n_items, D], user_bias [n_users], item_bias [n_items],
 *               global_bias [1] (n_users = total_users + 1, n_items = total_items + 1, MF.py:14-24);
 *               the two table entries are ignored when D == 0 ('bias_only').
 *               m == v == NULL: forward only (pred, and se when y is given).
 *   dropout   : Philox4x32-10(seed, offset + b*2D + d) for the user row, + D for the item row
 *   adam_step : 1-based update count; also tags the rows this step touched (workspace)
 *   ws        : r4r_mf_ws_bytes; its first r4r_mf_ws_persist_bytes (per-row step tags and the
 *               owner-election slots) must be ZERO on first use and are kept consistent by the
 *               kernels as long as adam_step only grows -- allocate once, zero once; a caller
 *               that switches buffers (another B) carries that head over.
 *   B <= 16384 for training steps.  sse_accum (nullable) += sum_b se[b] on training steps. */
size_t r4r_mf_ws_bytes(int64_t B, int D, int64_t n_users, int64_t n_items);
size_t r4r_mf_ws_persist_bytes(int64_t B, int D, int64_t n_users, int64_t n_items);
size_t r4r_mf_ws_mult_offset(int64_t B, int D, int64_t n_users, int64_t n_items);   /* [B,2D] dropout multipliers (tests) */
size_t r4r_mf_ws_grad_offset(int64_t B, int D, int64_t n_users, int64_t n_items, int which);   /* 0: user rows [B,D], 1: item rows, 2: d loss/d pred [B] (tests) */
int r4r_mf_step(const int64_t *uid, const int64_t *iid, const float *y,
                const uint64_t *p, const uint64_t *m, const uint64_t *v,
                int64_t n_users, int64_t n_items, int D,
                float *pred, float *se, float *sse_accum, void *ws, size_t ws_bytes, int64_t B,
                float dropout_p, int training, uint64_t seed, uint64_t offset, float inv_denom,
                int sweep_period, int64_t sweep_base, int sweep_all,
                float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                void *stream);

/* sweep_period P in 2..8: the Adam sweep over the two ID tables, temporally blocked on a SCHEDULE (round 4).  A table
 * element no rating names goes through the same gradient-zero update whether it is applied now or together with the
 * next ones (it reads nothing but its own p, m, v and the step's two bias corrections), so chunk c of 1,024 table
 * elements is visited only at the steps s with (c % P + c / P + s) % P == 0 -- one chunk of every P consecutive
 * ones per step, a launch of exactly the due chunks -- and then takes all its pending updates in order, each with its
 * own step's scalars: the fp32 operations per element are those of the dense sweep (bit-identical tables and
 * moments: tests/test_gpu_full_size.py), the traffic a P-th.  Nothing is announced and no per-chunk state is kept:
 * the step of a chunk's last visit is a function of (c, step), a row a rating names is current through
 * max(sweep_base, that visit, the row's own last update [kept per row in ws]), and whoever needs it newer applies the
 * missing updates on the way -- the forward in registers, the row's entry wave before its gradient update.
 *   sweep_base : a completed step through which EVERY element is current (0 before the first step).  The caller keeps
 *                it: it becomes adam_step after a call with sweep_all = 1 or P = 1 and after r4r_mf_rows_flush, and
 *                stays otherwise.  At most 8 updates may be pending anywhere (guaranteed while every step since
 *                sweep_base ran this function with the same P); the int at r4r_mf_ws_flag_offset is set if not.
 *   sweep_all  : 1 = visit every chunk now (the plain dense sweep, applying whatever is pending under the schedule
 *                (P, sweep_base)); required on the step that changes P.
 * r4r_mf_rows_flush(adam_step = the last COMPLETED step) brings every element up to date without a step; a forward-only
 * r4r_mf_step (m == NULL) and anything else that reads the tables needs it first.  16-byte aligned tables; others
 * are swept densely every step. */
int r4r_mf_rows_flush(const uint64_t *p, const uint64_t *m, const uint64_t *v,
                      int64_t n_users, int64_t n_items, int D, void *ws, size_t ws_bytes, int64_t B,
                      int sweep_period, int64_t sweep_base,
                      float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                      void *stream);
size_t r4r_mf_ws_flag_offset(int64_t B, int D, int64_t n_users, int64_t n_items);

/* accum[0] += sum of se[0 .. n) in a fixed order (the running train metric of main.py:57, kept on the device). */
int r4r_sse_accumulate(const float *se, int64_t n, float *accum, void *stream);

/* ---- MF under data parallelism: one process per GPU, replicated tables.
 * r4r_mf_grad: this rank's forward + compact gradient rows into one packed `block`
 *   (r4r_mf_dp_block_bytes(B_pad, D) bytes: uid32 [B_pad] | iid32 | g | gu [B_pad, D] | gi; entries
 *   past this rank's B carry id -1: ragged shards).  inv_denom = 1 / B_global.
 * The caller all_gathers the blocks (rank order), then every rank calls
 * r4r_mf_apply(blocks [world], ...): row tags + owner election over the gathered entries, then the
 *   same tagged sweep as r4r_mf_step -- identical bits on every rank, and identical to the
 *   single-process step on the concatenated batch.  ws: r4r_mf_ws_bytes(world * B_pad, ...), zeroed
 *   once; world * B_pad <= 16384. */
size_t r4r_mf_dp_block_bytes(int64_t B_pad, int D);
int r4r_mf_grad(const int64_t *uid, const int64_t *iid, const float *y, const uint64_t *p,
                const uint64_t *m, const uint64_t *v,
                int64_t n_users, int64_t n_items, int D, float *pred, float *se, void *block, float *mult,
                int64_t B, int64_t B_pad, float dropout_p, int training, uint64_t seed, uint64_t offset,
                float inv_denom, void *ws, int sweep_period, int64_t sweep_base,
                float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                void *stream);
int r4r_mf_apply(const void *blocks, int world, int64_t B_pad, const uint64_t *p, const uint64_t *m,
                 const uint64_t *v, int64_t n_users, int64_t n_items, int D, void *ws, size_t ws_bytes,
                 int sweep_period, int64_t sweep_base, int sweep_all,
                 const float *se, int64_t se_n, float *sse_accum,
                 float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                 void *stream);
/* se [se_n] / sse_accum (nullable): sse_accum[0] += the sum of THIS rank's se in a fixed order (the running train metric
 * of main.py:57), on the launch's global-bias workgroup instead of a launch of its own.
 * The scheduled sweep under data parallelism: r4r_mf_apply(sweep_period, sweep_base, sweep_all) -- the SAME values on
 * every rank -- is r4r_mf_step's sweep over the gathered entries; r4r_mf_grad(m, v, ws = r4r_mf_apply's workspace,
 * the same schedule and optimiser scalars; all NULL / ignored when nothing can be pending) reads the rows its ratings
 * name as of step adam_step - 1.  r4r_mf_rows_flush on every rank before anything else reads the tables.
 * Up to 2,048 gathered entries r4r_mf_apply is ONE launch: its workgroups read the ids out of the blocks and find the
 * rows the step names themselves (beyond, a registering launch runs first).
 *
 * The same step with NO collective call -- the exchange rides on the two launches, over peer-mapped memory (the
 * segments, flags and epochs of r4r_peer_* below; one process per GPU on an xGMI mesh):
 * r4r_mf_grad_push: r4r_mf_grad whose block goes straight into slot `rank` of EVERY rank's gathered buffer
 *   (peer_dst[r] = rank r's buffer of this epoch's parity, world slots of r4r_mf_dp_block_bytes each); the launch's last
 *   workgroup raises flags[r][rank] = epoch on every rank (system-scope release).  arrive: one zeroed uint32 of this rank.
 * r4r_mf_apply_peer: r4r_mf_apply on this rank's gathered buffer, every workgroup first waiting (bounded by
 *   timeout_s; *timed_out = 1 + the missing rank otherwise) until all `world` flags of wait_flags (THIS rank's flag
 *   array) have reached `epoch`.  world * B_pad <= 2,048. */
int r4r_mf_grad_push(const int64_t *uid, const int64_t *iid, const float *y, const uint64_t *p,
                     const uint64_t *m, const uint64_t *v,
                     int64_t n_users, int64_t n_items, int D, float *pred, float *se, float *mult,
                     int64_t B, int64_t B_pad, float dropout_p, int training, uint64_t seed, uint64_t offset,
                     float inv_denom, void *ws, int sweep_period, int64_t sweep_base,
                     float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                     const uint64_t *peer_dst, const uint64_t *peer_flags, uint32_t *arrive, int rank, int world,
                     uint32_t epoch, void *stream);
int r4r_mf_apply_peer(const void *blocks, int world, int64_t B_pad, const uint64_t *p, const uint64_t *m,
                      const uint64_t *v, int64_t n_users, int64_t n_items, int D, void *ws, size_t ws_bytes,
                      int sweep_period, int64_t sweep_base, int sweep_all,
                      const float *se, int64_t se_n, float *sse_accum,
                      float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                      const uint32_t *wait_flags, uint32_t epoch, uint32_t *timed_out, double timeout_s,
                      void *stream);

/* ---- data parallel: the glue around the step's ONE all_gather of the ranks' compact entries (csrc/dp_pack.hip).
 * A family's entries are `nfields` arrays of `widths[f]` 4-byte units per rating (int64 ids: 2; d loss / d pred: 1; a
 * gradient row of an ID table: its width).  r4r_dp_pack copies this rank's n entries of every field into one block
 * (r4r_dp_block_bytes; entries n .. B_pad - 1 are padding: all ones where pad_ones[f] -- an int64 id of -1 --, zero
 * elsewhere); the caller all_gathers the blocks (rank order); r4r_dp_unpack writes field f of all ranks, rank-major,
 * to dst[f] [world * B_pad, widths[f]]: the arrays the r4r_*_rows_apply launches take.  At most 8 fields. */
size_t r4r_dp_block_bytes(int nfields, const int *widths, int64_t B_pad);
int r4r_dp_pack(int nfields, const uint64_t *src, const int *widths, const int *pad_ones, int64_t n,
                int64_t B_pad, void *block, void *stream);
int r4r_dp_unpack(int nfields, const uint64_t *dst, const int *widths, const void *blocks, int world,
                  int64_t B_pad, void *stream);

/* ---- fused native step for NARRE (pytorch_models/NARRE.py:10-124)
 * Replaces, per training step: the word gathers + TextCNN over the B*R review documents of each
 * side, TextCNN's FC + dropout, both attention scorers + softmax (NARRE.py:53-64), the four
 * ID-embedding gathers, the interaction, `final`, the bias head, MSELoss (loss.py:7-11),
 * loss.backward() and torch.optim.Adam.step() (main.py:56-60,94-96) -- five launches.
 *   user_reviews / item_reviews [B, R, T] token ids; reviewed_items / users_who_reviewed [B, R]
 *   (R = narre_num_reviews = the neighbour count of data.py:274-279); uid / iid [B].
 *   flat_p / flat_g / flat_m / flat_v : the DENSE parameters in the layout of r4r_narre_layout
 *     (21 slots, 16-byte aligned): user_conv.convs.0.weight, .bias, user_conv.fc.weight, .bias,
 *     item_conv.(same four), attention_scorer_user.0.weight, .0.bias, .3.weight, .3.bias,
 *     attention_scorer_item.(same four), final.1.weight, .1.bias, .3.weight, .3.bias, global_bias.
 *     flat_g == NULL: forward only.
 *   rows_p / rows_m / rows_v : HOST arrays of 4 DEVICE pointers -- user_embedding.weight
 *     [n_users, L], item_embedding.weight [n_items, L], user_bias [n_users], item_bias [n_items]
 *     (n_users = total_users + 2, n_items = total_items + 2: NARRE.py:17-18,44-45).  Their dense
 *     gradients are never materialised (compact rows + a tagged Adam sweep, as r4r_mf_step).
 *   dropout: Philox4x32-10(seed, offset + b*(4RL+3L) + k), k = site-major: user_conv.dropout [R,L],
 *     item_conv.dropout, attention_scorer_user.2, attention_scorer_item.2, dropout.user [L],
 *     dropout.item, final.0.
 *   token_buffer / tokens_ready / next_*_reviews / conv_algo: as r4r_deepconn_step.
 *   ws: r4r_narre_ws_bytes, ZERO on first use (row tags, token flags, gradient-row padding). */
int r4r_narre_nparam(void);
int r4r_narre_layout(int E, int L, int64_t *offsets, int64_t *sizes, int64_t *total);
size_t r4r_narre_ws_bytes(int64_t B, int R, int T, int E, int L, int64_t V, int64_t n_users, int64_t n_items);
size_t r4r_narre_ws_offset(int64_t B, int R, int T, int E, int L, int64_t V, int64_t n_users, int64_t n_items,
                           int which);   /* 0 dropout multipliers, 1/2 compact rows user/item, 3/4 their ids, 5 d loss/d pred (tests);
                                         6 + 2*tower + buffer: compaction counter of a token buffer (zero the int to discard
                                         a prepared-but-unused token state) */
int r4r_narre_step(const float *table, int64_t V,
                   const int64_t *user_reviews, const int64_t *item_reviews,
                   const int64_t *reviewed_items, const int64_t *users_who_reviewed,
                   const int64_t *uid, const int64_t *iid, const float *y,
                   float *flat_p, float *flat_g, float *flat_m, float *flat_v,
                   const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                   int64_t n_users, int64_t n_items,
                   float *pred, float *se, float *sse_accum, void *ws, size_t ws_bytes,
                   int64_t B, int R, int T, int E, int L,
                   float dropout_p, int training, uint64_t seed, uint64_t offset, float inv_denom,
                   int conv_algo, int token_buffer, int tokens_ready,
                   const int64_t *next_user_reviews, const int64_t *next_item_reviews,
                   float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                   void *stream);

/* Data parallel: r4r_narre_step with flat_m == NULL computes gradients only (flat_g; the compact ID entries
 * at r4r_narre_ws_offset 1..5); after the exchange -- all-reduce flat_g + r4r_adam_multi, all_gather of the
 * ranks' entries -- this updates the two ID tables and bias vectors from ALL ranks' entries in the order
 * given (ids -1 pad ragged shards).  g_entry: an entry's bias gradient (d loss / d pred for a rating's own
 * user / item entry, 0 for a neighbour entry).  Same `ws` and shape arguments as the step.  entries <= 16384. */
int r4r_narre_rows_apply(const int64_t *gid0, const int64_t *gid1, const float *grow0, const float *grow1,
                         const float *g_entry, int64_t entries,
                         const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                         int64_t n_users, int64_t n_items, void *ws, size_t ws_bytes,
                         int64_t B, int R, int T, int E, int L, int64_t V,
                         float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                         void *stream);
/* The same update with one exchange and no glue launches (world * B_pad * (1 + R) <= 16,384 entries, latent_size <= 32):
 * r4r_narre_dp_block packs this rank's entries of the LAST gradients-only r4r_narre_step on `ws` into `block`
 *   (r4r_narre_dp_block_bytes(B_pad, R, L) bytes: ids as int32, an entry's bias gradient, the gradient rows; the B_pad
 *   self entries first, then the neighbour entries; entries past the rank's own carry id -1), the caller all_gathers
 *   the blocks (rank order) and every rank calls
 * r4r_narre_rows_apply_blocks(blocks [world], ...): r4r_narre_rows_apply straight over the blocks, entries in (rank,
 *   in-block) order on every rank. */
size_t r4r_narre_dp_block_bytes(int64_t B_pad, int R, int L);
int r4r_narre_dp_block(void *ws, size_t ws_bytes, int64_t B, int R, int T, int E, int L, int64_t V,
                       int64_t n_users, int64_t n_items, void *block, int64_t B_pad, void *stream);
int r4r_narre_rows_apply_blocks(const void *blocks, int world, int64_t B_pad,
                                const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                                int64_t n_users, int64_t n_items, void *ws, size_t ws_bytes,
                                int64_t B, int R, int T, int E, int L, int64_t V,
                                float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                                void *stream);

/* The same for ANY number of entries (a data-parallel step at a global batch of 8,192 gathers 90,112 entries per
 * table; a single-process step beyond the fused launch's 4,096): the named rows go through r4r_rows_apply_large
 * (below), the rows nobody names through the same tagged sweep.  `scratch`: r4r_rows_large_ws_bytes(entries) bytes. */
int r4r_narre_rows_apply_large(const int64_t *gid0, const int64_t *gid1, const float *grow0, const float *grow1,
                               const float *g_entry, int64_t entries,
                               const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                               int64_t n_users, int64_t n_items, void *ws, size_t ws_bytes,
                               int64_t B, int R, int T, int E, int L, int64_t V,
                               float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                               void *scratch, size_t scratch_bytes, void *stream);

/* Dense-Adam update (torch.optim.Adam as at main.py:94-96) of the rows of ONE ID table [rows, W] -- and, with bp, of
 * its bias vector [rows] -- that `entries` compact entries name: a row's gradient is the sum of grads[e, :] over the
 * entries e with ids[e] == row (ids outside 0..rows-1 are padding), its bias gradient the sum of gbias[e] (nullable:
 * zero); rows no entry names are NOT touched (the caller's sweep).  Any number of entries; deterministic (no
 * floating-point atomics, no scheduling-dependent order): replicas applying the same entries hold the same bits.
 * W <= 1024.  `scratch`: r4r_rows_large_ws_bytes(entries) bytes of device memory, contents irrelevant. */
size_t r4r_rows_large_ws_bytes(int64_t entries);
int r4r_rows_apply_large(const int64_t *ids, const float *grads, const float *gbias, int64_t entries, int W,
                         float *p, float *m, float *v, float *bp, float *bm, float *bv, int64_t rows,
                         void *scratch, size_t scratch_bytes,
                         float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                         void *stream);

/* ---- fused native step for DeepCoNN++ (DeepCoNN.py:37-72, model_type 'deepconn++'): the two
 * TextCNN towers, `final` = Linear(2L, L) -> ReLU -> Dropout -> Linear(L, 1), user / item / global
 * bias.  Same structure and arguments as r4r_narre_step; user_idx / item_idx [B, T] as in
 * r4r_deepconn_step.  Flat layout (13 slots, r4r_deepconnpp_layout): user_conv.convs.0.weight, .bias,
 * user_conv.fc.weight, .bias, item_conv.(same four), final.0.weight, final.0.bias, final.3.weight,
 * final.3.bias, global_bias.  rows_p / rows_m / rows_v: HOST arrays of 2 DEVICE pointers -- user_bias
 * [n_users], item_bias [n_items] (n = total + 2).  Dropout draws per rating: user_conv.dropout [L],
 * item_conv.dropout [L], final.2 [L] at Philox counter offset + b*3L + k.  (The `fm` module the
 * reference also constructs is unused in this mode and never trained.) */
int r4r_deepconnpp_nparam(void);
int r4r_deepconnpp_layout(int E, int L, int64_t *offsets, int64_t *sizes, int64_t *total);
size_t r4r_deepconnpp_ws_bytes(int64_t B, int T, int E, int L, int64_t V, int64_t n_users, int64_t n_items);
size_t r4r_deepconnpp_ws_offset(int64_t B, int T, int E, int L, int64_t V, int64_t n_users, int64_t n_items,
                                int which);   /* 0 dropout multipliers [B,3L], 5 d loss/d pred, 6 + 2*tower + buffer: token counter */
int r4r_deepconnpp_step(const float *table, int64_t V, const int64_t *user_idx, const int64_t *item_idx,
                        const int64_t *uid, const int64_t *iid, const float *y,
                        float *flat_p, float *flat_g, float *flat_m, float *flat_v,
                        const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                        int64_t n_users, int64_t n_items,
                        float *pred, float *se, float *sse_accum, void *ws, size_t ws_bytes,
                        int64_t B, int T, int E, int L,
                        float dropout_p, int training, uint64_t seed, uint64_t offset, float inv_denom,
                        int conv_algo, int token_buffer, int tokens_ready,
                        const int64_t *next_user_idx, const int64_t *next_item_idx,
                        float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                        void *stream);

/* Data parallel: r4r_deepconnpp_step with flat_m == NULL computes gradients only (flat_g; d loss / d pred
 * at r4r_deepconnpp_ws_offset 5); after the exchange -- all-reduce flat_g + r4r_adam_multi, all_gather of
 * the ranks' (uid, iid, d loss / d pred), ids -1 padding ragged shards -- this updates the two ID bias
 * vectors from ALL ranks' ratings (same `ws` and shape arguments as the step).  B_all <= 16384. */
int r4r_deepconnpp_rows_apply(const int64_t *uid_all, const int64_t *iid_all, const float *g_all, int64_t B_all,
                              const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                              int64_t n_users, int64_t n_items, void *ws, size_t ws_bytes,
                              int64_t B, int T, int E, int L, int64_t V,
                              float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                              void *stream);

/* ---- fused native step for TransNet / TransNet++ (TransNet.py:9-122; the three-optimiser step of
 * main.py:26-53 with utils.init_transnet_optim, utils.py:70-92).  Three TextCNN towers (user
 * documents, item documents, the review being rated: user_idx / item_idx / this_idx [B, T]) and
 * three losses from one forward -- target MSE -> target.*, transform loss mean ||s_ir - t_ir||^2 ->
 * source.*, source MSE -> source_fm.* and (TransNet++, plus = 1) the ID vectors.  The gradient each
 * of the reference's three optimisers consumes is that of its own loss on its own parameter group
 * at the pre-step weights, so the step is one backward with three disjoint groups and -- the three
 * Adams sharing lr, weight decay and step count -- one flat Adam (csrc/transnet_engine.hip has the
 * argument; the reference-generated trajectories pin it).
 * Flat layout (22 slots, r4r_transnet_layout): source.user_conv.convs.0.weight, .bias,
 * source.item_conv.convs.0.(same), target.conv.convs.0.(same), source.user_conv.fc.weight, .bias,
 * source.item_conv.fc.(same), target.conv.fc.(same), source.project.0.weight, .bias,
 * source.project.2.weight, .bias, source_fm.V, source_fm.lin.weight, .bias, target.fm.V,
 * target.fm.lin.weight, .bias.  rows_p / rows_m / rows_v: HOST arrays of 2 DEVICE pointers --
 * user_embedding.weight [n_users, 5], item_embedding.weight [n_items, 5] (TransNet++; NULL otherwise).
 * pred / se: the SOURCE prediction and its squared error (what main.py:57 sums); sse_accum (3 floats,
 * nullable) += [sum of se, this batch's mean target SE, this batch's transform loss]; the workspace
 * holds per rating (r4r_transnet_ws_offset which = 3) the target prediction, its squared error and
 * ||s_ir - t_ir||^2.  Dropout draws per rating at Philox counter offset + b*(5L+10) + k:
 * source.user_conv.dropout [L], source.item_conv.dropout [L], target.conv.dropout [L],
 * source.dropout [L], target.dropout [L], dropout.user [5], dropout.item [5]. */
int r4r_transnet_nparam(void);
int r4r_transnet_layout(int E, int L, int plus, int64_t *offsets, int64_t *sizes, int64_t *total);
size_t r4r_transnet_ws_bytes(int64_t B, int T, int E, int L, int plus, int64_t V, int64_t n_users, int64_t n_items);
size_t r4r_transnet_ws_offset(int64_t B, int T, int E, int L, int plus, int64_t V, int64_t n_users, int64_t n_items,
                              int which);   /* 0 dropout multipliers, 1 / 2 ID-vector gradient rows [B,5], 3 aux [B,3], 4 size of the persistent head, 5 the broken-schedule flag (int), 6 + 2*tower + buffer: token counter */
int r4r_transnet_step(const float *table, int64_t V,
                      const int64_t *user_idx, const int64_t *item_idx, const int64_t *this_idx,
                      const int64_t *uid, const int64_t *iid, const float *y,
                      float *flat_p, float *flat_g, float *flat_m, float *flat_v,
                      const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                      int64_t n_users, int64_t n_items,
                      float *pred, float *se, float *sse_accum, void *ws, size_t ws_bytes,
                      int64_t B, int T, int E, int L, int plus,
                      float dropout_p, int training, uint64_t seed, uint64_t offset, float inv_denom,
                      int conv_algo, int token_buffer, int tokens_ready,
                      const int64_t *next_user_idx, const int64_t *next_item_idx, const int64_t *next_this_idx,
                      int sweep_period, int64_t sweep_base, int sweep_all,
                      float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                      void *stream);

/* The ID-vector sweep, temporally blocked on a schedule (TransNet++; the contract is r4r_mf_step's, spelled out
 * there).  torch.optim.Adam moves EVERY row of user_embedding / item_embedding every step (main.py:94-96: weight
 * decay, decaying moments), 24 bytes of traffic per element -- but an element no rating of the batch names goes
 * through the same gradient-zero update whether it is applied now or together with the next few.  With
 * sweep_period P in 2..8 a chunk of 1,024 table elements is visited every P-th step only (chunk c at the steps s with
 * (c % P + c / P + s) % P == 0) and then takes its pending updates at once, in step order, each with its own step's
 * scalars: the same fp32 operations per element as P = 1, a P-th of the bytes.  The ID vectors a rating reads catch
 * up in the head kernel's registers (rows_m / rows_v are read for that, also on a gradients-only step: pass them
 * whenever updates may be pending), a named row's entry wave applies what is missing before its gradient update.
 *   sweep_base / sweep_all: as for r4r_mf_step (the caller keeps the base; it becomes adam_step after sweep_all = 1,
 *   P = 1 or r4r_transnet_rows_flush).  lr, betas, eps, weight_decay do not change while updates are pending;
 *   r4r_transnet_rows_flush (adam_step = the last completed step; same `ws` / shapes as the steps) before anything else
 *   reads or writes the tables or their moments (evaluation through r4r_transnet_step included).  The int at
 *   r4r_transnet_ws_offset(which = 5) is set if more than 8 updates were ever pending (a broken schedule).
 * The data-parallel update launches (r4r_mf_apply, r4r_transnet_rows_apply, r4r_idnet_rows_apply) run the same
 * schedule over the gathered entries -- the same (P, base, all) on every rank. */
int r4r_transnet_rows_flush(const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                            int64_t n_users, int64_t n_items, void *ws, size_t ws_bytes,
                            int64_t B, int T, int E, int L, int64_t V,
                            int sweep_period, int64_t sweep_base,
                            float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                            void *stream);

/* Data parallel: call r4r_transnet_step with flat_m == NULL (gradients only: flat_g and, TransNet++,
 * the compact ID-vector rows at r4r_transnet_ws_offset 1 / 2), exchange -- all-reduce flat_g and apply
 * r4r_adam_multi; all_gather the ranks' (uid, iid, gradient rows), ids -1 padding ragged shards --
 * then update the ID-vector tables from ALL ranks' rows (same `ws` and shape arguments as the step:
 * the row tags live there).  B_all <= 32768.  sweep_period / sweep_base / sweep_all: the schedule of the sweep, the
 * same on every rank (contract above; r4r_transnet_rows_flush on every rank). */
int r4r_transnet_rows_apply(const int64_t *uid_all, const int64_t *iid_all, const float *gu_all, const float *gi_all,
                            int sweep_period, int64_t sweep_base, int sweep_all,
                            int64_t B_all, const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                            int64_t n_users, int64_t n_items, void *ws, size_t ws_bytes,
                            int64_t B, int T, int E, int L, int64_t V,
                            float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                            void *stream);
/* The same update with one exchange and no glue launches (world * B_pad <= 2,048 gathered ratings):
 * r4r_transnet_dp_block packs this rank's entries of the LAST gradients-only r4r_transnet_step on (uid, iid, ws)
 *   into `block` (r4r_transnet_dp_block_bytes(B_pad) bytes: ids as int32 + the two gradient rows; entries past B carry
 *   id -1), the caller all_gathers the blocks (rank order) and every rank calls
 * r4r_transnet_rows_apply_blocks(blocks [world], ...): the scheduled sweep straight over the blocks, ONE launch. */
size_t r4r_transnet_dp_block_bytes(int64_t B_pad);
int r4r_transnet_dp_block(const int64_t *uid, const int64_t *iid, void *ws, size_t ws_bytes, int64_t B, int T,
                          int E, int L, int64_t V, int64_t n_users, int64_t n_items, void *block, int64_t B_pad,
                          void *stream);
int r4r_transnet_rows_apply_blocks(const void *blocks, int world, int64_t B_pad, int sweep_period, int64_t sweep_base,
                                   int sweep_all, const uint64_t *rows_p, const uint64_t *rows_m,
                                   const uint64_t *rows_v, int64_t n_users, int64_t n_items, void *ws, size_t ws_bytes,
                                   int64_t B, int T, int E, int L, int64_t V,
                                   float lr, double beta1, double beta2, float eps, float weight_decay,
                                   int64_t adam_step, void *stream);

/* ------------------------------------------------------------------------
 * Fused native step for the ID-only recommenders with dense layers: model_type 'MF' (MF.py:60-68) and
 * the NeuMF family (NeuMF.py: GMF :10-36, MLP :38-72, NeuMF :74-143; main.py:289-340 trains them in
 * three stages).  Replaces, per training step, the model's forward, MSELoss (loss.py:7-11),
 * loss.backward() and torch.optim.Adam.step() (main.py:56-60,94-96) by 3 launches (4 for NeuMF): a head
 * kernel per rating (forward + backward, compact ID-table gradient rows), the dense-gradient column sums
 * + Adam, and the tagged Adam sweeps over the ID tables and bias vectors (every row moves every step;
 * no dense table gradient is materialised).
 *   variant    0 MF   1 GMF   2 MLP   3 NeuMF
 *   flat_*     the dense parameters in r4r_idnet_layout order (8 slots; absent ones have size 0):
 *              projection/project .1.weight [L,2L], .1.bias, .3.weight [L,L], .3.bias, final.V [2L,L]
 *              (MF), final.lin.weight | final.weight [1, 2L or L], its bias, global_bias
 *   rows_*     HOST arrays of 6 device pointers: user / item table of the first pair (MF, GMF, MLP:
 *              user_embedding, item_embedding; NeuMF: gmf_*), of the second pair (NeuMF: mlp_*; else
 *              unused), user_bias, item_bias -- parameters, and their Adam moments
 *   flat_g == NULL: eval-mode forward only.  L <= 32, B <= 16384 for training steps. */
int r4r_idnet_nparam(void);
int r4r_idnet_layout(int variant, int L, int64_t *offsets, int64_t *sizes, int64_t *total);
size_t r4r_idnet_ws_bytes(int variant, int64_t B, int L, int64_t n_users, int64_t n_items);
size_t r4r_idnet_ws_offset(int variant, int64_t B, int L, int64_t n_users, int64_t n_items,
                           int which);   /* 0 dropout multipliers [B, draws], 1 d loss/d pred [B], 2 size of the persistent head (row tags), 3 the broken-schedule flag (int), 4 + 2*pair + side: compact gradient rows [B, L] */
int r4r_idnet_step(int variant, const int64_t *uid, const int64_t *iid, const float *y,
                   float *flat_p, float *flat_g, float *flat_m, float *flat_v,
                   const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                   int64_t n_users, int64_t n_items,
                   float *pred, float *se, float *sse_accum, void *ws, size_t ws_bytes,
                   int64_t B, int L, float dropout_p, int training, uint64_t seed, uint64_t offset,
                   float inv_denom, int sweep_period, int64_t sweep_base, int sweep_all,
                   float lr, double beta1, double beta2, float eps, float weight_decay,
                   int64_t adam_step, void *stream);
/* sweep_period / sweep_base / sweep_all: the sweeps over the variant's table pair(s), temporally blocked on the
 * schedule of r4r_mf_step (contract there and at r4r_transnet_rows_flush; rows_m / rows_v are read by the head kernel
 * for the catch-up of the rows a rating names, also on a gradients-only step; r4r_idnet_ws_offset which = 3: the
 * broken-schedule flag).  r4r_idnet_rows_flush applies what is pending; adam_step = the last completed step. */
int r4r_idnet_rows_flush(int variant, const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                         int64_t n_users, int64_t n_items, void *ws, size_t ws_bytes, int64_t B, int L,
                         int sweep_period, int64_t sweep_base,
                         float lr, double beta1, double beta2, float eps, float weight_decay, int64_t adam_step,
                         void *stream);

/* Data parallel: r4r_idnet_step with flat_m == NULL computes gradients only (flat_g, and in the workspace
 * d loss/d pred [B] and the compact rows [B, L] per table: r4r_idnet_ws_offset 1, 4..7); after the exchange --
 * all-reduce flat_g + r4r_adam_multi; all_gather of (uid, iid, d loss/d pred, rows), ids -1 padding ragged
 * shards -- the ID tables and bias vectors are updated from ALL ranks' rows.  gu_all / gi_all: HOST arrays of
 * 2 device pointers ([B_all, L] rows of the first / second table pair).  `ws`, B: the step's own.  B_all <= 16384.
 * sweep_period / sweep_base / sweep_all: as for r4r_transnet_rows_apply. */
int r4r_idnet_rows_apply(int variant, const int64_t *uid_all, const int64_t *iid_all, const float *g_all,
                         const uint64_t *gu_all, const uint64_t *gi_all,
                         int sweep_period, int64_t sweep_base, int sweep_all,
                         int64_t B_all,
                         const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                         int64_t n_users, int64_t n_items, void *ws, size_t ws_bytes, int64_t B, int L,
                         float lr, double beta1, double beta2, float eps, float weight_decay,
                         int64_t adam_step, void *stream);

/* ------------------------------------------------------------------------
 * Device-side batch construction (the loader side of the path).
 * Replaces  data.DataLoader.remove_overlap / pad_and_join / pad_only and the 10-wide neighbour
 *           padding                                   data.py:144-236, 273-279, 375-447
 *           the per-batch LongTensor constructions    data.py:293-301, data_fast.py:101-109
 * The reviews of a dataset live in HBM as token pools (one per side; reviews4rec_amd/data.py):
 *   tok [tokens] int32, rev_off [reviews + 1] token range of a review, first [owners + 1] review
 *   range of a user / item, nb [reviews] the item / user a review is about (u_to_i_map / i_to_u_map)
 *   held_tok / held_off: the held-out reviews (test_reviews) as a pool of single reviews
 * For n ratings -- u, i: ids; nb_item: the item whose reviewer list is reported (== i except in
 * iter_negs, data.py:398); ku / ki: index of the rating's own review among the user's / item's
 * reviews, or -1 (nothing removed); held: index into the held pool or -1 ([0], data.py:245); train:
 * 1 = the own review is the user's review ku (data.py:221) -- ONE launch writes, as one int64 block,
 *   [n, doc] own review | [n, 10] users who reviewed the item (pad_user) | [n, 10] items the user
 *   reviewed (pad_item) | [n, doc] user document | [n, doc] item document
 * with doc = T (documents: reviews concatenated, cut / zero padded, R == 0) or R * W (NARRE: R
 * reviews of W words). */
int r4r_batch_build(const int32_t *user_tok, const int64_t *user_rev_off, const int64_t *user_first,
                    const int64_t *user_nb, const int32_t *item_tok, const int64_t *item_rev_off,
                    const int64_t *item_first, const int64_t *item_nb, const int32_t *held_tok,
                    const int64_t *held_off, const int64_t *u, const int64_t *i, const int64_t *nb_item,
                    const int64_t *ku, const int64_t *ki, const int64_t *held, int train, int64_t *out,
                    int64_t n, int T, int R, int W, int64_t pad_user, int64_t pad_item, void *stream);

/* ------------------------------------------------------------------------
 * Spans: K training steps per host call (csrc/span.hip).
 * Replaces  the body of the epoch loop, K iterations at a time        main.py:23-60
 *             for data, y in reader.iter(): zero_grad -> model(data) -> loss -> backward -> optimizer.step()
 *           with the batcher's slices                                  data_fast.py:99-109, data.py:250-372
 * for K consecutive FULL batches of an epoch: r4r_batch_build for a GROUP of batches per launch into a ring
 * of two groups, then the family's native step per batch, everything enqueued on `stream` from C.  The kernels are
 * those of the per-step entry points with the same arguments -- a span and K single steps give the same bits --; what
 * changes between two steps (batch pointers, adam_step, the dropout stream position, the token buffer, the sweep
 * schedule) advances here exactly as the per-step caller advances it.
 *
 * `loader`: HOST array of R4R_LOADER_WORDS uint64 describing the epoch (reviews4rec_amd/data.py builds it):
 *    0..3  user pool tok, rev_off, first, nb     4..7  item pool     8, 9  held_tok, held_off   (r4r_batch_build's)
 *   10..15 u, i, ku, ki, held (int64 [N]), y (float [N]): the epoch's ratings in loader order
 *   16 train  17 T  18 R  19 W  20 pad_user  21 pad_item                                          (r4r_batch_build's)
 *   22 ring: int64 device buffer of 2 groups, 0 = an ids-only loader (iter_simple: words 0..9, 12..14, 16..23 unused)
 *   23 ring stride per group (int64 elements, >= G * B * (3 doc + 20))   24 N   25 G batches per group   26 B
 *   27 0, or a HOST table [G][8] of the device pointers (r4r_span_batch's order) of G batches of B ratings that are
 *      already built and resident: batch b is entry b % G, nothing is constructed (a pool of batches cycled through;
 *      words 0..23 unused)
 * Batch b = ratings [b B, (b + 1) B); only the N / B full batches are reachable (a ragged tail goes through the
 * per-step entry).  built_group (host, in / out): the highest group whose block is in the ring, -1 before the
 * epoch's first span; groups are built in order as the steps reach them (the next one before the current one's last
 * step).  announce: a full batch follows the span and the last step announces it (its token marks ride on that
 * step's backward launch: pass tokens_ready = 1, token_buffer = the other buffer to whatever trains on it next).
 * steps_done (host, out, nullable): steps enqueued before an error.  token_buffer / tokens_ready / offset /
 * adam_step / sweep_*: the FIRST step's; step k runs with token_buffer ^ (k & 1), tokens_ready = 1 (k > 0),
 * offset + k * draws_per_step, adam_step + k.  sweep_period / sweep_want: the schedule in force and the one wanted
 * (engine.py _SweepSchedule: a step visits every chunk iff they differ or the period is 1, and the wanted one is in
 * force afterwards); *sweep_base / *sweep_period_out carry the schedule across calls.  Training steps only
 * (flat_g, moments and adam_step >= 1 required); all other arguments as in the family's r4r_*_step.
 * r4r_span_build: make sure the group of `batch` is in the ring.  r4r_span_batch: the eight device pointers of a
 * built batch -- this, who, what, user doc, item doc, uid, iid, y (data_fast.py:101-109 order) -- into HOST slots[8]. */
#define R4R_LOADER_WORDS 28
int r4r_span_build(const uint64_t *loader, int64_t batch, int64_t *built_group, void *stream);
int r4r_span_batch(const uint64_t *loader, int64_t batch, uint64_t *slots);
int r4r_deepconn_span(const uint64_t *loader, int64_t first_batch, int64_t steps, int announce,
                      int64_t *built_group, int64_t *steps_done, const float *table, int64_t V, float *flat_p,
                      float *flat_g, float *pred, float *se, float *sse_accum, void *ws, size_t ws_bytes, int T,
                      int E, int L, float dropout_p, int training, uint64_t seed, uint64_t offset,
                      uint64_t draws_per_step, float inv_denom, int conv_algo, int token_buffer, int tokens_ready,
                      float *flat_m, float *flat_v, float lr, double beta1, double beta2, float eps,
                      float weight_decay, int64_t adam_step, void *stream);
/* Data parallel: r4r_deepconn_span with the gradient exchange between every step's gradients and its
 * Adam update -- step k = r4r_deepconn_step (gradients only, inv_denom = 1 / B_global) -> the collective -> the update,
 * everything enqueued from C.  exchange 0: `collective` = the address of RCCL's ncclAllReduce (in place on flat_g,
 * float32 sum), then r4r_adam_multi; exchange 1: ncclAllGather of flat_g into `gathered` [world][total], then
 * r4r_adam_gathered (rank order).  `collective` / `comm`: resolved and built by the caller in the RCCL library it
 * already holds (reviews4rec_amd/dist.py: StreamRccl); total = the flat layout's element count.  Replaces, per step,
 * the reference loop body of main.py:26-60 on every rank + the gradient sum the reference (single-process,
 * main.py:407) does not have. */
int r4r_deepconn_span_dp(const uint64_t *loader, int64_t first_batch, int64_t steps, int announce,
                         int64_t *built_group, int64_t *steps_done, const float *table, int64_t V, float *flat_p,
                         float *flat_g, float *pred, float *se, float *sse_accum, void *ws, size_t ws_bytes, int T,
                         int E, int L, float dropout_p, int training, uint64_t seed, uint64_t offset,
                         uint64_t draws_per_step, float inv_denom, int conv_algo, int token_buffer, int tokens_ready,
                         float *flat_m, float *flat_v, int64_t total, float lr, double beta1, double beta2, float eps,
                         float weight_decay, int64_t adam_step, int exchange, void *collective, void *comm, int world,
                         float *gathered, void *stream);
int r4r_deepconnpp_span(const uint64_t *loader, int64_t first_batch, int64_t steps, int announce,
                        int64_t *built_group, int64_t *steps_done, const float *table, int64_t V, float *flat_p,
                        float *flat_g, float *flat_m, float *flat_v, const uint64_t *rows_p,
                        const uint64_t *rows_m, const uint64_t *rows_v, int64_t n_users, int64_t n_items,
                        float *pred, float *se, float *sse_accum, void *ws, size_t ws_bytes, int T, int E, int L,
                        float dropout_p, int training, uint64_t seed, uint64_t offset, uint64_t draws_per_step,
                        float inv_denom, int conv_algo, int token_buffer, int tokens_ready, float lr, double beta1,
                        double beta2, float eps, float weight_decay, int64_t adam_step, void *stream);
int r4r_narre_span(const uint64_t *loader, int64_t first_batch, int64_t steps, int announce, int64_t *built_group,
                   int64_t *steps_done, const float *table, int64_t V, float *flat_p, float *flat_g, float *flat_m,
                   float *flat_v, const uint64_t *rows_p, const uint64_t *rows_m, const uint64_t *rows_v,
                   int64_t n_users, int64_t n_items, float *pred, float *se, float *sse_accum, void *ws,
                   size_t ws_bytes, int R, int T, int E, int L, float dropout_p, int training, uint64_t seed,
                   uint64_t offset, uint64_t draws_per_step, float inv_denom, int conv_algo, int token_buffer,
                   int tokens_ready, float lr, double beta1, double beta2, float eps, float weight_decay,
                   int64_t adam_step, void *stream);
int r4r_transnet_span(const uint64_t *loader, int64_t first_batch, int64_t steps, int announce,
                      int64_t *built_group, int64_t *steps_done, const float *table, int64_t V, float *flat_p,
                      float *flat_g, float *flat_m, float *flat_v, const uint64_t *rows_p, const uint64_t *rows_m,
                      const uint64_t *rows_v, int64_t n_users, int64_t n_items, float *pred, float *se,
                      float *sse_accum, void *ws, size_t ws_bytes, int T, int E, int L, int plus, float dropout_p,
                      int training, uint64_t seed, uint64_t offset, uint64_t draws_per_step, float inv_denom,
                      int conv_algo, int token_buffer, int tokens_ready, int sweep_period, int sweep_want,
                      int64_t *sweep_base, int *sweep_period_out, float lr, double beta1, double beta2, float eps,
                      float weight_decay, int64_t adam_step, void *stream);
int r4r_mf_span(const uint64_t *loader, int64_t first_batch, int64_t steps, int64_t *steps_done, const uint64_t *p,
                const uint64_t *m, const uint64_t *v, int64_t n_users, int64_t n_items, int D, float *pred,
                float *se, float *sse_accum, void *ws, size_t ws_bytes, float dropout_p, int training,
                uint64_t seed, uint64_t offset, uint64_t draws_per_step, float inv_denom, int sweep_period,
                int sweep_want, int64_t *sweep_base, int *sweep_period_out, float lr, double beta1, double beta2,
                float eps, float weight_decay, int64_t adam_step, void *stream);
int r4r_idnet_span(const uint64_t *loader, int64_t first_batch, int64_t steps, int64_t *steps_done, int variant,
                   float *flat_p, float *flat_g, float *flat_m, float *flat_v, const uint64_t *rows_p,
                   const uint64_t *rows_m, const uint64_t *rows_v, int64_t n_users, int64_t n_items, float *pred,
                   float *se, float *sse_accum, void *ws, size_t ws_bytes, int L, float dropout_p, int training,
                   uint64_t seed, uint64_t offset, uint64_t draws_per_step, float inv_denom, int sweep_period,
                   int sweep_want, int64_t *sweep_base, int *sweep_period_out, float lr, double beta1,
                   double beta2, float eps, float weight_decay, int64_t adam_step, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* R4R_H */
