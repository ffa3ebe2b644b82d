// Internal launch interface of the TextCNN tower kernels (textcnn.hip), shared with
// the fused model steps (engine.hip).  Up to 4 towers run in ONE grid (blockIdx.y).
#pragma once
#include "common.h"

namespace r4r {

constexpr int NP = 112;          // filters padded to 7 MFMA column tiles of 16
constexpr int MAX_TOWERS = 4;

struct FwdTower {
    const int64_t *idx;          // [N, T] token ids
    const float *conv_w;         // [F, 3, E]
    const float *conv_b;         // [F]
    float *wp;                   // packed weight image, textcnn_wp_floats(E) floats
    float *pmax;                 // [N, tiles, NP] per-tile running max (pre-relu)
    int *parg;                   // [N, tiles, NP] per-tile first argmax
};

struct WgradTower {
    const int64_t *idx;          // [N, T]
    const float *g_pooled;       // [N, F]
    const int *argmax;           // [N, F]
    float *part_w;               // [nsplit, F, 3E]
    float *part_b;               // [nsplit, F]
    float *d_w;                  // [F, 3, E]   (overwritten)
    float *d_b;                  // [F]
};

// project-then-gather path (project.hip)
struct ProjTower {
    const int64_t *idx;          // [N, T]
    const float *conv_w;         // [F, 3, E]
    const float *conv_b;         // [F]
    int *flags;                  // [V rounded up to 4] token-used marks; all-zero on entry, left all-zero
    int *slot;                   // [V]   token -> dense row of ptab, -1 if unused
    int *list;                   // [cap] dense row -> token
    int *count;                  // [1]   number of distinct tokens
    float *ptab;                 // [cap, 300] projected rows: 3 taps x 100 filters
    float *pmax;                 // [N, proj_tiles(T), NP]
    int *parg;
    float *wimg = nullptr;       // optional scratch of textcnn_wp_floats(E) floats (the direct conv's weight-image region):
                                 // the fp16-split GEMM packs its B operand there; absent -> fp32 GEMM
};
int proj_tiles(int T);
int64_t proj_row_capacity(int64_t N, int T, int64_t V);
size_t proj_ptab_floats(int64_t N, int T, int64_t V);
// zero_state: memset flags + count first (callers whose workspace is not persistently zeroed)
int textcnn_proj_fwd_launch(const float *table, int64_t V, const ProjTower *tw, int ntower,
                            int64_t N, int T, int E, int F, bool zero_state, hipStream_t st);
int textcnn_proj_tokens_launch(int64_t V, const ProjTower *tw, int ntower, int64_t N, int T,
                               bool zero_state, hipStream_t st);
int textcnn_proj_compute_launch(const float *table, int64_t V, const ProjTower *tw, int ntower,
                                int64_t N, int T, int E, int F, hipStream_t st);
void proj_gemm_set_math(int mode, float table_maxabs, float weight_maxabs);
int proj_gemm_math_mode();   // 0: fp32 MFMA (default), 1: fp16-split operands, fp32 accumulate
void proj_gemm_set_form(int balanced);          // 1: balanced 7-row-tile form where it applies (default), 0: tile form, 2: tile form with whole tiles only, -1: env
// R4R_CONV_AUTO / _DIRECT / _PROJECT (include/r4r.h) -> the algorithm to run; honours R4R_CONV_ALGO
int textcnn_pick_algo(int requested, int64_t N, int T, int E, int F);

size_t textcnn_wp_floats(int E);
int textcnn_tile_rows(int T);                  // conv positions per workgroup tile chosen for T
int textcnn_tiles(int T);                      // tiles per document
int textcnn_wgrad_splits(int64_t N);

// pack the weight images, then the MFMA tile kernel over all towers; partials land in pmax/parg
int textcnn_fwd_launch(const float *table, const FwdTower *tw, int ntower,
                       int64_t N, int T, int E, int F, hipStream_t st);
// pooled = max(0, max over tiles), argmax = first position or -1
int textcnn_pool_finish_launch(const float *pmax, const int *parg, float *pooled, int *argmax,
                               int64_t N, int tiles, int F, hipStream_t st);
int textcnn_wgrad_launch(const float *table, const WgradTower *tw, int ntower,
                         int64_t N, int T, int E, int F, hipStream_t st, int64_t table_bytes = 0);

// second stage of the wgrad: add the nsplit partials of every tower in a fixed order
int textcnn_wgrad_reduce_launch(const WgradTower *tw, int ntower, int64_t N, int E, int F, hipStream_t st);

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace r4r
