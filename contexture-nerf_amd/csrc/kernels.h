// Internal (non-ABI) host entry points shared between the translation units of libctxnerf.so.
#pragma once
#include "common.h"

struct GemmArgs {
    const f16 *X;
    const f16 *Wt;
    const f16 *bias;       // [N] or null (GEGLU: packed/interleaved order)
    const f16 *rowbias;    // [M/rows_per_batch, N] or null
    const f16 *residual;   // [M, ldr] or null
    f16 *out;
    int M, N, K;
    int ldc, ldr;
    int rows_per_batch;
    int ldrb;              // row stride of rowbias
    int epi;               // 0 plain, 1 GEGLU (out has N/2 columns)
    // conv
    int H, W, Cin, Ho, Wo, stride, ups;
    int poff;              // conv input-coordinate offset: 0 = symmetric padding 1; 1 = no top/left padding (diffusers' VAE
                           // Downsample2D: F.pad(x, (0,1,0,1)) then a stride-2 conv with padding 0)
    int res32, out32;      // 1: `residual` / `out` point at fp32 tensors (the fp32 residual-stream experiment; gemm.hip epilogues only)
    int zins;              // conv: 1 = with ups, the 2x grid is ZERO-INSERTED (odd rows / columns read the zero page) instead of
                           // nearest-upsampled: the input gradient of a stride-2 convolution (poff = -1 there); gemm.hip kernel only
    int ntm, ntn;
    int splitk;            // >1: grid.y = splitk, fp32 partials [splitk][M][N] go to `part`
    float *part;
    int pk;                // K-stage depth chosen by the launcher (32 or 64)
    int krot;              // 1: per-workgroup K rotation (spreads concurrent accesses to shared operand rows); 2: no rotation and no
                           // steady-state K loop (CTX_GEMM_STEADY=0, the A/B switch of gemm.hip's branch-free loop)
    int tile, use8;        // plan: tile id of gemm.hip (-1 = heuristic); use8: -1 heuristic, 0 gemm.hip, 1 gemm8.hip, 2 / 3 conv_halo.hip (128 / 64 features), 4 .. 8 gemm144.hip (6 waves; 15 waves lockstep; pipelined; barrier per two stages; 288 x 160 lockstep)
    int stage_epi;         // 1: epilogue staged through LDS (whole-line stores / residual reads)
    int mfast;             // 1: consecutive workgroups walk M first (share the weight panel in their XCD's L2)
};

int ctx_gemm_dispatch(GemmArgs &a, bool conv, hipStream_t s);
// 256x256 8-wave kernel (gemm8.hip): launches and returns 1 when the problem suits it, else 0
int ctx_gemm8_try(GemmArgs &a, bool conv, bool force, hipStream_t s);
// halo-staged 3x3 convolution (conv_halo.hip), ni = 1 | 2 (64 / 128 features per workgroup): 1 when launched
int ctx_conv_halo_try(GemmArgs &a, int ni, hipStream_t s);
// 144x160 kernel (gemm144.hip; plan value use8 = 4: 6 waves, 5: 15 waves): 1 when launched
int ctx_gemm144_try(GemmArgs &a, bool conv, int form, hipStream_t s);
// split-K factor the heuristic would use for this problem (1 = none); caller provides a.part = S*M*N floats
int ctx_gemm_pick_split(int M, int N, int K, int epi);
// fills a.splitk / a.tile / a.use8 for a fully described problem: the tuned table (gemm_tuned.h) first, heuristics otherwise
void ctx_gemm_plan(GemmArgs &a, bool conv);
int ctx_gemv_f16(const f16 *x, const f16 *w, const f16 *bias, int Bm, int N, int K, int silu_in, int silu_out, f16 *out, hipStream_t s);
int ctx_concat_f16(const f16 *a, const f16 *b, int64_t M, int Ca, int Cb, f16 *y, hipStream_t s);
int ctx_transpose_v_f16(const f16 *v, int B, int S, int ld, int heads, int Sp, int perm, f16 *vt, hipStream_t s);
int ctx_f32_to_f16(const float *x, int64_t n, f16 *y, hipStream_t s);
int ctx_time_embed_f16(const float *t, int B, int dim, f16 *out, hipStream_t s);
int ctx_conv_in_f16(const float *x, const f16 *w, const f16 *bias, int B, int Cin, int H, int W, int Cout, f16 *y, hipStream_t s);
int ctx_conv_out_f16(const f16 *x, const f16 *w, const f16 *bias, int B, int H, int W, int C, int Cout, float *out, hipStream_t s);
int ctx_attention_core(const f16 *Q, const f16 *K, const f16 *V, int B, int Sq, int Skv, int heads, int q_stride,
                       int kv_stride, float scale, f16 *O, int o_stride, hipStream_t s);
extern "C" int64_t ctx_groupnorm_ws_bytes(int32_t B, int32_t groups);
// norms over the fp32 residual stream (x32 != 0) or fp16 activations; output fp16
int ctx_groupnorm_any(const void *x, int x32, const void *gamma, const void *beta, int B, int HW, int C, int groups, float eps, int silu,
                      void *y, void *stats_ws, hipStream_t stream);
int ctx_layernorm_any(const void *x, int x32, const void *gamma, const void *beta, int64_t rows, int C, float eps, void *y, hipStream_t stream);
int ctx_concat_f32(const float *a, const float *b, int64_t M, int Ca, int Cb, float *y, hipStream_t s);
int ctx_f16_to_f32(const f16 *x, int64_t n, float *y, hipStream_t s);

// profiling hooks (api.cpp): when on, MFMA kernels are launched with hipExtLaunchKernelGGL start/stop events
bool ctx_prof_on(void);
void ctx_prof_events(int klass, hipEvent_t *a, hipEvent_t *b);
