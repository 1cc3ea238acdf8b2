// Normalisation / elementwise kernels of the UNet denoiser (NHWC fp16 activations, fp32 statistics).
// HBM-bound: every access is a 16-byte (8 x f16) vector per lane.
//   GroupNorm(+SiLU)   diffusers ResnetBlock2D norm1/norm2, Transformer2DModel.norm, conv_norm_out
//   LayerNorm          BasicTransformerBlock norm1/2/3
//   GEGLU, channel concat, V transpose, layout converters, timestep embedding, conv_in / conv_out,
//   CFG + PLMS scheduler step (src/stable_diffusion_depth.py:428-430,514).
#include "common.h"
#include "kernels.h"
#include <math.h>

// ------------------------------------------------------------------------------------------------
// GroupNorm.  Both passes are latency-bound at the UNet's sizes (a 12 MB tensor is ~2 HBM latencies deep), so the
// structure is "every load of a thread in flight at once":
// Pass 1: grid (NS, B), block = C/8 chunk columns x PL pixel lanes (~1000 threads); a thread owns 8 channels and
// walks its split GN_U pixels at a time, each batch loaded before its first add; per-channel partials -> LDS -> per-group
// sums.  NS by sample size (16-32 for the UNet's tensors, 128 for the VAE's: see the launcher).
// Pass 2: grid (nb, B), block = C/8 chunk columns x >= 256/(C/8) pixel lanes; a thread keeps ONE column (scale / shift in
// registers), issues its GN_AU 16-byte loads first, then the block folds the NS split partials per (b, group) in a fixed
// order (deterministic: no float atomics anywhere in GroupNorm) while they fly.
// 8 consecutive channels of the input as floats: fp16 activations (16 bytes) or the fp32 residual stream (32 bytes)
struct F8 { float v[8]; };
template <bool X32>
__device__ __forceinline__ F8 ld8(const void *base, size_t elem)
{
    F8 r;
    if (X32) {
        const f32x4 a = *(const f32x4 *)((const float *)base + elem), b = *(const f32x4 *)((const float *)base + elem + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { r.v[j] = a[j]; r.v[4 + j] = b[j]; }
    } else {
        const f16x8 a = *(const f16x8 *)((const f16 *)base + elem);
#pragma unroll
        for (int j = 0; j < 8; ++j) r.v[j] = (float)a[j];
    }
    return r;
}

#define GN_MAX_GROUPS 64
#define GN_MAX_SPLITS 128
#define GN_MAX_C 4096
#define GN_U 4
#define GN_AU 6
#define GN_AU32 6
#define GN_FOLD 8

template <bool X32>
__global__ __launch_bounds__(1024) void k_gn_stats(const void *__restrict__ x, int HW, int C, int G, int NS, int PL,
                                                   float *__restrict__ part)
{
    extern __shared__ float s_part[];          // [PL][C][2]
    const int c8n = C / 8;
    const int b = blockIdx.y, sp = blockIdx.x;
    const int c8 = threadIdx.x % c8n, pl = threadIdx.x / c8n;
    const int per = (HW + NS - 1) / NS;
    const int p0 = sp * per, p1 = min(HW, p0 + per);
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
    const size_t base = ((size_t)b * HW) * C + c8 * 8;
    for (int p = p0 + pl; p < p1; p += PL * GN_U) {
        F8 v[GN_U];
#pragma unroll
        for (int u = 0; u < GN_U; ++u) v[u] = ld8<X32>(x, base + (size_t)min(p + u * PL, p1 - 1) * C);   // unconditional
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            const bool live = p + u * PL < p1;
#pragma unroll
            for (int j = 0; j < 8; ++j) { float f = live ? v[u].v[j] : 0.f; s[j] += f; q[j] += f * f; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s_part[((size_t)pl * C + c8 * 8 + j) * 2 + 0] = s[j];
        s_part[((size_t)pl * C + c8 * 8 + j) * 2 + 1] = q[j];
    }
    __syncthreads();
    // fixed-order fold, two short steps.  A: (channel, slice a of the pixel lanes) -> s_ch[a][C]; B: 8 lanes per group
    // walk that group's na x cg cells, then a shuffle tree.
    float *s_ch = s_part + (size_t)PL * C * 2;
    const int na = max(1, (int)blockDim.x / C);
    for (int it = threadIdx.x; it < C * na; it += blockDim.x) {
        const int a = it / C, c = it - a * C;
        float ss = 0.f, qq = 0.f;
#pragma unroll 4
        for (int l = a; l < PL; l += na) {
            ss += s_part[((size_t)l * C + c) * 2 + 0];
            qq += s_part[((size_t)l * C + c) * 2 + 1];
        }
        s_ch[((size_t)a * C + c) * 2 + 0] = ss;
        s_ch[((size_t)a * C + c) * 2 + 1] = qq;
    }
    __syncthreads();
    const int cg = C / G, cells = na * cg;
    const int octs = (int)blockDim.x >> 3;               // whole 8-lane groups only
    for (int g0 = 0; g0 < G; g0 += octs) {
        const int oc = (int)threadIdx.x >> 3, l = threadIdx.x & 7;
        const int g = oc < octs ? g0 + oc : G;
        float ss = 0.f, qq = 0.f;
        if (g < G)
            for (int k = l; k < cells; k += 8) {
                int a = k / cg, c = g * cg + (k - a * cg);
                ss += s_ch[((size_t)a * C + c) * 2 + 0];
                qq += s_ch[((size_t)a * C + c) * 2 + 1];
            }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) { ss += __shfl_xor(ss, o, 64); qq += __shfl_xor(qq, o, 64); }
        if (g < G && l == 0) {
            part[(((size_t)b * NS + sp) * G + g) * 2 + 0] = ss;
            part[(((size_t)b * NS + sp) * G + g) * 2 + 1] = qq;
        }
    }
}

// 8 consecutive channels as they lie in memory (converted at use: 4 VGPRs per chunk in flight as fp16, 8 as floats).  GN_AU = 6 chunks
// per thread measured best over the UNet's shapes (4 / 6 / 8 / 12: 6.7 / 5.8 / 6.4 / 7.7 us at 2304 x 640 x batch 2, 9.0 / 9.2 / 10.5 / 9.5 at
// 9216 x 320): with 12 a SIMD holds one wave whose ~1.7 us of SiLU arithmetic follows its loads instead of hiding under another wave's
template <bool X32> struct Raw8;
template <> struct Raw8<false> {
    f16x8 a;
    __device__ __forceinline__ float get(int j) const { return (float)a[j]; }
};
template <> struct Raw8<true> {
    f32x4 a, b;
    __device__ __forceinline__ float get(int j) const { return j < 4 ? a[j] : b[j - 4]; }
};
template <bool X32>
__device__ __forceinline__ Raw8<X32> ldraw(const void *base, size_t elem)
{
    Raw8<X32> r;
    if constexpr (X32) { r.a = *(const f32x4 *)((const float *)base + elem); r.b = *(const f32x4 *)((const float *)base + elem + 4); }
    else r.a = *(const f16x8 *)((const f16 *)base + elem);
    return r;
}

// Pass 2.  Block = c8n chunk columns x PL pixel lanes (>= 256 threads); a thread owns ONE column of 8 channels, so its scale and shift
// live in 16 registers, and walks AU pixels of the block's contiguous pixel run, all AU 16-byte loads in flight before anything else
// (the fold of the split partials happens under them).  The form before this one let a thread's column vary with the chunk and
// fetched scale / shift from an LDS table: four ds_read_b128 per chunk at a 32-byte lane stride, i.e. bank-conflicted LDS reads of
// four times the payload — tools/probes/stream_probe.hip: a copy-shaped kernel goes from 3.9 to 8.8 us on 12 MB with exactly that
// table read added, and stays at 4.3 with the values in registers (SiLU and the index arithmetic cost nothing measurable).
template <bool X32>
__global__ __launch_bounds__(512) void k_gn_apply(const void *__restrict__ x, const float *__restrict__ part,
                                                  const f16 *__restrict__ gamma, const f16 *__restrict__ beta, int HW, int C,
                                                  int G, int NS, int PL, float eps, int silu, f16 *__restrict__ y)
{
    constexpr int AU = X32 ? GN_AU32 : GN_AU;
    __shared__ float s_mean[GN_MAX_GROUPS], s_rstd[GN_MAX_GROUPS];
    const int b = blockIdx.y;
    const int c8n = C / 8;
    const int c8 = threadIdx.x % c8n, pl = threadIdx.x / c8n;
    const int p0 = blockIdx.x * (PL * AU);
    const size_t xb = (size_t)b * HW * C + c8 * 8;
    f16 *yb = y + (size_t)b * HW * C + c8 * 8;
    Raw8<X32> v[AU];
#pragma unroll
    for (int u = 0; u < AU; ++u) v[u] = ldraw<X32>(x, xb + (size_t)min(p0 + u * PL + pl, HW - 1) * C);    // unconditional (clamped)
    const f16x8 ga = *(const f16x8 *)(gamma + c8 * 8), be = *(const f16x8 *)(beta + c8 * 8);
    if (threadIdx.x < 256) {
        const int lpg = min(256 / G, 64);                                   // lanes per group: a power of two inside one wave (G < 4: idle lanes)
        const int gi = threadIdx.x / lpg, l = threadIdx.x % lpg;
        const int g = min(gi, G - 1);
        float s = 0.f, q = 0.f;
        for (int k0 = 0; k0 < NS; k0 += GN_FOLD * lpg) {                    // GN_FOLD independent loads per trip, fixed order
            float2 pv[GN_FOLD];
#pragma unroll
            for (int k = 0; k < GN_FOLD; ++k)
                pv[k] = *(const float2 *)(part + (((size_t)b * NS + min(k0 + k * lpg + l, NS - 1)) * G + g) * 2);
#pragma unroll
            for (int k = 0; k < GN_FOLD; ++k)
                if (k0 + k * lpg + l < NS) { s += pv[k].x; q += pv[k].y; }
        }
        for (int o = lpg >> 1; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        if (l == 0 && gi < G) {
            float n = (float)HW * (float)(C / G);
            float mean = s / n;
            float var = fmaxf(q / n - mean * mean, 0.f);
            s_mean[g] = mean;
            s_rstd[g] = rsqrtf(var + eps);
        }
    }
    __syncthreads();
    const int cg = C / G;
    float sa[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int gg = (c8 * 8 + j) / cg;
        sa[j] = s_rstd[gg] * (float)ga[j];
        sh[j] = (float)be[j] - s_mean[gg] * sa[j];
    }
#pragma unroll
    for (int u = 0; u < AU; ++u) {
        const int p = p0 + u * PL + pl;
        if (p < HW) {
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = v[u].get(j) * sa[j] + sh[j];
                if (silu) t = t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t));
                o[j] = (f16)t;
            }
            *(f16x8 *)(yb + (size_t)p * C) = o;
        }
    }
}

// Small-tensor GroupNorm in ONE kernel: when a group's channels are whole 16-byte chunks (C/G % 8 == 0) and one
// (batch, group) slab fits a workgroup's registers (<= GN_FT x GN_FU chunks), a workgroup loads its slab once, reduces
// it (fixed order: lane partials -> DPP wave sums -> 8 wave partials), and writes the normalised slab from registers.
// At the UNet's two deepest levels this replaces two latency-bound launches and one re-read of the tensor.
#define GN_FT 512
#define GN_FU 12
template <bool X32>
__global__ __launch_bounds__(GN_FT) void k_gn_fused(const void *__restrict__ x, const f16 *__restrict__ gamma,
                                                    const f16 *__restrict__ beta, int HW, int C, int G, int PLF, float eps, int silu,
                                                    f16 *__restrict__ y)
{
    // threads = cpg chunk columns x PLF pixel lanes: a thread keeps one column, so gamma / beta are 16 registers (an LDS table read per
    // chunk at a 32-byte lane stride is bank-conflicted and was the slowest part of the apply kernels: tools/probes/stream_probe.hip)
    __shared__ float s_red[2][GN_FT / 64];
    const int g = blockIdx.x, b = blockIdx.y;
    const int cg = C / G, cpg = cg / 8;                // chunks per pixel in this group
    const int c = threadIdx.x % cpg, pl = threadIdx.x / cpg;
    const bool act = pl < PLF;                         // blockDim.x is rounded up to whole waves
    const size_t xb = (size_t)b * HW * C + g * cg + c * 8;
    f16 *yb = y + (size_t)b * HW * C + g * cg + c * 8;
    const f16x8 ga = *(const f16x8 *)(gamma + g * cg + c * 8), be = *(const f16x8 *)(beta + g * cg + c * 8);
    F8 v[GN_FU];
#pragma unroll
    for (int u = 0; u < GN_FU; ++u) v[u] = ld8<X32>(x, xb + (size_t)min(pl + PLF * u, HW - 1) * C);     // unconditional (clamped), all in flight
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int u = 0; u < GN_FU; ++u)
        if (act && pl + PLF * u < HW) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { float f = v[u].v[j]; s += f; q += f * f; }
        }
    s = wave_sum_dpp(s); q = wave_sum_dpp(q);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_red[0][wave] = s; s_red[1][wave] = q; }
    __syncthreads();
    float ss = 0.f, qq = 0.f;
    const int nw = ((int)blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; ++w) { ss += s_red[0][w]; qq += s_red[1][w]; }
    const float n = (float)HW * (float)cg;
    const float mean = ss / n;
    const float rstd = rsqrtf(fmaxf(qq / n - mean * mean, 0.f) + eps);
    float sa[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sa[j] = rstd * (float)ga[j]; sh[j] = (float)be[j] - mean * sa[j]; }
#pragma unroll
    for (int u = 0; u < GN_FU; ++u) {
        const int p = pl + PLF * u;
        if (act && p < HW) {
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = v[u].v[j] * sa[j] + sh[j];
                if (silu) t = t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t));
                o[j] = (f16)t;
            }
            *(f16x8 *)(yb + (size_t)p * C) = o;
        }
    }
}

extern "C" int64_t ctx_groupnorm_ws_bytes(int32_t B, int32_t groups)
{
    return ((int64_t)B * GN_MAX_SPLITS * groups * 2 + (int64_t)B * 2 * GN_MAX_C) * 4;
}

extern "C" int32_t ctx_groupnorm_f16(const void *x, const void *gamma, const void *beta, int32_t B, int32_t HW, int32_t C,
                                     int32_t groups, float eps, int32_t silu, void *y, void *stats_ws, ctx_stream_t stream)
{
    return ctx_groupnorm_any(x, 0, gamma, beta, B, HW, C, groups, eps, silu, y, stats_ws, (hipStream_t)stream);
}

// x32 != 0: the input is the fp32 residual stream (the output stays fp16: it is the next GEMM's operand)
int ctx_groupnorm_any(const void *x, int x32, const void *gamma, const void *beta, int B, int HW, int C, int groups, float eps, int silu,
                      void *y, void *stats_ws, hipStream_t stream)
{
    CTX_REQUIRE(x && gamma && beta && y && stats_ws, "groupnorm: null pointer");
    CTX_REQUIRE(B > 0 && HW > 0 && C % 8 == 0 && C % groups == 0 && groups <= GN_MAX_GROUPS && C <= GN_MAX_C &&
                    256 % groups == 0 && (256 / groups & (256 / groups - 1)) == 0,
                "groupnorm: unsupported B=%d HW=%d C=%d groups=%d", B, HW, C, groups);
    hipStream_t s = stream;
    {
        static int fuse = -1;
        if (fuse < 0) { const char *e = getenv("CTX_GN_FUSED"); fuse = e ? atoi(e) : 1; }
        const int cg = C / groups;
        const int cpg = cg / 8;
        const int plf = cg % 8 == 0 ? GN_FT / cpg : 0;                  // pixel lanes of the one-kernel form
        if (fuse && cg % 8 == 0 && cpg <= GN_FT && (HW + plf - 1) / plf <= GN_FU) {
            const int thr = (cpg * plf + 63) / 64 * 64;
            if (x32) hipLaunchKernelGGL(k_gn_fused<true>, dim3(groups, B), dim3(thr), 0, s, x, (const f16 *)gamma, (const f16 *)beta, HW, C, groups, plf, eps, silu, (f16 *)y);
            else hipLaunchKernelGGL(k_gn_fused<false>, dim3(groups, B), dim3(thr), 0, s, x, (const f16 *)gamma, (const f16 *)beta, HW, C, groups, plf, eps, silu, (f16 *)y);
            CTX_CHECK_LAUNCH("groupnorm");
            return CTX_OK;
        }
    }
    int c8n = C / 8;
    int PL = 1024 / c8n;                                      // pixel lanes: ~1000 threads per block
    if (PL < 1) PL = 1;
    if (PL > HW) PL = HW;
    int threads = c8n * PL;
    int NS = min(GN_MAX_SPLITS, max(1, HW / PL));             // >= one pixel per lane per split
    {
        // Fat splits: a block's fold (two barriers, LDS walks, a shuffle tree) costs the same whatever it summed, and every apply block
        // re-reads all NS partials of its sample — but a 150 MB VAE tensor at batch 1 still needs all 128 of them to fill the chip.
        // ~384 KB of the sample per split, at least 64 stats blocks in all (measured, GroupNorm per evaluation: UNet batch 12 3.30 ms at
        // 128 splits / 2.13 at 16; batch 2 1.01 / 0.94 at 32; the VAE decoder's at 768^2 1.36 ms at 128 / 1.71 at 48).  CTX_GN_NS overrides.
        static const int ns_env = [] { const char *e = getenv("CTX_GN_NS"); return e ? atoi(e) : 0; }();
        const int64_t sample_bytes = (int64_t)HW * C * (x32 ? 4 : 2);
        const int by_size = (int)min((int64_t)GN_MAX_SPLITS, sample_bytes / (384 * 1024));
        const int want = ns_env > 0 ? ns_env : max(16, max(by_size, (64 + B - 1) / B));
        NS = min(NS, want);
    }
    float *part = (float *)stats_ws;
    const int na = threads / C > 1 ? threads / C : 1;
    size_t lds = (size_t)(PL + na) * C * 2 * sizeof(float);   // <= 72 KiB (threads <= 1024, 8 channels each)
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)k_gn_stats<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void *)k_gn_stats<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr = true;
    }
    if (x32) hipLaunchKernelGGL(k_gn_stats<true>, dim3(NS, B), dim3(threads), lds, s, x, HW, C, groups, NS, PL, part);
    else hipLaunchKernelGGL(k_gn_stats<false>, dim3(NS, B), dim3(threads), lds, s, x, HW, C, groups, NS, PL, part);
    size_t total = (size_t)HW * c8n;
    // fat blocks (the per-block fold of the split partials is amortised), at least one batch each
    {
        const int au = x32 ? GN_AU32 : GN_AU;
        const int apl = (256 + c8n - 1) / c8n;                      // pixel lanes: >= 256 threads (the fold uses 256), <= 512
        const int athreads = c8n * apl;
        const int nb = (HW + apl * au - 1) / (apl * au);
        if (x32) hipLaunchKernelGGL(k_gn_apply<true>, dim3(nb, B), dim3(athreads), 0, s, x, part, (const f16 *)gamma, (const f16 *)beta, HW, C,
                                    groups, NS, apl, eps, silu, (f16 *)y);
        else hipLaunchKernelGGL(k_gn_apply<false>, dim3(nb, B), dim3(athreads), 0, s, x, part, (const f16 *)gamma, (const f16 *)beta, HW, C,
                                groups, NS, apl, eps, silu, (f16 *)y);
    }
    CTX_CHECK_LAUNCH("groupnorm");
    return CTX_OK;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over the last dim: one wave per LN_R rows at a time, up to 4 x 16-byte chunks per lane per row
// (C <= 2048); all LN_R rows' loads are issued before the first reduction.
template <int KC, int R, bool X32>
__global__ __launch_bounds__(256) void k_layernorm(const void *__restrict__ x, const f16 *__restrict__ gamma,
                                                   const f16 *__restrict__ beta, int64_t rows, int C, float eps,
                                                   f16 *__restrict__ y)
{
    const int lane = threadIdx.x & 63;
    const int c8n = C / 8;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t row0 = wave * R;
    if (row0 >= rows) return;
    F8 v[R][KC];
#pragma unroll
    for (int rr = 0; rr < R; ++rr)
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            int c8 = lane + 64 * k;
            const int64_t rw = row0 + rr < rows ? row0 + rr : rows - 1;
            v[rr][k] = ld8<X32>(x, (size_t)(rw * C + min(c8, c8n - 1) * 8));          // unconditional load, masked below
            if (c8 >= c8n) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[rr][k].v[j] = 0.f;
            }
        }
    f16x8 ga[KC], be[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        int c8 = lane + 64 * k;
        ga[k] = *(const f16x8 *)(gamma + min(c8, c8n - 1) * 8);
        be[k] = *(const f16x8 *)(beta + min(c8, c8n - 1) * 8);
    }
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        if (row0 + rr >= rows) break;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < KC; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[rr][k].v[j];                 // padded lanes hold zeros
        float mean = wave_sum_dpp(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            int c8 = lane + 64 * k;
            if (c8 < c8n) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { float d = v[rr][k].v[j] - mean; q += d * d; }
            }
        }
        float rstd = rsqrtf(wave_sum_dpp(q) / (float)C + eps);
        f16 *yr = y + (row0 + rr) * C;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            int c8 = lane + 64 * k;
            if (c8 < c8n) {
                f16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (f16)((v[rr][k].v[j] - mean) * rstd * (float)ga[k][j] + (float)be[k][j]);
                *(f16x8 *)(yr + c8 * 8) = o;
            }
        }
    }
}

// 8 lanes per row, KC 16-byte chunks per lane (C = 64 KC): every lane of the wave carries data (the one-wave-per-row form above leaves
// 24 of 64 lanes idle at C = 320 and 640), a lane's chunks k*8 + s make 128-byte runs with its 7 neighbours, and a wave keeps 8 rows x KC
// loads in flight.  The 8-lane sums are three DPP adds (quad_perm, quad_perm, row_half_mirror); same two-pass mean / variance and the same
// rounding points as k_layernorm, another summation order.
__device__ __forceinline__ float sum8_dpp(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    return v;
}
template <int KC>
__global__ __launch_bounds__(256) void k_layernorm_g8(const f16 *__restrict__ x, const f16 *__restrict__ gamma, const f16 *__restrict__ beta,
                                                      int64_t rows, float eps, f16 *__restrict__ y)
{
    constexpr int C = 64 * KC;
    const int lane = threadIdx.x & 63, s = lane & 7, rw = lane >> 3;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t row = wave * 8 + rw;
    const bool live = row < rows;                                   // uniform over the row's 8 lanes
    const f16 *xr = x + (live ? row : rows - 1) * C + s * 8;
    f16x8 v[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) v[k] = *(const f16x8 *)(xr + k * 64);
    f16x8 ga[KC], be[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        ga[k] = *(const f16x8 *)(gamma + k * 64 + s * 8);
        be[k] = *(const f16x8 *)(beta + k * 64 + s * 8);
    }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < KC; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += (float)v[k][j];
    const float mean = sum8_dpp(sum) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < KC; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = (float)v[k][j] - mean; q += d * d; }
    const float rstd = rsqrtf(sum8_dpp(q) / (float)C + eps);
    if (!live) return;
    f16 *yr = y + row * C + s * 8;
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)(((float)v[k][j] - mean) * rstd * (float)ga[k][j] + (float)be[k][j]);
        *(f16x8 *)(yr + k * 64) = o;
    }
}

extern "C" int32_t ctx_layernorm_f16(const void *x, const void *gamma, const void *beta, int64_t rows, int32_t C, float eps,
                                     void *y, ctx_stream_t stream)
{
    return ctx_layernorm_any(x, 0, gamma, beta, rows, C, eps, y, (hipStream_t)stream);
}

int ctx_layernorm_any(const void *x, int x32, const void *gamma, const void *beta, int64_t rows, int C, float eps, void *y, hipStream_t stream)
{
    CTX_REQUIRE(x && gamma && beta && y && rows > 0 && C % 8 == 0 && C <= 2048, "layernorm: unsupported rows=%lld C=%d", (long long)rows, C);
    {
        // fp16 input, C a multiple of 64 up to 640 (beyond that the one-wave-per-row form fills >= 83 % of its lanes and gamma / beta for 20
        // chunks per lane would not fit the registers): 8 lanes per row (CTX_LN_G8=0: the one-wave-per-row kernels)
        static const int g8 = [] { const char *e = getenv("CTX_LN_G8"); return e ? atoi(e) : 1; }();
        if (g8 && !x32 && C % 64 == 0 && C <= 640 && rows >= 64) {
            const unsigned nb = (unsigned)cdiv64(cdiv64(rows, 8), 4);
#define LN_G8(KC_) case KC_: hipLaunchKernelGGL(k_layernorm_g8<KC_>, dim3(nb), dim3(256), 0, stream, (const f16 *)x, (const f16 *)gamma, (const f16 *)beta, rows, eps, (f16 *)y); break
            switch (C / 64) {
                LN_G8(1); LN_G8(2); LN_G8(3); LN_G8(4); LN_G8(5); LN_G8(6); LN_G8(7); LN_G8(8); LN_G8(9); LN_G8(10);
            }
#undef LN_G8
            CTX_CHECK_LAUNCH("layernorm");
            return CTX_OK;
        }
    }
    const int kc = (C / 8 + 63) / 64;                          // 16-byte chunks per lane per row
#define LN_GO(KC_, R_) do { int64_t nb = cdiv64(cdiv64(rows, R_), 4); \
        if (x32) hipLaunchKernelGGL((k_layernorm<KC_, R_, true>), dim3((unsigned)nb), dim3(256), 0, stream, x, \
                           (const f16 *)gamma, (const f16 *)beta, rows, C, eps, (f16 *)y); \
        else hipLaunchKernelGGL((k_layernorm<KC_, R_, false>), dim3((unsigned)nb), dim3(256), 0, stream, x, \
                           (const f16 *)gamma, (const f16 *)beta, rows, C, eps, (f16 *)y); } while (0)
    // rows per wave: ~4 loads in flight per lane, but keep >= ~2 waves per SIMD of work on the chip
    const bool many = rows >= 8192;
    if (kc == 1) { if (many) LN_GO(1, 4); else LN_GO(1, 1); }
    else if (kc == 2) { if (many) LN_GO(2, 2); else LN_GO(2, 1); }
    else if (kc == 3) LN_GO(3, 1);
    else LN_GO(4, 1);
#undef LN_GO
    CTX_CHECK_LAUNCH("layernorm");
    return CTX_OK;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_geglu(const f16 *__restrict__ h, int64_t M, int C4, f16 *__restrict__ y)
{
    const int c8n = C4 / 8;
    const int64_t total = M * c8n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t m = i / c8n;
        int c8 = (int)(i % c8n);
        f16x8 a = *(const f16x8 *)(h + m * 2 * C4 + c8 * 8);
        f16x8 g = *(const f16x8 *)(h + m * 2 * C4 + C4 + c8 * 8);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float gf = (float)g[j];
            o[j] = (f16)((float)a[j] * (0.5f * gf * (1.0f + erff(gf * 0.70710678118654752f))));
        }
        *(f16x8 *)(y + m * C4 + c8 * 8) = o;
    }
}

extern "C" int32_t ctx_geglu_f16(const void *h, int64_t M, int32_t C4, void *y, ctx_stream_t stream)
{
    CTX_REQUIRE(h && y && M > 0 && C4 % 8 == 0, "geglu: bad args");
    int64_t nb = cdiv64(M * (C4 / 8), 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_geglu, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const f16 *)h, M, C4, (f16 *)y);
    CTX_CHECK_LAUNCH("geglu");
    return CTX_OK;
}

// ------------------------------------------------------------------------------------------------
// Channel concat (NHWC): y[m, :Ca] = a[m], y[m, Ca:] = b[m].
__global__ __launch_bounds__(256) void k_concat(const f16 *__restrict__ a, const f16 *__restrict__ b, int64_t M, int Ca,
                                                int Cb, f16 *__restrict__ y)
{
    const int n8 = (Ca + Cb) / 8, a8 = Ca / 8;
    const int64_t total = M * n8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t m = i / n8;
        int c8 = (int)(i % n8);
        f16x8 v = c8 < a8 ? *(const f16x8 *)(a + m * Ca + c8 * 8) : *(const f16x8 *)(b + m * Cb + (c8 - a8) * 8);
        *(f16x8 *)(y + i * 8) = v;
    }
}

// same copy with 4-byte elements (the fp32 residual stream): channels counted in f16-equivalents of 2 x the float count
int ctx_concat_f32(const float *a, const float *b, int64_t M, int Ca, int Cb, float *y, hipStream_t s)
{
    return ctx_concat_f16((const f16 *)a, (const f16 *)b, M, 2 * Ca, 2 * Cb, (f16 *)y, s);
}

int ctx_concat_f16(const f16 *a, const f16 *b, int64_t M, int Ca, int Cb, f16 *y, hipStream_t s)
{
    int64_t nb = cdiv64(M * ((Ca + Cb) / 8), 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_concat, dim3((unsigned)nb), dim3(256), 0, s, a, b, M, Ca, Cb, y);
    return CTX_OK;
}

// V [B, S, ld] (head slice at column h*64) -> Vt [B, heads, 64, Sp] (keys contiguous, zero padded to Sp).  Inside every
// group of 16 keys the order is [0-3, 8-11, 4-7, 12-15] when perm != 0 (attention's layout; perm = 0 is a plain transpose): the 8 keys one lane half feeds to a PV MFMA k-step (the P
// fragment is a 32x32 accumulator: keys 8(j>>2) + 4h + (j&3)) are then one 16-byte chunk (attention.hip: attn_tile).
__global__ __launch_bounds__(256) void k_transpose_v(const f16 *__restrict__ v, int S, int ld, int heads, int Sp, int perm,
                                                     f16 *__restrict__ vt)
{
    __shared__ f16 tile[64][66];
    const int b = blockIdx.z, hd = blockIdx.y, s0 = blockIdx.x * 64;
    // load 64 keys x 64 d, 16 B per lane: thread t -> key t/8 + 32*i, chunk t%8
    for (int i = 0; i < 2; ++i) {
        int key = (threadIdx.x >> 3) + 32 * i, c = threadIdx.x & 7;
        f16x8 val = {0, 0, 0, 0, 0, 0, 0, 0};
        if (s0 + key < S) val = *(const f16x8 *)(v + ((size_t)b * S + s0 + key) * ld + hd * 64 + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[key][c * 8 + j] = val[j];
    }
    __syncthreads();
    for (int i = 0; i < 2; ++i) {
        int d = (threadIdx.x >> 3) + 32 * i, c = threadIdx.x & 7;
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = perm ? tile[16 * (c >> 1) + 8 * (j >> 2) + 4 * (c & 1) + (j & 3)][d] : tile[c * 8 + j][d];
        if (s0 + c * 8 < Sp) *(f16x8 *)(vt + (((size_t)b * heads + hd) * 64 + d) * Sp + s0 + c * 8) = o;
    }
}

int ctx_transpose_v_f16(const f16 *v, int B, int S, int ld, int heads, int Sp, int perm, f16 *vt, hipStream_t s)
{
    hipLaunchKernelGGL(k_transpose_v, dim3(cdiv(Sp, 64), heads, B), dim3(256), 0, s, v, S, ld, heads, Sp, perm, vt);
    return CTX_OK;
}

// ------------------------------------------------------------------------------------------------
// f32 -> f16 row-major copy (context embeddings).
__global__ __launch_bounds__(256) void k_f32_to_f16(const float *__restrict__ x, int64_t n, f16 *__restrict__ y)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = (f16)x[i];
}
__global__ __launch_bounds__(256) void k_f16_to_f32(const f16 *__restrict__ x, int64_t n8, float *__restrict__ y)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const f16x8 v = *(const f16x8 *)(x + i * 8);
        f32x4 a = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]}, b = {(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
        *(f32x4 *)(y + i * 8) = a; *(f32x4 *)(y + i * 8 + 4) = b;
    }
}
int ctx_f16_to_f32(const f16 *x, int64_t n, float *y, hipStream_t s)
{
    int64_t nb = cdiv64(n / 8, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_f16_to_f32, dim3((unsigned)nb), dim3(256), 0, s, x, n / 8, y);
    return CTX_OK;
}
int ctx_f32_to_f16(const float *x, int64_t n, f16 *y, hipStream_t s)
{
    int64_t nb = cdiv64(n, 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_f32_to_f16, dim3((unsigned)nb), dim3(256), 0, s, x, n, y);
    return CTX_OK;
}

// Sinusoidal timestep embedding, diffusers get_timestep_embedding(flip_sin_to_cos=True, freq_shift=0):
// emb[b] = [cos(t*f_0..f_{h-1}), sin(t*f_0..)] with f_i = exp(-ln(10000) * i / h), h = dim/2.
__global__ void k_time_embed(const float *__restrict__ t, int B, int dim, f16 *__restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int half = dim / 2;
    if (i >= half) return;
    float fr = expf(-9.210340371976184f * (float)i / (float)half);
    float a = t[0] * fr;
    float c = cosf(a), sn = sinf(a);
    for (int b = 0; b < B; ++b) {
        out[(size_t)b * dim + i] = (f16)c;
        out[(size_t)b * dim + half + i] = (f16)sn;
    }
}
int ctx_time_embed_f16(const float *t, int B, int dim, f16 *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_time_embed, dim3(cdiv(dim / 2, 64)), dim3(64), 0, s, t, B, dim, out);
    return CTX_OK;
}

// conv_in: sample [B,Cin,H,W] f32 NCHW (Cin <= 8) -> y [B,H,W,Cout] f16, 3x3 pad 1.  w packed [Cout][3][3][8] f16.
// One pixel per lane: its 9 x Cin inputs live in registers; the weights of this block's slice of output channels
// sit in LDS and are read as wave-wide broadcasts; grid.y splits the output channels.
#define CI_SPLIT 16
// CP = padded input channels of the weight pack [Cout][3][3][CP]: 8 (latents + depth, VAE) or 16 (the 9-channel inpainting UNet);
// CX = channels actually multiplied (the pack's zero padding is skipped).  The block's weight slice is converted to fp32 once
// when it is staged (the kernel is VALU-bound: one cvt per FMA otherwise).
template <int CP, int CX>
__global__ __launch_bounds__(256) void k_conv_in(const float *__restrict__ x, const f16 *__restrict__ w,
                                                 const f16 *__restrict__ bias, int B, int Cin, int H, int W, int Cout,
                                                 f16 *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) float s_w[];   // [o_per * 8][9][CX]
    constexpr int WR = 9 * CP, WX = 9 * CX;
    const int o8n = Cout / 8;
    const int o8_per = (o8n + CI_SPLIT - 1) / CI_SPLIT;
    const int o8_0 = blockIdx.y * o8_per, o8_1 = min(o8n, o8_0 + o8_per);
    const int no = (o8_1 - o8_0) * 8;
    for (int i = threadIdx.x; i < no * WX; i += 256) {
        const int o = i / WX, r = i - o * WX, t = r / CX, c = r - t * CX;
        s_w[i] = (float)w[(size_t)(o8_0 * 8 + o) * WR + t * CP + c];
    }
    __syncthreads();
    const int64_t npix = (int64_t)B * H * W;
    const int64_t pix0 = (int64_t)blockIdx.x * 256;
    const int64_t pix = pix0 + threadIdx.x;
    const bool live = pix < npix;
    const int64_t pq = live ? pix : npix - 1;
    int b = (int)(pq / (H * W)), p = (int)(pq % (H * W));
    int oy = p / W, ox = p % W;
    float in[9][CX];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
        bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
#pragma unroll
        for (int c = 0; c < CX; ++c)
            in[t][c] = (ok && c < Cin) ? (float)(f16)x[(((size_t)b * Cin + c) * H + iy) * W + ix] : 0.f;
    }
    // A lane's 8 channels are 16 bytes of a pixel row that is Cout x 2 bytes long: stored directly that is one 16-byte piece
    // per cache line and instruction (the kernel was bound by those stores: 53 us for 11.8 MB at 96^2 x 320).  Eight channel
    // groups at a time go through an LDS patch [256 pixels][64 channels] and leave as 128-byte row segments.
    f16 *patch = (f16 *)(s_w + no * WX);
    for (int g0 = o8_0; g0 < o8_1; g0 += 8) {
        const int ng = min(8, o8_1 - g0);
        for (int gi = 0; gi < ng; ++gi) {
            const int o8 = g0 + gi;
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float *wr = s_w + ((o8 - o8_0) * 8 + j) * WX;
                float acc = (float)bias[o8 * 8 + j];
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int c = 0; c < CX; ++c) acc += in[t][c] * wr[t * CX + c];
                o[j] = (f16)acc;
            }
            *(f16x8 *)(patch + threadIdx.x * 72 + gi * 8) = o;      // row stride 72 f16 = 144 B: conflict-free 16-byte writes
        }
        __syncthreads();
        for (int ch = threadIdx.x; ch < 256 * ng; ch += 256) {
            const int px = ch / ng, gi = ch - px * ng;
            if (pix0 + px < npix) *(f16x8 *)(y + (pix0 + px) * Cout + (g0 + gi) * 8) = *(const f16x8 *)(patch + px * 72 + gi * 8);
        }
        __syncthreads();
    }
}
int ctx_conv_in_f16(const float *x, const f16 *w, const f16 *bias, int B, int Cin, int H, int W, int Cout, f16 *y, hipStream_t s)
{
    int64_t npix = (int64_t)B * H * W;
    int o8_per = (Cout / 8 + CI_SPLIT - 1) / CI_SPLIT;
    const dim3 grid((unsigned)cdiv64(npix, 256), CI_SPLIT);
#define CI_GO(CP_, CX_) hipLaunchKernelGGL((k_conv_in<CP_, CX_>), grid, dim3(256), (size_t)o8_per * 8 * 9 * CX_ * sizeof(float) + 256 * 72 * sizeof(f16), s, x, w, bias, B, Cin, H, W, Cout, y)
    if (Cin <= 3) CI_GO(8, 3);
    else if (Cin == 4) CI_GO(8, 4);
    else if (Cin == 5) CI_GO(8, 5);
    else if (Cin <= 8) CI_GO(8, 8);
    else if (Cin == 9) CI_GO(16, 9);
    else CI_GO(16, 16);
#undef CI_GO
    return CTX_OK;
}

// conv_out: x [B,H,W,C] f16 (already GN+SiLU'd) -> out [B,Cout,H,W] f32 NCHW, Cout <= 4, 3x3 pad 1.
// w packed [Cout][3][3][C] f16, staged in LDS once per block.  One wave per output pixel: the 9 taps x C/8 16-byte chunks
// of the pixel's neighbourhood are one flat item list dealt over the 64 lanes (all lanes busy for any C), every load of a
// lane issued (unconditionally, clamped) before the first use; the Cout sums are DPP wave reductions.
// WF32: the staged weights are converted to fp32 once (the kernel is VALU-bound: 40 cvt per 32 FMA otherwise); used when the
// fp32 copy fits 64 KiB of LDS
template <int NU, bool WF32>
__global__ __launch_bounds__(256) void k_conv_out(const f16 *__restrict__ x, const f16 *__restrict__ w,
                                                  const f16 *__restrict__ bias, int B, int H, int W, int C, int Cout,
                                                  float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char s_cw_raw[];    // [Cout][9][C] f16 or fp32
    f16 *s_cw = (f16 *)s_cw_raw;
    float *s_cf = (float *)s_cw_raw;
    const int nwt = Cout * 9 * C;
    if (WF32) {
        for (int i = threadIdx.x * 8; i < nwt; i += 256 * 8) {
            const f16x8 v = *(const f16x8 *)(w + i);
#pragma unroll
            for (int j = 0; j < 8; ++j) s_cf[i + j] = (float)v[j];
        }
    } else {
        for (int i = threadIdx.x * 8; i < nwt; i += 256 * 8) *(f16x8 *)(s_cw + i) = *(const f16x8 *)(w + i);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int c8n = C / 8, nitems = 9 * c8n;
    const int64_t npix = (int64_t)B * H * W;
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    int itap[NU], ic[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int i = min(lane + 64 * u, nitems - 1);
        itap[u] = i / c8n; ic[u] = (i - itap[u] * c8n) * 8;
    }
    // a wave walks its pixels with the NEXT pixel's loads issued before the current one is reduced (the kernel is bound by the
    // latency of those loads, not by the arithmetic: one pixel in flight per wave took 41 us at 96^2 x 320 channels)
    const int64_t pstep = ((int64_t)gridDim.x * 256) >> 6;
    f16x8 xn[NU];
    auto fetch = [&](int64_t pp_) {
        const int64_t q = pp_ < npix ? pp_ : npix - 1;
        const int b = (int)(q / (H * W)), p = (int)(q % (H * W));
        const int oy = p / W, ox = p % W;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int iy = oy + itap[u] / 3 - 1, ix = ox + itap[u] % 3 - 1;
            const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
            xn[u] = *(const f16x8 *)(x + (((size_t)b * H + cy) * W + cx) * C + ic[u]);
            if (lane + 64 * u >= nitems || iy != cy || ix != cx) xn[u] = zero8;
        }
    };
    int64_t pix = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (pix < npix) fetch(pix);
    for (; pix < npix; pix += pstep) {
        const int b = (int)(pix / (H * W)), p = (int)(pix % (H * W));
        const int oy = p / W, ox = p % W;
        f16x8 xv[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) xv[u] = xn[u];
        if (pix + pstep < npix) fetch(pix + pstep);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            float xf[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[j] = (float)xv[u][j];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (o < Cout) {
                    if (WF32) {
                        const float *wr = s_cf + itap[u] * C + ic[u] + o * 9 * C;
                        const f32x4 w0 = *(const f32x4 *)wr, w1 = *(const f32x4 *)(wr + 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { acc[o] += xf[j] * w0[j]; }
#pragma unroll
                        for (int j = 0; j < 4; ++j) { acc[o] += xf[4 + j] * w1[j]; }
                    } else {
                        const f16x8 wv = *(const f16x8 *)(s_cw + itap[u] * C + ic[u] + o * 9 * C);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[o] += xf[j] * (float)wv[j];
                    }
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (o < Cout) {
                float v = wave_sum_dpp(acc[o]);
                if (lane == 0) out[(((size_t)b * Cout + o) * H + oy) * W + ox] = v + (float)bias[o];
            }
        }
    }
}
int ctx_conv_out_f16(const f16 *x, const f16 *w, const f16 *bias, int B, int H, int W, int C, int Cout, float *out, hipStream_t s)
{
    int64_t nb = cdiv64((int64_t)B * H * W, 4);
    if (nb > 2048) nb = 2048;                                 // every block stages the weights: keep them few and persistent
    const int nitems = 9 * (C / 8);
    const size_t lds16 = (size_t)Cout * 9 * C * sizeof(f16);
    if (Cout > 4 || C % 8 != 0 || nitems > 64 * 12 || lds16 > 64 * 1024) {
        ctx_set_error("conv_out: unsupported C=%d Cout=%d", C, Cout);
        return CTX_E_ARG;
    }
    const bool f32w = 2 * lds16 <= 64 * 1024;
    const size_t lds = f32w ? 2 * lds16 : lds16;
#define CO_GO(NU_) do { if (f32w) hipLaunchKernelGGL((k_conv_out<NU_, true>), dim3((unsigned)nb), dim3(256), lds, s, x, w, bias, B, H, W, C, Cout, out); \
                        else hipLaunchKernelGGL((k_conv_out<NU_, false>), dim3((unsigned)nb), dim3(256), lds, s, x, w, bias, B, H, W, C, Cout, out); } while (0)
    if (nitems <= 64 * 3) CO_GO(3);
    else if (nitems <= 64 * 6) CO_GO(6);
    else CO_GO(12);
#undef CO_GO
    return CTX_OK;
}

// ------------------------------------------------------------------------------------------------
// CFG combine + PNDM/PLMS linear-multistep update, one pass over the latent.
__global__ __launch_bounds__(256) void k_cfg_plms(const float *__restrict__ eps_pair, int64_t n, float guidance,
                                                  float *__restrict__ ets, int head, float c0, float c1, float c2,
                                                  float c3, float sample_coeff, float eps_coeff, int mode,
                                                  float *__restrict__ cur, float *__restrict__ x)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float eu = eps_pair[i], et = eps_pair[n + i];
        float e = eu + guidance * (et - eu);
        float xs;
        float comb;
        if (mode == 1) {
            // second evaluation of the first PLMS step: average with the stored epsilon, restart from cur_sample
            comb = (e + ets[(size_t)head * n + i]) / 2.0f;
            xs = cur[i];
        } else {
            ets[(size_t)head * n + i] = e;
            float e1 = ets[(size_t)((head + 3) & 3) * n + i];
            float e2 = ets[(size_t)((head + 2) & 3) * n + i];
            float e3 = ets[(size_t)((head + 1) & 3) * n + i];
            comb = c0 * e;
            if (c1 != 0.f) comb += c1 * e1;
            if (c2 != 0.f) comb += c2 * e2;
            if (c3 != 0.f) comb += c3 * e3;
            xs = x[i];
            if (mode == 2) cur[i] = xs;      // first step: remember cur_sample
        }
        x[i] = sample_coeff * xs - eps_coeff * comb;
    }
}

extern "C" int32_t ctx_cfg_plms_step(const float *eps_pair, int64_t n, float guidance, float *ets, int32_t head,
                                     const float *coef4, float sample_coeff, float eps_coeff, int32_t mode,
                                     float *cur_sample_ws, float *x, ctx_stream_t stream)
{
    CTX_REQUIRE(eps_pair && ets && coef4 && x && n > 0 && head >= 0 && head < 4, "cfg_plms_step: bad args");
    CTX_REQUIRE(mode == 0 || cur_sample_ws, "cfg_plms_step: mode %d needs cur_sample_ws", mode);
    int64_t nb = cdiv64(n, 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_cfg_plms, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, eps_pair, n, guidance, ets, head,
                       coef4[0], coef4[1], coef4[2], coef4[3], sample_coeff, eps_coeff, mode, cur_sample_ws, x);
    CTX_CHECK_LAUNCH("cfg_plms_step");
    return CTX_OK;
}
