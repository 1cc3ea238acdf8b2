// fp16 MFMA GEMM / implicit-GEMM 3x3 convolution for the UNet denoiser (gfx950), small and medium tiles.
//
//   out[M,N] = X[M,K] . Wt[N,K]^T  (+bias[N]) (+rowbias[row/rows_per_batch, N]) (+residual[M,N])
//
// Replaces the cuDNN/cuBLAS conv2d / linear calls under diffusers' UNet2DConditionModel
// (reference call site src/stable_diffusion_depth.py:422-423).  fp16 operands, fp32 accumulate.
//
// One kernel template, k_gemm_pipe<WM, WN, MI, NI, CONV, NS, PKT>: WM x WN waves, each owning MI x NI accumulators of
// v_mfma_f32_32x32x16_f16.  The WEIGHT fragment is the A operand and the ACTIVATION fragment the B operand, so a lane
// owns one output row (token / pixel) and its registers walk the features.  Operand tiles go global -> LDS by
// global_load_lds_dwordx4 into a ring of NS K-stages (PKT deep), one raw s_barrier per stage and a counted vmcnt so the
// newer stages stay in flight.  What limits these layers is how many bytes per clock a CU can stage (about 1 KiB per
// ~120 cycles per issuing wave, ~30 B/clk per CU), not the matrix pipe: the tile list below trades flop per staged
// byte (big tiles) against waves that issue (small problems want 4-8 waves even on a 64x64 tile).  The conv variant
// only changes the activation address generator (im2col on the fly, NHWC, zero padding by pointing at a zero page,
// optional fused nearest x2 upsample and stride 2).  The large layers go to gemm8.hip (256x256, 8 staggered waves).
#include "common.h"
#include "kernels.h"
#include <hip/hip_ext.h>
#include <stdlib.h>

__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// exact-GELU (erf form) with a branch-free erf: Abramowitz-Stegun 7.1.26, |abs err| <= 1.5e-7 (far below fp16 resolution)
__device__ __forceinline__ float fast_erf(float x)
{
    float ax = __builtin_fabsf(x);
    float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    float p = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    float e = 1.0f - p * __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    return __builtin_copysignf(e, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752f)); }

// Direct epilogue (GEGLU, or when strides are not 16-byte friendly): lane owns row m, registers walk n.
template <int MI, int NI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &a, f32x16 (&acc)[MI][NI], int mw, int nw, int r, int h)
{
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        int m = mw + mi * 32 + r;
        if (m >= a.M) continue;
        int bidx = a.rowbias ? m / a.rows_per_batch : 0;
        if (a.epi == 1) {
            if constexpr (NI == 2) {
                // GEGLU: ni=0 -> value half, ni=1 -> gate half of the same 32 features (packed weight order)
                int fbase = nw / 2;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int nn = nw + 8 * g + 4 * h;       // packed column of the value half
                    if (nn >= a.N) continue;
                    f16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float xv = acc[mi][0][4 * g + j], gv = acc[mi][1][4 * g + j];
                        if (a.bias) { xv += (float)a.bias[nn + j]; gv += (float)a.bias[nn + 32 + j]; }
                        o[j] = (f16)(xv * gelu_erf(gv));
                    }
                    *(f16x4 *)(a.out + (size_t)m * a.ldc + fbase + 8 * g + 4 * h) = o;
                }
            }
        } else {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int nn = nw + ni * 32 + 8 * g + 4 * h;
                    if (nn >= a.N) continue;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[mi][ni][4 * g + j];
                    if (a.bias) {
                        f16x4 b = *(const f16x4 *)(a.bias + nn);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += (float)b[j];
                    }
                    if (a.rowbias) {
                        f16x4 b = *(const f16x4 *)(a.rowbias + (size_t)bidx * a.ldrb + nn);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += (float)b[j];
                    }
                    if (a.residual) {
                        if (a.res32) {
                            f32x4 b = *(const f32x4 *)((const float *)a.residual + (size_t)m * a.ldr + nn);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] += b[j];
                        } else {
                            f16x4 b = *(const f16x4 *)(a.residual + (size_t)m * a.ldr + nn);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] += (float)b[j];
                        }
                    }
                    if (a.out32) {
                        *(f32x4 *)((float *)a.out + (size_t)m * a.ldc + nn) = (f32x4){v[0], v[1], v[2], v[3]};
                    } else {
                        f16x4 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = (f16)v[j];
                        *(f16x4 *)(a.out + (size_t)m * a.ldc + nn) = o;
                    }
                }
        }
    }
}

// Epilogue through LDS: the accumulator layout gives every lane one token row and 4-feature pieces, i.e. 8-byte
// accesses scattered over 32 cache lines per wave instruction, for the store AND for the residual read.  Each wave
// instead parks 32 rows x (32 NI) features of fp32 in its own LDS patch (row stride 32 NI + 4 floats: conflict-free
// ds_write_b128) and walks it back in whole rows: bias / row bias / residual / output all move as 64- or 128-byte
// row segments, 16 bytes per lane, and the sum is still rounded to fp16 exactly once.
template <int MI, int NI>
__device__ __forceinline__ void gemm_epilogue_staged(const GemmArgs &a, f32x16 (&acc)[MI][NI], int mw, int nw, int r, int h,
                                                     float *stage, int lane)
{
    constexpr int RS = 32 * NI + 4;                    // patch row stride (floats)
    constexpr int LPR = 4 * NI;                        // lanes per row (8 features each)
    constexpr int RPP = 64 / LPR;                      // rows per pass
    const int prow = lane / LPR, c8 = (lane % LPR) * 8;
    const int n = nw + c8;
    f16x8 bs = {0, 0, 0, 0, 0, 0, 0, 0};
    if (a.bias && n < a.N) bs = *(const f16x8 *)(a.bias + n);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
                *(f32x4 *)(stage + r * RS + ni * 32 + 8 * g + 4 * h) = v;
            }
#pragma unroll
        for (int p = 0; p < 32 / RPP; ++p) {
            const int row = RPP * p + prow;
            const int m = mw + mi * 32 + row;
            f32x4 v0 = *(const f32x4 *)(stage + row * RS + c8), v1 = *(const f32x4 *)(stage + row * RS + c8 + 4);
            if (m < a.M && n < a.N) {
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += (float)bs[j];
                if (a.rowbias) {
                    f16x8 b = *(const f16x8 *)(a.rowbias + (size_t)(m / a.rows_per_batch) * a.ldrb + n);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += (float)b[j];
                }
                if (a.residual) {
                    if (a.res32) {
                        const float *rp = (const float *)a.residual + (size_t)m * a.ldr + n;
                        f32x4 b0 = *(const f32x4 *)rp, b1 = *(const f32x4 *)(rp + 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { v[j] += b0[j]; v[4 + j] += b1[j]; }
                    } else {
                        f16x8 b = *(const f16x8 *)(a.residual + (size_t)m * a.ldr + n);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] += (float)b[j];
                    }
                }
                if (a.out32) {
                    float *op = (float *)a.out + (size_t)m * a.ldc + n;
                    *(f32x4 *)op = (f32x4){v[0], v[1], v[2], v[3]}; *(f32x4 *)(op + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                } else {
                    f16x8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (f16)v[j];
                    *(f16x8 *)(a.out + (size_t)m * a.ldc + n) = o;
                }
            }
        }
    }
}

__device__ __attribute__((aligned(16))) f16 g_zero_page[64];

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// ------------------------------------------------------------------------------------------------
// Pipelined LDS-DMA variant: K-stages of 32, ring of NS stages, prefetch distance NS-1, ONE raw s_barrier per
// stage and a counted s_waitcnt vmcnt so the newer stages' DMA stays in flight across the barrier.
//   iteration kt:  vmcnt((NS-2)*G)  -> this wave's pieces of stage kt have landed
//                  s_barrier        -> everybody's pieces landed AND everybody finished stage kt-1
//                  issue stage kt+NS-1 into the buffer stage kt-1 used
//                  8 x MFMA on stage kt
// 64-byte LDS rows ([rows][32 f16]), chunk swizzle c ^ ((row>>2)&3) applied on the DMA source address and on the
// fragment read (conflict-free ds_read_b128 under the 64-bank / 16-lane-group rule).
template <int WM, int WN, int MI, int NI, bool CONV, int NS, int PKT>
__global__ __launch_bounds__(64 * WM * WN) void k_gemm_pipe(GemmArgs a)
{
    constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN, NW = WM * WN;
    constexpr int RP = 512 / PKT;                   // rows per 1-KiB DMA piece (16 x 64 B or 8 x 128 B)
    constexpr int LPR = PKT / 8;                    // lanes (16-B chunks) per row
    constexpr int XI = BM / (RP * NW), WI = BN / (RP * NW);   // DMA pieces per wave per stage
    static_assert(XI >= 1 && WI >= 1, "tile too small for the wave count");
    constexpr int G = XI + WI;
    constexpr int STAGE = (BM + BN) * PKT;           // f16 per stage
    extern __shared__ __attribute__((aligned(16))) f16 smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: SGPR, scalar branches
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    // 1-D grid of tiles x K-slices, slice-major, cut into 8 contiguous runs (one per XCD: blocks b and b+8 share an L2):
    // an XCD then streams ~1/8 of the K range of both operands.  Inside a slice the tile order walks the operand that is
    // re-read most (mfast: consecutive blocks share the weight panel; else the activation panel).
    const int ntiles = a.ntm * a.ntn;
    const int lin = xcd_remap(blockIdx.x, ntiles * a.splitk);
    const int slice = lin / ntiles, bid = lin - slice * ntiles;
    const int tile_n = a.mfast ? bid / a.ntm : bid % a.ntn, tile_m = a.mfast ? bid % a.ntm : bid / a.ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int lr = lane / LPR, pc = lane % LPR;        // row within the piece's 16 rows, physical 16-B chunk

    const int nk_all = a.K / PKT;
    const int kbeg = (int)((long)nk_all * slice / a.splitk);
    const int nk = (int)((long)nk_all * (slice + 1) / a.splitk) - kbeg;

    // ---- issue-side state: one running source pointer per DMA piece (advanced by a fixed step per stage; rows that
    // are out of range / conv padding point at a zero page with step 0), so a stage costs G loads + G pointer adds
    const f16 *xp[XI], *wp[WI];
    int xst[XI], wst[WI];
    int xoff[XI], xoy[XI], xox[XI], xlc[XI];
    bool xok[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        int row = RP * (wave + NW * i) + lr;
        int m = m0 + row;
        xlc[i] = (pc ^ (PKT == 32 ? ((row >> 2) & 3) : ((row >> 1) & 7))) * 8;
        xok[i] = m < a.M;
        if (CONV) {
            int hw = a.Ho * a.Wo;
            int b = m / hw, p = m - b * hw;
            int oy = p / a.Wo, ox = p - oy * a.Wo;
            xoy[i] = oy * a.stride; xox[i] = ox * a.stride;
            xoff[i] = b * a.H * a.W * a.Cin;
            xp[i] = g_zero_page; xst[i] = 0;
        } else {
            xoff[i] = 0; xoy[i] = 0; xox[i] = 0;
            xp[i] = xok[i] ? a.X + (size_t)m * a.K + kbeg * PKT + xlc[i] : g_zero_page;
            xst[i] = xok[i] ? PKT : 0;
        }
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        int row = RP * (wave + NW * i) + lr;
        bool ok = (n0 + row) < a.N;
        wp[i] = ok ? a.Wt + (size_t)(n0 + row) * a.K + kbeg * PKT + ((pc ^ (PKT == 32 ? ((row >> 2) & 3) : ((row >> 1) & 7))) * 8) : g_zero_page;
        wst[i] = ok ? PKT : 0;
    }
    // K rotation: workgroups that stream the same weight rows (same tile_n) or the same activation rows (same tile_m)
    // start at different K offsets and wrap, so at any instant they hit different cache lines / L2 channels.
    const int k_lo = kbeg * PKT, k_hi = (kbeg + nk) * PKT;
    const int rot = a.krot == 1 ? (int)(((unsigned)tile_m * 13u + (unsigned)tile_n * 5u) % (unsigned)nk) : 0;
    int k_issue = k_lo + rot * PKT, issued = 0, tap_left = 0;
    if (!CONV) {
#pragma unroll
        for (int i = 0; i < XI; ++i) xp[i] += (xst[i] ? rot * PKT : 0);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) wp[i] += (wst[i] ? rot * PKT : 0);

    auto retap = [&]() {                            // CONV: new 3x3 tap -> recompute the activation pointers
        int tap = k_issue / a.Cin, c0 = k_issue - tap * a.Cin;
        int dy = tap / 3 - 1 + a.poff, dx = tap % 3 - 1 + a.poff;
        int Hv = a.H << a.ups, Wv = a.W << a.ups;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            int iy = xoy[i] + dy, ix = xox[i] + dx;
            bool ok = xok[i] && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv && !(a.zins && ((iy | ix) & 1));
            xp[i] = ok ? a.X + xoff[i] + (((iy >> a.ups) * a.W + (ix >> a.ups)) * a.Cin) + c0 + xlc[i] : g_zero_page;
            xst[i] = ok ? PKT : 0;
        }
        tap_left = (a.Cin - c0) / PKT;
    };
    auto issue = [&](int buf) {
        if (k_issue == k_hi) {                       // wrap of the rotated K range
            k_issue = k_lo;
            const int span = k_hi - k_lo;
            if (!CONV) {
#pragma unroll
                for (int i = 0; i < XI; ++i) xp[i] -= (xst[i] ? span : 0);
            }
#pragma unroll
            for (int i = 0; i < WI; ++i) wp[i] -= (wst[i] ? span : 0);
            tap_left = 0;
        }
        if (CONV) {
            if (tap_left == 0) retap();
            --tap_left;
        }
        f16 *Xs = smem + buf * STAGE;
        f16 *Ws = Xs + BM * PKT;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)xp[i], (lptr_t)(Xs + (wave + NW * i) * 512), 16, 0, 0);
            xp[i] += xst[i];
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)wp[i], (lptr_t)(Ws + (wave + NW * i) * 512), 16, 0, 0);
            wp[i] += wst[i];
        }
        k_issue += PKT;
        ++issued;
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < nk) issue(p);
    const int swz = PKT == 32 ? ((r >> 2) & 3) : ((r >> 1) & 7);
    const int xrow = (wm * 32 * MI + r) * PKT, wrow = BM * PKT + (wn * 32 * NI + r) * PKT;
    int kt = 0;
    // steady groups of NS stages: every stage of the group still issues a stage NS-1 ahead, so exactly NS-2 newer stages are in
    // flight at each wait — none of the wave-uniform branches of the general form below (about ten s_cbranch per stage in the ISA)
    for (; a.krot == 0 && kt + 2 * NS - 2 < nk; kt += NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            // lgkmcnt(0): the barrier does not wait for LDS reads in flight, and the stage issued right after it lands in the
            // buffer the previous stage's fragments were read from (hipcc sinks those reads' MFMAs below the barrier when it can)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * G) : "memory");
            ctx_barrier();
            issue((u + NS - 1) % NS);
            const f16 *sb = smem + u * STAGE;
#pragma unroll
            for (int ks = 0; ks < PKT / 16; ++ks) {
                const int pch = ((2 * ks + h) ^ swz) * 8;
                f16x8 xf[MI], wf[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) xf[i] = *(const f16x8 *)(sb + xrow + i * 32 * PKT + pch);
#pragma unroll
                for (int j = 0; j < NI; ++j) wf[j] = *(const f16x8 *)(sb + wrow + j * 32 * PKT + pch);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
            }
        }
    }
    for (; kt < nk; kt += NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            if (kt + u < nk) {
                // pieces still allowed in flight after this wait: those of the (up to NS-2) newer issued stages
                int newer = issued - 1 - (kt + u);
                if (NS >= 8 && newer >= 6) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * G) : "memory");
                else if (NS >= 7 && newer >= 5) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * G) : "memory");
                else if (NS >= 6 && newer >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * G) : "memory");
                else if (NS >= 5 && newer >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * G) : "memory");
                else if (NS >= 4 && newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
                else if (newer >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // see the steady loop
                ctx_barrier();
                if (issued < nk) issue((u + NS - 1) % NS);
                const f16 *sb = smem + u * STAGE;
#pragma unroll
                for (int ks = 0; ks < PKT / 16; ++ks) {
                    const int pch = ((2 * ks + h) ^ swz) * 8;
                    f16x8 xf[MI], wf[NI];
#pragma unroll
                    for (int i = 0; i < MI; ++i) xf[i] = *(const f16x8 *)(sb + xrow + i * 32 * PKT + pch);
#pragma unroll
                    for (int j = 0; j < NI; ++j) wf[j] = *(const f16x8 *)(sb + wrow + j * 32 * PKT + pch);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
                }
            }
        }
    }
    if (a.splitk > 1) {
        float *pb = a.part + (size_t)slice * a.M * a.N;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            int m = m0 + wm * 32 * MI + mi * 32 + r;
            if (m >= a.M) continue;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int nn = n0 + wn * 32 * NI + ni * 32 + 8 * g + 4 * h;
                    if (nn >= a.N) continue;
                    f32x4 v = {acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
                    *(f32x4 *)(pb + (size_t)m * a.N + nn) = v;
                }
        }
        return;
    }
    if (a.epi == 0 && a.stage_epi) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        ctx_barrier();                 // every wave is done reading the ring
        gemm_epilogue_staged<MI, NI>(a, acc, m0 + wm * 32 * MI, n0 + wn * 32 * NI, r, h, (float *)smem + wave * (32 * (32 * NI + 4)), lane);
        return;
    }
    gemm_epilogue<MI, NI>(a, acc, m0 + wm * 32 * MI, n0 + wn * 32 * NI, r, h);
}

// Split-K second pass: out = sum_s part[s] (+bias)(+rowbias)(+residual) -> f16.  Fixed summation order.
__global__ __launch_bounds__(256) void k_splitk_reduce(GemmArgs a)
{
    const int n4 = a.N / 4;
    const size_t total = (size_t)a.M * n4, MN = (size_t)a.M * a.N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        int m = (int)(i / n4), nn = (int)(i % n4) * 4;
        f32x4 v = *(const f32x4 *)(a.part + (size_t)m * a.N + nn);
        for (int sidx = 1; sidx < a.splitk; ++sidx) {
            f32x4 p = *(const f32x4 *)(a.part + sidx * MN + (size_t)m * a.N + nn);
            v += p;
        }
        if (a.bias) { f16x4 b = *(const f16x4 *)(a.bias + nn); for (int j = 0; j < 4; ++j) v[j] += (float)b[j]; }
        if (a.rowbias) {
            f16x4 b = *(const f16x4 *)(a.rowbias + (size_t)(m / a.rows_per_batch) * a.ldrb + nn);
            for (int j = 0; j < 4; ++j) v[j] += (float)b[j];
        }
        if (a.residual) {
            if (a.res32) { f32x4 b = *(const f32x4 *)((const float *)a.residual + (size_t)m * a.ldr + nn); v += b; }
            else { f16x4 b = *(const f16x4 *)(a.residual + (size_t)m * a.ldr + nn); for (int j = 0; j < 4; ++j) v[j] += (float)b[j]; }
        }
        if (a.out32) { *(f32x4 *)((float *)a.out + (size_t)m * a.ldc + nn) = v; continue; }
        f16x4 o;
        for (int j = 0; j < 4; ++j) o[j] = (f16)v[j];
        *(f16x4 *)(a.out + (size_t)m * a.ldc + nn) = o;
    }
}

// tile ids (CTX_GEMM_TILE / ctx_gemm_tune): WM x WN waves of MI x NI 32x32 blocks
//   0: 256x128 8w   1: 128x128 4w   2: 256x64 4w   3: 128x64 2w   4: 64x64 1w
//   5: 64x64 4w (32x32 per wave)   6: 64x64 2w (64x32)   7: 128x128 8w (32x64)   8: 64x128 4w (32x64)   9: 128x64 4w (64x32)
//   10-18: 64-deep K stages, ring of 3: 256x128 16w (32x64) | 256x128 16w (64x32) | 256x128 8w | 128x128 16w (32x32) | 128x128 8w |
//          64x64 4w | 64x128 4w | 128x64 4w | 128x128 4w;  19-22: ring of 2: 64x64 4w | 128x128 8w | 128x128 16w | 128x128 4w
//   23-26: deep rings: 64x64 4w x6 | 64x128 4w x5 | 128x128 8w x4 | 128x64 4w x5;  27: 256x320 8w (64x160), ring of 2
static int g_force_tile = -1, g_force_gemm8 = -1;
extern "C" void ctx_gemm_tune(int32_t tile, int32_t gemm8)
{
    g_force_tile = tile;            // -1: heuristic
    g_force_gemm8 = gemm8;          // -1: heuristic, 0: never, 1: always when applicable
}

template <int WM, int WN, int MI, int NI, bool CONV, int NS = 4, int PKT = 32>
static void launch_gemm(GemmArgs &a, hipStream_t s)
{
    constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN;
    a.ntm = cdiv(a.M, BM);
    a.ntn = cdiv(a.N, BN);
    if (a.splitk < 1 || !a.part) a.splitk = 1;
    static int mfast = -2;
    if (mfast == -2) { const char *e = getenv("CTX_GEMM_MFAST"); mfast = e ? atoi(e) : -1; }
    // unique operand bytes: weights N*K vs activations M*K (conv: M*Cin, the 9 taps re-read the same pixels)
    const double wbytes = (double)a.N * a.K, xbytes = (double)a.M * (CONV ? a.Cin : a.K);
    a.mfast = mfast >= 0 ? mfast : (wbytes > xbytes ? 1 : 0);
    static int stg = -1;
    if (stg < 0) { const char *e = getenv("CTX_GEMM_STAGE_EPI"); stg = e ? atoi(e) : 1; }
    a.stage_epi = stg && (a.ldc % 8 == 0) && (!a.residual || a.ldr % 8 == 0) && (!a.rowbias || a.ldrb % 8 == 0) &&
                  (size_t)WM * WN * 32 * (32 * NI + 4) * sizeof(float) <= 160 * 1024;       // the per-wave fp32 patch must fit the LDS
    static int steady = -1;
    if (steady < 0) { const char *e = getenv("CTX_GEMM_STEADY"); steady = e ? atoi(e) : 1; }
    a.pk = PKT; a.krot = steady ? 0 : 2;                 // krot = 2: no rotation, general K loop only (A/B switch of the steady loop)
    constexpr int NT = 64 * WM * WN;
    constexpr size_t ring = (size_t)NS * (BM + BN) * PKT * sizeof(f16);
    constexpr size_t patch = (size_t)WM * WN * 32 * (32 * NI + 4) * sizeof(float);
    constexpr size_t lds = (ring > patch || patch > 160 * 1024) ? ring : patch;
    auto kern = k_gemm_pipe<WM, WN, MI, NI, CONV, NS, PKT>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    if (ctx_prof_on()) {
        hipEvent_t e0, e1;
        ctx_prof_events(0, &e0, &e1);
        hipExtLaunchKernelGGL(kern, dim3(a.ntm * a.ntn * a.splitk), dim3(NT), lds, s, e0, e1, 0, a);
    } else
        hipLaunchKernelGGL(kern, dim3(a.ntm * a.ntn * a.splitk), dim3(NT), lds, s, a);
}

static void launch_reduce(GemmArgs &a, hipStream_t s)
{
    size_t total = (size_t)a.M * (a.N / 4);
    unsigned nb = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    if (ctx_prof_on()) {
        hipEvent_t e0, e1;
        ctx_prof_events(0, &e0, &e1);
        hipExtLaunchKernelGGL(k_splitk_reduce, dim3(nb), dim3(256), 0, s, e0, e1, 0, a);
    } else
        hipLaunchKernelGGL(k_splitk_reduce, dim3(nb), dim3(256), 0, s, a);
}

int ctx_gemm_pick_split(int M, int N, int K, int epi)
{
    static int en = -1;
    if (en < 0) { const char *e = getenv("CTX_SPLITK"); en = e ? atoi(e) : 1; }
    if (!en || epi != 0 || N % 4) return 1;
    bool wide = (N % 128 == 0);
    int tiles = wide ? cdiv(M, 128) * cdiv(N, 128) : cdiv(M, 256) * cdiv(N, 64);
    if (tiles >= 200) return 1;
    if (K < 2048 && tiles > 48) return 1;      // short-K problems: the slab round trip + extra launch costs more than it buys
    int S = 640 / tiles;                       // aim at ~2.5 workgroups per CU
    int maxS = (K / 32) / 8;                   // keep >= 8 stages per split
    if (S > maxS) S = maxS;
    if (S > 32) S = 32;
    return S < 2 ? 1 : S;
}

#include "gemm_tuned.h"
void ctx_gemm_plan(GemmArgs &a, bool conv)
{
    a.tile = -1; a.use8 = -1;
    static int use_table = -1;
    if (use_table < 0) { const char *e = getenv("CTX_GEMM_TUNED"); use_table = e ? atoi(e) : 1; }
    if (use_table) {
        const int flags = conv ? ((a.stride == 2 ? 1 : 0) | (a.ups ? 2 : 0)) : 0;
        for (const TunedGemm &t : g_tuned)
            if (t.conv == (conv ? 1 : 0) && t.M == a.M && t.N == a.N && t.K == a.K && t.flags == flags && t.epi == a.epi) {
                a.tile = t.tile; a.use8 = t.use8; a.splitk = t.splitk;
                return;
            }
    }
    // shapes outside the table: the rules the plan search kept producing (tools/tune_gemm.py, latents 96 / 64 / 32)
    if (a.K % 64 == 0 && (!conv || a.Cin % 64 == 0)) {
        const double MN = (double)a.M * a.N;
        // Large plain-epilogue problems (other lockstep batch sizes than the tuned 2 and 12, other latent sizes): the 144 x 160 kernel
        // family of gemm144.hip where it has enough tiles to fill the chip and <= 10 % masked waste along N — the 288-row form from ~512
        // tiles on, the software-pipelined 144-row form from ~220 (what the plan search chose for such shapes at batch 2 / 12).
        static const int heur144 = [] { const char *e = getenv("CTX_GEMM_HEUR144"); return e ? atoi(e) : 1; }();    // 0: the r2 rules only (A/B)
        if (heur144 && a.epi == 0 && a.N % 8 == 0 && !a.zins && !a.res32 && !a.out32) {
            const int nt = cdiv(a.N, 160);
            const double fill = (double)a.N / (160.0 * nt);
            if (fill >= 0.9) {                                  // the UNet's widths (multiples of 160); the VAE's 128 / 256 / 512 would mask a fifth of every tile (measured: the SDS loop's VAE passes lose 1.5 ms)
                if (cdiv(a.M, 288) * nt >= 512) { a.tile = -1; a.use8 = 8; a.splitk = 1; return; }
                if (cdiv(a.M, 144) * nt >= 220) { a.tile = -1; a.use8 = 6; a.splitk = 1; return; }
            }
        }
        int tile, bm, bn;
        if (MN >= 5.0e6) { tile = a.epi == 1 ? 10 : (conv ? 12 : 11); bm = 256; bn = 128; }          // 256x128, 8 / 16 waves
        else if ((MN >= 2.5e6 && a.K >= 960) || (conv && MN >= 1.2e6 && a.K >= 5760)) { tile = 14; bm = 128; bn = 128; }
        else if (a.epi == 1) { tile = 16; bm = 64; bn = 128; }                                        // GEGLU needs 64-wide wave tiles
        else { tile = 15; bm = 64; bn = 64; }                                                         // 64x64, 4 waves
        const int tiles = cdiv(a.M, bm) * cdiv(a.N, bn);
        int S = 1;
        if (a.epi == 0 && a.N % 4 == 0) {
            S = 320 / tiles;                                   // ~1.25 workgroups per CU
            const int maxS = a.K / 64 / 4;                     // >= 4 stages per slice
            if (S > maxS) S = maxS;
            if (S > 12) S = 12;
            if (S < 1) S = 1;
        }
        a.tile = tile; a.splitk = S;
        a.use8 = (a.epi == 1 && MN >= 1.0e7) ? -1 : 0;        // the 256x256 kernel only where its own heuristic wants it
        return;
    }
    a.splitk = ctx_gemm_pick_split(a.M, a.N, a.K, a.epi);
}

// Tile choice: the largest tile (most flops per staged byte) that still gives the chip >= ~1.5 workgroups per CU once
// split-K is counted; small problems fall through to tiles with more waves per staged byte.
int ctx_gemm_dispatch(GemmArgs &a, bool conv, hipStream_t s)
{
    const bool only_pipe = a.zins || a.res32 || a.out32;                     // features of this file's kernel only
    if (only_pipe) a.use8 = 0;
    const int want8 = only_pipe ? 0 : (g_force_gemm8 >= 0 ? g_force_gemm8 : a.use8);          // -1: gemm8's own heuristic
    const int want_tile = g_force_tile >= 0 ? g_force_tile : (g_force_gemm8 >= 0 ? -1 : a.tile);
    if (want_tile < 0 && want8 >= 4 && want8 <= 8 && ctx_gemm144_try(a, conv, want8 - 4, s)) {
        if (a.splitk > 1) launch_reduce(a, s);
    } else if (want_tile < 0 && conv && (want8 == 2 || want8 == 3) && ctx_conv_halo_try(a, want8 == 2 ? 2 : 1, s)) {
        if (a.splitk > 1) launch_reduce(a, s);
    } else if (want_tile < 0 && want8 != 0 && want8 < 2 && ctx_gemm8_try(a, conv, want8 == 1, s)) {
        if (a.splitk > 1) launch_reduce(a, s);
    } else {
        static int big = -1;
        if (big < 0) { const char *e = getenv("CTX_GEMM_BIG"); big = e ? atoi(e) : 1; }
        const int S = a.splitk > 1 && a.part ? a.splitk : 1;
        auto wgs = [&](int bm, int bn) { return cdiv(a.M, bm) * cdiv(a.N, bn) * S; };
        const bool n128 = (a.N % 128 == 0);
        int pick;
        static int bigk = -1;
        if (bigk < 0) { const char *e = getenv("CTX_GEMM_BIGK"); bigk = e ? atoi(e) : 1024; }
        if (big && a.M >= 8192 && a.N >= 256 && a.K >= bigk) pick = 0;                 // 256x128, 8 waves
        else if (n128 && wgs(128, 128) >= 384) pick = 1;                               // 128x128
        else if (!n128 && wgs(256, 64) >= 384) pick = 2;                               // 256x64
        else if (wgs(128, 64) >= 320) pick = 3;                                        // 128x64, 2 waves
        else if (S > 1) pick = n128 ? 1 : 2;                                           // split-K already spreads it
        else pick = 4;                                                                 // 64x64, 1 wave
        static int force = -2;
        if (force == -2) { const char *e = getenv("CTX_GEMM_TILE"); force = e ? atoi(e) : -1; }
        if (force >= 0) pick = force;
        if (want_tile >= 0) pick = want_tile;
        if (a.epi == 1 && (pick == 5 || pick == 6 || pick == 9 || pick == 11 || pick == 13 || pick == 15 || pick == 17 || pick == 19 || pick == 21 || pick == 23 || pick == 26 || pick == 27)) pick = 1;   // GEGLU needs 64-wide wave tiles
        if (pick >= 10 && (a.K % 64 != 0 || (conv && a.Cin % 64 != 0))) pick = 1;   // 64-deep stages
#define CTX_LAUNCH(WM_, WN_, MI_, NI_) do { if (conv) launch_gemm<WM_, WN_, MI_, NI_, true>(a, s); else launch_gemm<WM_, WN_, MI_, NI_, false>(a, s); } while (0)
        switch (pick) {
        case 0: CTX_LAUNCH(4, 2, 2, 2); break;
        case 1: CTX_LAUNCH(2, 2, 2, 2); break;
        case 2: CTX_LAUNCH(4, 1, 2, 2); break;
        case 3: CTX_LAUNCH(2, 1, 2, 2); break;
        case 4: CTX_LAUNCH(1, 1, 2, 2); break;
        case 5: CTX_LAUNCH(2, 2, 1, 1); break;
        case 6: CTX_LAUNCH(1, 2, 2, 1); break;
        case 7: CTX_LAUNCH(4, 2, 1, 2); break;
        case 8: CTX_LAUNCH(2, 2, 1, 2); break;
        case 9: CTX_LAUNCH(2, 2, 2, 1); break;
#define CTX_LAUNCH64(WM_, WN_, MI_, NI_) do { if (conv) launch_gemm<WM_, WN_, MI_, NI_, true, 3, 64>(a, s); else launch_gemm<WM_, WN_, MI_, NI_, false, 3, 64>(a, s); } while (0)
        case 10: CTX_LAUNCH64(8, 2, 1, 2); break;     // 256x128, 16 waves (32x64), 64-deep stages
        case 11: CTX_LAUNCH64(4, 4, 2, 1); break;     // 256x128, 16 waves (64x32)
        case 12: CTX_LAUNCH64(4, 2, 2, 2); break;     // 256x128, 8 waves
        case 13: CTX_LAUNCH64(4, 4, 1, 1); break;     // 128x128, 16 waves (32x32)
        case 14: CTX_LAUNCH64(4, 2, 1, 2); break;     // 128x128, 8 waves (32x64)
        case 15: CTX_LAUNCH64(2, 2, 1, 1); break;     // 64x64, 4 waves (32x32)
        case 16: CTX_LAUNCH64(2, 2, 1, 2); break;     // 64x128, 4 waves (32x64)
        case 17: CTX_LAUNCH64(2, 2, 2, 1); break;     // 128x64, 4 waves (64x32)
        case 18: CTX_LAUNCH64(2, 2, 2, 2); break;     // 128x128, 4 waves
#define CTX_LAUNCH64N2(WM_, WN_, MI_, NI_) do { if (conv) launch_gemm<WM_, WN_, MI_, NI_, true, 2, 64>(a, s); else launch_gemm<WM_, WN_, MI_, NI_, false, 2, 64>(a, s); } while (0)
        case 19: CTX_LAUNCH64N2(2, 2, 1, 1); break;   // ring of 2 (half the LDS, more workgroups per CU): 64x64, 4 waves
        case 20: CTX_LAUNCH64N2(4, 2, 1, 2); break;   // 128x128, 8 waves
        case 21: CTX_LAUNCH64N2(4, 4, 1, 1); break;   // 128x128, 16 waves
        case 22: CTX_LAUNCH64N2(2, 2, 2, 2); break;   // 128x128, 4 waves
        // deep rings (one workgroup per CU, 4-5 stages in flight) for the weight-streaming layers of the two deepest levels,
        // whose few workgroups wait on HBM latency rather than on staging bandwidth
#define CTX_LAUNCH64NS(NS_, WM_, WN_, MI_, NI_) do { if (conv) launch_gemm<WM_, WN_, MI_, NI_, true, NS_, 64>(a, s); else launch_gemm<WM_, WN_, MI_, NI_, false, NS_, 64>(a, s); } while (0)
        case 23: CTX_LAUNCH64NS(6, 2, 2, 1, 1); break;   // 64x64, 4 waves, ring of 6
        case 24: CTX_LAUNCH64NS(5, 2, 2, 1, 2); break;   // 64x128, 4 waves, ring of 5
        case 25: CTX_LAUNCH64NS(4, 4, 2, 1, 2); break;   // 128x128, 8 waves, ring of 4
        case 26: CTX_LAUNCH64NS(5, 2, 2, 2, 1); break;   // 128x64, 4 waves, ring of 5
        default: CTX_LAUNCH64N2(4, 2, 2, 5); break;      // 27: 256x320, 8 waves (64x160 each), ring of 2: 142 FLOP per staged byte (the lockstep batch)
#undef CTX_LAUNCH64NS
#undef CTX_LAUNCH64N2
#undef CTX_LAUNCH64
        }
#undef CTX_LAUNCH
        if (a.splitk > 1) launch_reduce(a, s);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ctx_set_error("gemm launch failed: %s", hipGetErrorString(e));
        return CTX_E_LAUNCH;
    }
    return CTX_OK;
}

extern "C" int32_t ctx_gemm_f16(const void *A, const void *Wt, const void *bias, const void *residual, int32_t M,
                                int32_t N, int32_t K, void *C, ctx_stream_t stream)
{
    CTX_REQUIRE(A && Wt && C, "gemm: null pointer");
    CTX_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0 && N % 8 == 0, "gemm: need K%%64==0, N%%8==0 (M=%d N=%d K=%d)", M, N, K);
    CTX_REQUIRE((int64_t)M * K < (1ll << 31) && (int64_t)N * K < (1ll << 31), "gemm: operand too large for 32-bit offsets");
    GemmArgs a = {};
    a.X = (const f16 *)A; a.Wt = (const f16 *)Wt; a.bias = (const f16 *)bias; a.residual = (const f16 *)residual;
    a.out = (f16 *)C; a.M = M; a.N = N; a.K = K; a.ldc = N; a.ldr = N; a.rows_per_batch = 1; a.ldrb = N; a.epi = 0;
    a.tile = -1; a.use8 = -1;
    return ctx_gemm_dispatch(a, false, (hipStream_t)stream);
}

extern "C" int32_t ctx_conv3x3_f16(const void *x, const void *w, const void *bias, const void *rowbias,
                                   const void *residual, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                                   int32_t stride, int32_t upsample, void *y, ctx_stream_t stream)
{
    CTX_REQUIRE(x && w && y, "conv3x3: null pointer");
    CTX_REQUIRE(B > 0 && H > 0 && W > 0 && Cin % 64 == 0 && Cout % 8 == 0 && (stride == 1 || stride == 2) &&
                    (upsample == 0 || upsample == 1) && !(upsample && stride == 2),
                "conv3x3: need Cin%%64==0, Cout%%8==0, stride 1|2 (B=%d H=%d W=%d Cin=%d Cout=%d s=%d up=%d)", B, H, W, Cin, Cout, stride, upsample);
    GemmArgs a = {};
    int Hv = H << upsample, Wv = W << upsample;
    a.Ho = (Hv + 2 - 3) / stride + 1; a.Wo = (Wv + 2 - 3) / stride + 1;
    a.X = (const f16 *)x; a.Wt = (const f16 *)w; a.bias = (const f16 *)bias; a.rowbias = (const f16 *)rowbias;
    a.residual = (const f16 *)residual; a.out = (f16 *)y;
    a.M = B * a.Ho * a.Wo; a.N = Cout; a.K = 9 * Cin; a.ldc = Cout; a.ldr = Cout; a.rows_per_batch = a.Ho * a.Wo; a.ldrb = Cout; a.epi = 0;
    a.H = H; a.W = W; a.Cin = Cin; a.stride = stride; a.ups = upsample;
    a.tile = -1; a.use8 = -1;
    CTX_REQUIRE((int64_t)B * H * W * Cin < (1ll << 31) && (int64_t)Cout * a.K < (1ll << 31), "conv3x3: tensor too large for 32-bit offsets");
    return ctx_gemm_dispatch(a, true, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// Small-M path (time embedding MLP, per-resnet temb projections; M = CFG batch = 2): weight-streaming
// GEMV, one wave per output feature, 16-byte loads along K, fp32 accumulate.
//   out[b,n] = act( sum_k x[b,k] * w[n,k] + bias[n] ),  x optionally SiLU'd on load.
// Two output features per wave and up to GV_U 16-byte chunks per lane per feature, all loaded before the first FMA: the
// time_emb_proj GEMV streams 52 MB of weights per UNet evaluation and is bound by the bytes it keeps in flight.
#define GV_U 4
__global__ __launch_bounds__(256) void k_gemv_f16(const f16 *__restrict__ x, const f16 *__restrict__ w,
                                                  const f16 *__restrict__ bias, int Bm, int N, int K, int silu_in,
                                                  int silu_out, f16 *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int n0 = ((blockIdx.x * 256 + threadIdx.x) >> 6) * 2;
    if (n0 >= N) return;
    const int n1 = n0 + 1 < N ? n0 + 1 : n0;
    float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 512 * GV_U) {
        f16x8 wv[2][GV_U];
#pragma unroll
        for (int u = 0; u < GV_U; ++u) {
            const int k = k0 + u * 512 + lane * 8;
            const int kc = k < K ? k : K - 8;                              // unconditional (clamped) loads, masked below
            wv[0][u] = *(const f16x8 *)(w + (size_t)n0 * K + kc);
            wv[1][u] = *(const f16x8 *)(w + (size_t)n1 * K + kc);
        }
#pragma unroll
        for (int u = 0; u < GV_U; ++u) {
            const int k = k0 + u * 512 + lane * 8;
            if (k >= K) { wv[0][u] = zero8; wv[1][u] = zero8; }
            const int kc = k < K ? k : K - 8;
            for (int b = 0; b < Bm; ++b) {
                f16x8 xv = *(const f16x8 *)(x + (size_t)b * K + kc);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float xf = (float)xv[j];
                    if (silu_in) xf = xf / (1.0f + __expf(-xf));
                    acc[0][b] += xf * (float)wv[0][u][j];
                    acc[1][b] += xf * (float)wv[1][u][j];
                }
            }
        }
    }
    for (int r = 0; r < 2; ++r) {
        const int n = n0 + r;
        for (int b = 0; b < Bm; ++b) {
            float v = wave_sum(acc[r][b]);
            if (lane == 0 && n < N) {
                if (bias) v += (float)bias[n];
                if (silu_out) v = v / (1.0f + __expf(-v));
                out[(size_t)b * N + n] = (f16)v;
            }
        }
    }
}

int ctx_gemv_f16(const f16 *x, const f16 *w, const f16 *bias, int Bm, int N, int K, int silu_in, int silu_out, f16 *out,
                 hipStream_t s)
{
    if (Bm < 1 || K % 8 != 0 || K < 8) {
        ctx_set_error("gemv: Bm=%d (>=1) K=%d (%%8)", Bm, K);
        return CTX_E_ARG;
    }
    for (int b0 = 0; b0 < Bm; b0 += 4) {                       // four rows per pass over the weights
        const int nb = Bm - b0 < 4 ? Bm - b0 : 4;
        hipLaunchKernelGGL(k_gemv_f16, dim3(cdiv(N, 8)), dim3(256), 0, s, x + (size_t)b0 * K, w, bias, nb, N, K, silu_in, silu_out,
                           out + (size_t)b0 * N);
    }
    return CTX_OK;
}

// ------------------------------------------------------------------------------------------------
// Device-timed repeat launcher for tools/bench_gemm.py (the Python call overhead of ~10 us per launch hides the
// real duration of the small layers).  Returns the average milliseconds per launch, or a negative error code.
extern "C" float ctx_bench_gemm(const void *A, const void *Wt, const void *bias, const void *residual, int32_t M, int32_t N,
                                int32_t K, void *C, int32_t conv_B, int32_t conv_H, int32_t conv_W, int32_t conv_Cin,
                                int32_t conv_flags, int32_t epi, void *part, int32_t splitk, int32_t iters, ctx_stream_t stream)
{
    hipStream_t s = (hipStream_t)stream;
    GemmArgs a = {};
    a.X = (const f16 *)A; a.Wt = (const f16 *)Wt; a.bias = (const f16 *)bias; a.residual = (const f16 *)residual; a.out = (f16 *)C;
    bool conv = conv_B > 0;
    if (conv) {
        a.stride = (conv_flags & 1) ? 2 : 1; a.ups = (conv_flags & 2) ? 1 : 0;
        int Hv = conv_H << a.ups, Wv = conv_W << a.ups;
        a.H = conv_H; a.W = conv_W; a.Cin = conv_Cin; a.Ho = (Hv - 1) / a.stride + 1; a.Wo = (Wv - 1) / a.stride + 1;
        a.M = conv_B * a.Ho * a.Wo; a.N = N; a.K = 9 * conv_Cin; a.rows_per_batch = a.Ho * a.Wo;
    } else {
        a.M = M; a.N = N; a.K = K; a.rows_per_batch = 1;
    }
    a.epi = epi;
    a.ldc = epi == 1 ? a.N / 2 : a.N; a.ldr = a.N; a.ldrb = a.N;
    a.tile = -1; a.use8 = -1;
    if (splitk < 0) {                                       // what the UNet executor would choose
        if (part) { ctx_gemm_plan(a, conv); splitk = a.splitk; } else splitk = 1;
    }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < iters + 2; ++i) {
        if (i == 2) (void)hipEventRecord(e0, s);
        GemmArgs b = a;
        b.splitk = splitk; b.part = (float *)part;
        int rc = ctx_gemm_dispatch(b, conv, s);
        if (rc) return (float)rc;
    }
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms / iters;
}
