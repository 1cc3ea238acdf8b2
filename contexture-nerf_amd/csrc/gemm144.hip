// 144x160x64 fp16 MFMA GEMM / implicit-GEMM 3x3 convolution (gfx950): the tile that divides the UNet evenly over 256 CUs.
//
//   out[M,N] = X[M,K] . Wt[N,K]^T  (+bias)(+rowbias)(+residual) | fp32 split-K partials
//
// Replaces the cuBLAS / cuDNN calls under diffusers' UNet2DConditionModel (reference call site
// src/stable_diffusion_depth.py:422-423).  Why this shape: at the reference's 768^2 image (latent 96^2, CFG batch 2) the four
// UNet levels have M = 18432 / 4608 / 1152 / 288 tokens and N = 320 k features.  18432 x 320 outputs are exactly 256 tiles of
// 144 x 160 (one per CU, nothing masked), and the deeper levels are 128 / 64 / 16 such tiles: split-K by 2 / 4 / 16 gives 256
// workgroups again.  The 128- and 256-wide tiles of gemm.hip / gemm8.hip leave 16-30 % of the CUs idle or of the tile masked
// on these shapes (N = 320 is 2.5 tiles of 128), and their time is set by the bytes a CU stages per K step.
//
// Structure: WM x WN waves over the tile's 9 x 10 blocks of 16 x 16: 3 x 2 (wave tile 48 tokens x 80 features, 15 accumulators
// of v_mfma_f32_16x16x32_f16) or 3 x 5 (48 x 32, 6 accumulators, 15 waves: 4 / 4 / 4 / 3 per SIMD); weights = A operand,
// activations = B operand, so a lane owns one token and 4 consecutive features per accumulator.  Operand tiles go global -> LDS
// by global_load_lds_dwordx4 in 1-KiB pieces (8 rows x 128 B; 18 activation + 20 weight pieces per 64-deep K stage, dealt
// round-robin to the waves: a wave has SL or SL - 1 of them and waits with its own immediate vmcnt count), ring of 3 stages,
// one raw s_barrier per stage.
// LDS image and fragment reads as gemm8.hip (16-byte chunk index XORed with (row>>1)&7 on the DMA source and on the read).
#include "common.h"
#include "kernels.h"
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <type_traits>

typedef const __attribute__((address_space(1))) void *g144_gptr_t;
typedef __attribute__((address_space(3))) void *g144_lptr_t;
__device__ __attribute__((aligned(128))) f16 g144_zero[64];

#define G144_BM 144
#define G144_BN 160
#define G144_STAGE ((G144_BM + G144_BN) * 64)      // f16 per stage

__device__ __forceinline__ int g144_xcd_remap(int bid, int nwg)
{
    int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// UPS: the convolution has the fused nearest x2 upsample (generic per-tap address arithmetic); without it a tap is one scalar
// offset from the lane's centre-tap pointer and a bit of a 9-bit in-bounds mask
// BMB: token rows of the tile in 16-row blocks: 9 (144 x 160) or 18 (288 x 160: 103 instead of 76 FLOP per staged byte, for problems
// with >= ~512 such tiles, i.e. the lockstep batch; its 56 KB stages leave room for a ring of 2 only, so it runs the lockstep schedule)
template <int WM, int WN, bool CONV, int NS, int VAR, bool UPS = false, int BMB = 9>
__global__ __launch_bounds__(64 * WM * WN) void k_gemm144(GemmArgs a)
{
    constexpr int NW = WM * WN, MI = BMB / WM, NI = 10 / WN;        // waves; 16x16 blocks per wave along tokens / features
    constexpr int BM = 16 * BMB, NPX = BM / 8;                      // tile rows; activation pieces (8 rows x 128 B each)
    constexpr int NP = NPX + 20;                                    // DMA pieces per stage: activation + 20 weight
    constexpr int STAGE = (BM + G144_BN) * 64;                      // f16 per stage
    static_assert(BMB % WM == 0 && (BMB == 9 || (BMB == 18 && VAR == 0)), "288-row tiles: lockstep schedule only");
    constexpr int SL = (NP + NW - 1) / NW;                          // piece slots per wave (the last one empty on some waves)
    static_assert(9 % WM == 0 && 10 % WN == 0, "wave grid must divide 9 x 10 blocks");
    extern __shared__ __attribute__((aligned(16))) f16 smem[];     // [NS][144 X rows | 160 W rows][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: SGPR, scalar branches
    const int wm = wave / WN, wn = wave % WN;
    const int r16 = lane & 15, kg = lane >> 4;

    const int ntiles = a.ntm * a.ntn;
    const int lin = g144_xcd_remap(blockIdx.x, ntiles * a.splitk);
    const int slice = lin / ntiles, bid = lin - slice * ntiles;
    const int tile_n = a.mfast ? bid / a.ntm : bid % a.ntn, tile_m = a.mfast ? bid % a.ntm : bid / a.ntn;
    const int m0 = tile_m * BM, n0 = tile_n * G144_BN;
    const int nk_all = a.K / 64;
    const int kbeg = (int)((long)nk_all * slice / a.splitk);
    const int nk = (int)((long)nk_all * (slice + 1) / a.splitk) - kbeg;

    // ---- DMA state: piece p = wave + NW i of the stage image; p < 18 activation rows 8p.., else weight rows 8(p-18).. -----
    const int prow = lane >> 3, pc = lane & 7;
    constexpr bool FAST = CONV && !UPS;
    const f16 *pp[SL], *xcen[SL];
    int pst[SL], xoff[SL], xoy[SL], xox[SL], xlc[SL], xval[SL];
    bool xok[SL];
    const f16 *const zp = g144_zero;
    const bool full = wave + NW * (SL - 1) < NP;                    // this wave uses its last slot
#pragma unroll
    for (int i = 0; i < SL; ++i) {
        const int p = wave + NW * i;
        const bool isx = p < NPX, live = p < NP;
        const int s = isx ? 8 * p + prow : 8 * (p - NPX) + prow;     // row inside the activation / weight tile
        xlc[i] = (pc ^ ((s >> 1) & 7)) * 8;
        xoff[i] = 0; xoy[i] = 0; xox[i] = 0;
        if (isx) {
            const int m = m0 + s;
            xok[i] = m < a.M;
            if (CONV) {
                const int hw = a.Ho * a.Wo;
                const int mm = xok[i] ? m : 0;
                const int b = mm / hw, q = mm - b * hw;
                const int oy = q / a.Wo, ox = q - oy * a.Wo;
                xoy[i] = oy * a.stride; xox[i] = ox * a.stride;
                xoff[i] = b * a.H * a.W * a.Cin + xlc[i];
                pp[i] = zp; pst[i] = 0;
                if (FAST) {
                    xcen[i] = a.X + (size_t)xoff[i] + (size_t)(xoy[i] * a.W + xox[i]) * a.Cin;
                    int v = 0;
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int iy = xoy[i] + t / 3 - 1 + a.poff, ix = xox[i] + t % 3 - 1 + a.poff;
                        if (xok[i] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v |= 1 << t;
                    }
                    xval[i] = v;
                }
            } else {
                pp[i] = xok[i] ? a.X + (size_t)m * a.K + (size_t)kbeg * 64 + xlc[i] : g144_zero;
                pst[i] = xok[i] ? 64 : 0;
            }
        } else {
            const bool ok = live && (n0 + s) < a.N;
            xok[i] = ok;
            pp[i] = ok ? a.Wt + (size_t)(n0 + s) * a.K + (size_t)kbeg * 64 + xlc[i] : g144_zero;
            pst[i] = ok ? 64 : 0;
        }
    }

    int k_issue = kbeg * 64, issued = 0, tap_left = 0;
    int cur_tap = CONV ? (kbeg * 64) / a.Cin : 0, cur_c0 = CONV ? kbeg * 64 - cur_tap * a.Cin : 0;   // FAST: tap and first channel of the next stage
    auto retap = [&]() {                                            // CONV: new 3x3 tap -> recompute the activation pointers
        if (FAST) {
            const int dy = cur_tap / 3 - 1 + a.poff, dx = cur_tap - (cur_tap / 3) * 3 - 1 + a.poff;
            const int soff = (dy * a.W + dx) * a.Cin + cur_c0;     // scalar: the same for every lane
#pragma unroll
            for (int i = 0; i < SL; ++i) {
                if (wave + NW * i >= NPX) continue;
                const bool v = (xval[i] >> cur_tap) & 1;
                pp[i] = v ? xcen[i] + soff : zp;
                pst[i] = v ? 64 : 0;
            }
            tap_left = (a.Cin - cur_c0) >> 6;
            return;
        }
        const int tap = k_issue / a.Cin, c0 = k_issue - tap * a.Cin;
        const int dy = tap / 3 - 1 + a.poff, dx = tap % 3 - 1 + a.poff;
        const int Hv = a.H << a.ups, Wv = a.W << a.ups;
#pragma unroll
        for (int i = 0; i < SL; ++i) {
            if (wave + NW * i >= NPX) continue;
            const int iy = xoy[i] + dy, ix = xox[i] + dx;
            const bool ok = xok[i] && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
            pp[i] = ok ? a.X + xoff[i] + (((iy >> a.ups) * a.W + (ix >> a.ups)) * a.Cin) + c0 : zp;
            pst[i] = ok ? 64 : 0;
        }
        tap_left = (a.Cin - c0) / 64;
    };
    auto issue_begin = [&]() {
        if (CONV) {
            if (tap_left == 0) retap();
            --tap_left;
        }
    };
    auto issue_slot = [&](int buf, int i) {
        const int p = wave + NW * i;
        if (p < NP) {
            __builtin_amdgcn_global_load_lds((g144_gptr_t)pp[i], (g144_lptr_t)(smem + buf * STAGE + p * 512), 16, 0, 0);
            pp[i] += pst[i];
        }
    };
    auto issue_end = [&]() {
        k_issue += 64; ++issued;
        if (FAST) { cur_c0 += 64; if (cur_c0 == a.Cin) { cur_c0 = 0; ++cur_tap; } }
    };
    auto issue = [&](int buf) {
        issue_begin();
#pragma unroll
        for (int i = 0; i < SL; ++i) issue_slot(buf, i);
        issue_end();
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int swz = (r16 >> 1) & 7;
    const int ck[2] = {(kg ^ swz) * 8, ((4 + kg) ^ swz) * 8};
    const int xrow = (wm * 16 * MI + r16) * 64, wrow = BM * 64 + (wn * 16 * NI + r16) * 64;

#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < nk) issue(p);
    // wait until this wave's pieces of every stage but the `newer` most recent ones have landed
    auto wait_newer = [&](int newer) {
        if (newer >= 2) {
            if (full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * SL) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (SL - 1)) : "memory");
        } else if (newer >= 1) {
            if (full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SL) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SL - 1) : "memory");
        } else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    int buf = 0;
    if constexpr (VAR == 0 && BMB == 18) {
        // lockstep with the stage's own work interleaved (the 288-row tile: ring of 2, so nothing of the NEXT stage can be touched
        // early): k 0..31 fragments first, then one DMA piece of the next stage behind each MFMA row instead of a burst of four at the
        // barrier (the vector-memory port queues them: the attention kernel's time stamps), and the k 32..63
        // fragments read under the first half's MFMAs.
        f16x8 xa[MI], wa[NI], wb[NI];                               // the k 32..63 token fragments reuse xa row by row (128 VGPRs at 15 waves)
        auto mrow = [&](const f16x8 &xf, const f16x8 (&wf)[NI], int i) {
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf, acc[i][j], 0, 0, 0);
        };
        static_assert(BMB != 18 || SL <= MI, "one DMA piece per MFMA row");
        for (int kt = 0; kt < nk; ++kt) {
            wait_newer(issued - 1 - kt);                            // this wave's pieces of stage kt
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ctx_barrier();
            const bool go = issued < nk;
            const int pb = buf == 0 ? NS - 1 : buf - 1;
            const f16 *sb = smem + buf * STAGE;
#pragma unroll
            for (int i = 0; i < MI; ++i) xa[i] = *(const f16x8 *)(sb + xrow + i * 1024 + ck[0]);
#pragma unroll
            for (int j = 0; j < NI; ++j) wa[j] = *(const f16x8 *)(sb + wrow + j * 1024 + ck[0]);
            if (go) issue_begin();
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                mrow(xa[i], wa, i);
                __builtin_amdgcn_sched_barrier(0);
                xa[i] = *(const f16x8 *)(sb + xrow + i * 1024 + ck[1]);       // the row just consumed takes its second half
                if (i == 0) {
#pragma unroll
                    for (int j = 0; j < NI; ++j) wb[j] = *(const f16x8 *)(sb + wrow + j * 1024 + ck[1]);
                }
                if (go && i < SL) issue_slot(pb, i);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (go) issue_end();
#pragma unroll
            for (int i = 0; i < MI; ++i) mrow(xa[i], wb, i);
            buf = buf == NS - 1 ? 0 : buf + 1;
        }
    } else if constexpr (VAR == 0) {
        // lockstep schedule: one barrier per stage, every wave issues, reads and multiplies in the same order
        for (int kt = 0; kt < nk; ++kt) {
            wait_newer(issued - 1 - kt);                            // this wave's pieces of stage kt
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the barrier does not wait for LDS reads in flight (stage kt-1's)
            ctx_barrier();                                          // everybody's pieces landed AND everybody finished stage kt-1
            if (issued < nk) issue(buf == 0 ? NS - 1 : buf - 1);   // into the buffer stage kt-1 used
            const f16 *sb = smem + buf * STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 xf[MI], wf[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) xf[i] = *(const f16x8 *)(sb + xrow + i * 1024 + ck[ks]);
#pragma unroll
                for (int j = 0; j < NI; ++j) wf[j] = *(const f16x8 *)(sb + wrow + j * 1024 + ck[ks]);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
            }
            buf = buf == NS - 1 ? 0 : buf + 1;
        }
    } else if constexpr (VAR == 2) {
        // One barrier per TWO stages (ring of 4 = two pairs of buffers): PMC on the forms above has the waves 42 % of their time at
        // s_waitcnt / s_barrier with 15 waves meeting every 64 K elements.  Iteration j: the pair's pieces have landed (vmcnt(0):
        // they were issued one whole iteration ago), barrier, the next pair's pieces go into the buffers pair j-1 just left, one
        // after each MFMA row, and the four k-32 quarters of the pair are multiplied with the next quarter's fragments read under
        // the current quarter's MFMAs.
        static_assert(VAR != 2 || NS == 4, "written for a ring of 4");
        f16x8 xa[MI], wa[NI], xb[MI], wb[NI];
        auto rd = [&](f16x8 (&xf)[MI], f16x8 (&wf)[NI], const f16 *sb, int ks) {
#pragma unroll
            for (int i = 0; i < MI; ++i) xf[i] = *(const f16x8 *)(sb + xrow + i * 1024 + ck[ks]);
#pragma unroll
            for (int j = 0; j < NI; ++j) wf[j] = *(const f16x8 *)(sb + wrow + j * 1024 + ck[ks]);
        };
        auto mrow = [&](const f16x8 &xf, const f16x8 (&wf)[NI], int i) {
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf, acc[i][j], 0, 0, 0);
        };
        static_assert(SL <= MI, "one DMA piece per MFMA row of a quarter");
        if (issued < nk) issue(issued & 3);                         // stage 3 (0 .. 2 went above): the prologue holds two whole pairs
        // `steady`: a whole pair with both of the next pair's stages still to issue: no wave-uniform branches in the body
        auto pair = [&](int k0, auto steady) {
            constexpr bool ST = decltype(steady)::value;
            const int b0 = k0 & 3;                                  // buffers of this pair: b0, b0 + 1
            const bool two = ST || k0 + 1 < nk;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            ctx_barrier();                           // pair landed for everybody; the previous pair's buffers are free
            // quarter q of the pair: buffer b0 + (q >> 1), k half q & 1
            rd(xa, wa, smem + b0 * STAGE, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q >= 2 && !two) break;
                const bool even = (q & 1) == 0;
                if (q + 1 < (two ? 4 : 2)) {
                    const f16 *sn = smem + (b0 + ((q + 1) >> 1)) * STAGE;
                    if (even) rd(xb, wb, sn, (q + 1) & 1); else rd(xa, wa, sn, (q + 1) & 1);
                }
                // the next pair's two stages go into the buffers of the previous pair, one stage per even quarter
                const bool dma = even && (ST || (issued < nk && issued <= k0 + 3));
                if (dma) issue_begin();
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    if (even) mrow(xa[i], wa, i); else mrow(xb[i], wb, i);
                    __builtin_amdgcn_sched_barrier(0);
                    if (dma && i < SL) issue_slot(issued & 3, i);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (dma) issue_end();
            }
        };
        using T1 = std::integral_constant<bool, true>;
        using T0 = std::integral_constant<bool, false>;
        int k0 = 0;
        if (k0 < nk) { pair(k0, T0{}); k0 += 2; }                   // pair 0: the next pair was issued by the prologue
        for (; k0 + 3 < nk; k0 += 2) pair(k0, T1{});                // stages k0 + 2, k0 + 3 exist and are issued here
        for (; k0 < nk; k0 += 2) pair(k0, T0{});
    } else {
        // Software-pipelined schedule (ring of 4).  Measured on the lockstep schedule (each leg removed in turn): a 64-deep stage
        // costs ~1670 clk, of which DMA, fragment reads and MFMA each expose ~400 and ~390 are fixed: after the barrier every wave
        // first issues its DMA pieces, then waits for the fragment reads it has just issued, and only then multiplies.  Here the
        // first half of a stage's fragments (k 0..31) is read during the PREVIOUS stage's last MFMAs, so after the barrier the
        // MFMAs start at once, and the DMA pieces and the remaining reads ride between them:
        //   iteration kt:  vmcnt: own pieces of stage kt+1 landed;  lgkmcnt(0): half-fragments A of stage kt are in registers
        //                  s_barrier  -> everybody's pieces of kt+1 landed, everybody is done reading buffer kt-1
        //                  MFMA(A) x MI rows, one DMA piece of stage kt+3 (into buffer kt-1) after each row, read B (k 32..63)
        //                  MFMA(B), read A' (stage kt+1, k 0..31) under them
        // (Tried first on this tile and dropped, both neutral: two wave groups one barrier apart as in gemm8.hip; whole-stage
        // register double buffering, which spills at 128 VGPRs.)
        static_assert(VAR == 0 || NS == 4, "written for a ring of 4");
        wait_newer(issued - 1);                                     // stage 0 (stages 1, 2 may fly)
        ctx_barrier();
        f16x8 xp_[MI], wp_[NI], xq[MI], wq[NI], xb[MI], wb[NI];
        auto rd = [&](f16x8 (&xf)[MI], f16x8 (&wf)[NI], const f16 *sb, int ks) {
#pragma unroll
            for (int i = 0; i < MI; ++i) xf[i] = *(const f16x8 *)(sb + xrow + i * 1024 + ck[ks]);
#pragma unroll
            for (int j = 0; j < NI; ++j) wf[j] = *(const f16x8 *)(sb + wrow + j * 1024 + ck[ks]);
        };
        auto mrow = [&](const f16x8 &xf, const f16x8 (&wf)[NI], int i) {
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf, acc[i][j], 0, 0, 0);
        };
        rd(xp_, wp_, smem, 0);
        // cur = half-fragments A of this stage (loaded), nxt = where the next stage's go
        // `steady` (a std::integral_constant): the loop's middle, where a stage is still to be issued (go), a next stage exists
        // (more) and exactly one newer stage is in flight at the wait — none of the wave-uniform branches those conditions cost
        // in the general form (ten s_cbranch per stage in the ISA), only the wave's own piece count remains
        auto stage = [&](f16x8 (&xc)[MI], f16x8 (&wc)[NI], f16x8 (&xn)[MI], f16x8 (&wn_)[NI], int kt, auto steady) {
            constexpr bool ST = decltype(steady)::value;
            if (ST) {
                if (full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SL) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SL - 1) : "memory");
            } else
                wait_newer(issued - 1 - (kt + 1));                  // own pieces of stage kt+1
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // A of this stage has arrived; nothing of stage kt-1 is pending
            ctx_barrier();
            const bool go = ST || issued < nk;
            const int pb = buf == 0 ? NS - 1 : buf - 1;             // buffer of stage kt-1: takes stage kt+3
            const int nb_ = buf == NS - 1 ? 0 : buf + 1;
            const f16 *sc = smem + buf * STAGE, *sn = smem + nb_ * STAGE;
            const bool more = ST || kt + 1 < nk;
            if (go) issue_begin();
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                mrow(xc[i], wc, i);
                __builtin_amdgcn_sched_barrier(0);
                if (i == 0) rd(xb, wb, sc, 1);
                if (go && i < SL) issue_slot(pb, i);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (go) {
#pragma unroll
                for (int i = MI; i < SL; ++i) issue_slot(pb, i);
                issue_end();
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                mrow(xb[i], wb, i);
                if (i == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) rd(xn, wn_, sn, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            buf = nb_;
        };
        using T1 = std::integral_constant<bool, true>;
        using T0 = std::integral_constant<bool, false>;
        int kt = 0;
        // steady pairs: stages kt and kt+1 both issue a stage (kt + 4 < nk) ...
        for (; kt + 4 < nk; kt += 2) {
            stage(xp_, wp_, xq, wq, kt, T1{});
            stage(xq, wq, xp_, wp_, kt + 1, T1{});
        }
        // ... then the last (up to four) stages in the general form
        for (; kt + 1 < nk; kt += 2) {
            stage(xp_, wp_, xq, wq, kt, T0{});
            stage(xq, wq, xp_, wp_, kt + 1, T0{});
        }
        if (kt < nk) stage(xp_, wp_, xq, wq, kt, T0{});
    }

    // ---- epilogue: lane owns token m (column r16 of each 16x16 block), registers walk 4 consecutive features --------------
    const int mb = m0 + wm * 16 * MI, nb = n0 + wn * 16 * NI;
    if (a.splitk > 1) {
        float *pb = a.part + (size_t)slice * a.M * a.N;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = mb + 16 * i + r16;
            if (m >= a.M) continue;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int nn = nb + 16 * j + 4 * kg;
                if (nn < a.N) *(f32x4 *)(pb + (size_t)m * a.N + nn) = acc[i][j];
            }
        }
        return;
    }
    if (a.stage_epi) {
        // Through LDS: the accumulator layout gives a lane one token and 4 features, i.e. 8-byte accesses in 32-byte runs over 16
        // rows per wave instruction, for the store AND for the residual read.  The tile is parked as fp32 [144][164] in the (now
        // idle) ring and walked back in whole rows: 16 bytes of fp16 per lane, 320-byte row segments, one rounding to fp16.
        constexpr int RS = G144_BN + 4;
        float *tile = (float *)smem;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        ctx_barrier();                               // every wave is done with the ring
        constexpr int NT = 64 * NW, CPR = G144_BN / 8;              // chunks of 8 features per row
        // 144 rows at a time (the 288-row tile takes two passes through the same patch)
#pragma unroll
        for (int half = 0; half < BMB / 9; ++half) {
            if (half) __syncthreads();                              // the previous pass has been read out
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int blk = wm * MI + i;                        // 16-row block of the tile
                if (blk / 9 != half) continue;
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    *(f32x4 *)(tile + ((blk - 9 * half) * 16 + r16) * RS + wn * 16 * NI + 16 * j + 4 * kg) = acc[i][j];
            }
            __syncthreads();
            for (int ch = tid; ch < 144 * CPR; ch += NT) {
                const int row = ch / CPR, c8 = (ch - row * CPR) * 8;
                const int m = m0 + 144 * half + row, n = n0 + c8;
                if (m >= a.M || n >= a.N) continue;
                const f32x4 v0 = *(const f32x4 *)(tile + row * RS + c8), v1 = *(const f32x4 *)(tile + row * RS + c8 + 4);
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                if (a.bias) {
                    const f16x8 bb = *(const f16x8 *)(a.bias + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)bb[e];
                }
                if (a.rowbias) {
                    const f16x8 bb = *(const f16x8 *)(a.rowbias + (size_t)(m / a.rows_per_batch) * a.ldrb + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)bb[e];
                }
                if (a.residual) {
                    const f16x8 bb = *(const f16x8 *)(a.residual + (size_t)m * a.ldr + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)bb[e];
                }
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (f16)v[e];
                *(f16x8 *)(a.out + (size_t)m * a.ldc + n) = o;
            }
        }
        return;
    }
    f16x4 bs[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int nn = nb + 16 * j + 4 * kg;
        bs[j] = (a.bias && nn < a.N) ? *(const f16x4 *)(a.bias + nn) : (f16x4){0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = mb + 16 * i + r16;
        if (m >= a.M) continue;
        const int bidx = a.rowbias ? m / a.rows_per_batch : 0;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int nn = nb + 16 * j + 4 * kg;
            if (nn >= a.N) continue;
            f32x4 v = acc[i][j];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)bs[j][e];
            if (a.rowbias) {
                f16x4 b = *(const f16x4 *)(a.rowbias + (size_t)bidx * a.ldrb + nn);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)b[e];
            }
            if (a.residual) {
                f16x4 b = *(const f16x4 *)(a.residual + (size_t)m * a.ldr + nn);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)b[e];
            }
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (f16)v[e];
            *(f16x4 *)(a.out + (size_t)m * a.ldc + nn) = o;
        }
    }
}

// Launch when the problem fits (returns 1): K a multiple of 64 (conv: Cin too), plain epilogue, fp16 in and out.
// form 0: 6 waves (3 x 2, wave tile 48 x 80); 1: 15 waves (3 x 5, wave tile 48 x 32), lockstep; 2: 15 waves, ring of 4, software-pipelined (MFMAs first, DMA and reads between them); 3: 15 waves, one barrier per two stages.
int ctx_gemm144_try(GemmArgs &a, bool conv, int form, hipStream_t s)
{
    if (a.K % 64 != 0 || (conv && a.Cin % 64 != 0) || a.N % 4 != 0 || a.epi != 0 || a.res32 || a.out32 || a.zins) return 0;
    if (a.ldc % 4 != 0 || (a.residual && a.ldr % 4 != 0) || (a.rowbias && a.ldrb % 4 != 0)) return 0;
    const int bm = form == 4 ? 288 : G144_BM;
    a.ntm = cdiv(a.M, bm);
    a.ntn = cdiv(a.N, G144_BN);
    int S = (a.splitk > 1 && a.part) ? a.splitk : 1;
    if (S > a.K / 64) S = a.K / 64;
    a.splitk = S;
    const double wbytes = (double)a.N * a.K, xbytes = (double)a.M * (conv ? a.Cin : a.K);
    a.mfast = wbytes > xbytes ? 1 : 0;
    static int stg = -1;
    if (stg < 0) { const char *e = getenv("CTX_G144_STAGE"); stg = e ? atoi(e) : 1; }
    a.stage_epi = stg && form != 0 && a.N % 8 == 0 && a.ldc % 8 == 0 && (!a.residual || a.ldr % 8 == 0) && (!a.rowbias || a.ldrb % 8 == 0);
    static bool attr[20] = {};
    auto go = [&](auto kern, int which, int threads, int ns) {
        const size_t lds = (size_t)ns * (bm + G144_BN) * 64 * sizeof(f16);
        if (!attr[which]) {
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr[which] = true;
        }
        if (ctx_prof_on()) {
            hipEvent_t e0, e1;
            ctx_prof_events(0, &e0, &e1);
            hipExtLaunchKernelGGL(kern, dim3(a.ntm * a.ntn * S), dim3(threads), lds, s, e0, e1, 0, a);
        } else
            hipLaunchKernelGGL(kern, dim3(a.ntm * a.ntn * S), dim3(threads), lds, s, a);
    };
    const bool ups = conv && a.ups;
    switch (form) {
    case 4:                                                          // 288 x 160, 15 waves, lockstep, ring of 2
        if (!conv) go(k_gemm144<3, 5, false, 2, 0, false, 18>, 12, 960, 2);
        else if (ups) go(k_gemm144<3, 5, true, 2, 0, true, 18>, 13, 960, 2); else go(k_gemm144<3, 5, true, 2, 0, false, 18>, 14, 960, 2);
        break;
    case 1:
        if (!conv) go(k_gemm144<3, 5, false, 3, 0>, 2, 960, 3);
        else if (ups) go(k_gemm144<3, 5, true, 3, 0, true>, 8, 960, 3); else go(k_gemm144<3, 5, true, 3, 0>, 3, 960, 3);
        break;
    case 2:
        if (!conv) go(k_gemm144<3, 5, false, 4, 1>, 4, 960, 4);
        else if (ups) go(k_gemm144<3, 5, true, 4, 1, true>, 9, 960, 4); else go(k_gemm144<3, 5, true, 4, 1>, 5, 960, 4);
        break;
    case 3:
        if (!conv) go(k_gemm144<3, 5, false, 4, 2>, 6, 960, 4);
        else if (ups) go(k_gemm144<3, 5, true, 4, 2, true>, 10, 960, 4); else go(k_gemm144<3, 5, true, 4, 2>, 7, 960, 4);
        break;
    default:
        if (!conv) go(k_gemm144<3, 2, false, 3, 0>, 0, 384, 3);
        else if (ups) go(k_gemm144<3, 2, true, 3, 0, true>, 11, 384, 3); else go(k_gemm144<3, 2, true, 3, 0>, 1, 384, 3);
        break;
    }
    return 1;
}
