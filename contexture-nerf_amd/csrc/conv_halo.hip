// Halo-staged 3x3 convolution (stride 1, no upsample) for the large feature maps of the UNet (gfx950).
//
//   out[b, y, x, n] = sum_{tap, c} X[b, y + dy, x + dx, c] * Wt[n, tap, c]  (+bias)(+rowbias)(+residual) | fp32 split-K partials
//
// Replaces the cuDNN conv2d calls of diffusers' ResnetBlock2D (reference call site src/stable_diffusion_depth.py:422-423)
// where the implicit-GEMM kernel of gemm.hip is bound by the bytes it stages: that kernel re-stages every output pixel's
// input vector nine times (once per tap).  Here a workgroup owns a 16x16 patch of output pixels and, per 64-channel
// chunk, stages the 18x18 input patch (patch + halo) ONCE; the nine taps read it at shifted rows.  Staged bytes per
// (256 pixels x 128 features x 64 channels x 9 taps): 41 KiB + 144 KiB instead of 288 KiB + 144 KiB.
//
// Structure: 8 waves = 4 (pixel rows) x 2 (features); wave tile 64 pixels x 32*NI features as 2 x NI accumulators of
// v_mfma_f32_32x32x16_f16 (weights = A operand, pixels = B operand).  A 32-pixel MFMA block is patch rows {y, y + 8}
// (16 pixels each): with 18-pixel halo rows the second half then sits 8*18 = 144 = 0 (mod 16) rows after the first, which
// keeps the 16-lane groups of ds_read_b128 on 16 different (row, chunk) slots for EVERY tap shift (chunk index XORed with
// (row>>1)&7 on the DMA source address and on the read, as in gemm.hip).
// LDS: halo patch double-buffered (2 x 41 KiB, 8-row DMA pieces), weight ring of 4 tap-stages (BN x 64 channels each),
// 1 KiB of scratch per wave for dummy pieces.  One stage = one (chunk, tap), run as two barrier-separated sections by two
// wave groups one barrier apart (as gemm8.hip), counted vmcnt; every wave
// issues the same number of DMA pieces per stage (one slot of the NEXT chunk's halo + its share of the weight stage two
// ahead; slots past the end go to the scratch from a zero page) so one immediate vmcnt count is right for all waves.
#include "common.h"
#include "kernels.h"
#include <hip/hip_ext.h>
#include <stdlib.h>

typedef const __attribute__((address_space(1))) void *ch_gptr_t;
typedef __attribute__((address_space(3))) void *ch_lptr_t;
__device__ __attribute__((aligned(128))) f16 ch_zero[64];

#define CH_HP 328                 // halo rows per buffer: 18 x 18 = 324 pixels, padded to 41 pieces of 8
#define CH_XPIECES 41

__device__ __forceinline__ int ch_xcd_remap(int bid, int nwg)
{
    int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

template <int NI>
__global__ __launch_bounds__(512) void k_conv_halo(GemmArgs a)
{
    constexpr int BN = 64 * NI;
    constexpr int WP = NI;                                       // weight pieces per wave per stage (BN / 8 rows / 8 waves)
    constexpr int G = 1 + WP;                                    // DMA pieces per wave per stage
    constexpr int HALO = CH_HP * 64;                             // f16 per halo buffer
    constexpr int WST = BN * 64;                                 // f16 per weight stage
    extern __shared__ __attribute__((aligned(16))) f16 smem[];  // [2 halo][4 weight stages][8 x 512 scratch]
    f16 *halo = smem;
    f16 *wring = smem + 2 * HALO;
    f16 *scratch = wring + 4 * WST;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: SGPR, scalar branches
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    const int pw = a.W >> 4, ph = a.H >> 4;                      // patches per row / column
    const int ntiles = a.ntm * a.ntn;
    const int lin = ch_xcd_remap(blockIdx.x, ntiles * a.splitk);
    const int slice = lin / ntiles, bid = lin - slice * ntiles;
    const int tile_n = bid % a.ntn, tile_m = bid / a.ntn;
    const int b = tile_m / (pw * ph), pr = tile_m - b * (pw * ph);
    const int py0 = (pr / pw) * 16, px0 = (pr % pw) * 16;
    const int n0 = tile_n * BN;

    const int nch_all = a.Cin / 64;
    const int cbeg = (int)((long)nch_all * slice / a.splitk);
    const int nch = (int)((long)nch_all * (slice + 1) / a.splitk) - cbeg;
    const int nst = 9 * nch;                                     // stages of this workgroup

    // ---- DMA state ----------------------------------------------------------------------------------------------
    const int prow = lane >> 3, pc = lane & 7;
    const f16 *xbase = a.X + (size_t)b * a.H * a.W * a.Cin;
    // weight pieces: rows n0 + 8 (wave + 8 i) + prow
    const f16 *wp[WP];
    bool wok[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int row = 8 * (wave + 8 * i) + prow;
        const int n = n0 + row;
        wok[i] = n < a.N;
        wp[i] = a.Wt + (size_t)(wok[i] ? n : 0) * a.K + ((pc ^ ((row >> 1) & 7)) * 8);
    }
    // issue halo slot `xs` (0..71, only < 41 real) of chunk `c` into halo buffer `hb`
    auto issue_x = [&](int xs, int c, int hb) {
        const int hp = 8 * xs + prow;                            // halo pixel of this lane
        const int hy = hp / 18, hx = hp - hy * 18;
        const int y = py0 + hy - 1, x = px0 + hx - 1;
        const bool ok = xs < CH_XPIECES && hp < 324 && c < cbeg + nch && y >= 0 && y < a.H && x >= 0 && x < a.W;
        const f16 *src = ok ? xbase + ((size_t)y * a.W + x) * a.Cin + c * 64 + ((pc ^ ((hp >> 1) & 7)) * 8) : ch_zero;
        f16 *dst = (xs < CH_XPIECES && c < cbeg + nch) ? halo + hb * HALO + xs * 512 : scratch + wave * 512;
        __builtin_amdgcn_global_load_lds((ch_gptr_t)src, (ch_lptr_t)dst, 16, 0, 0);
    };
    // issue the weight pieces of stage g (chunk cbeg + g / 9, tap g % 9) into ring slot g % 4
    auto issue_w = [&](int g) {
        const bool live = g < nst;
        const int c = cbeg + g / 9, t = g - (g / 9) * 9;
        const size_t off = (size_t)t * a.Cin + (size_t)c * 64;
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            const f16 *src = (live && wok[i]) ? wp[i] + off : ch_zero;
            f16 *dst = live ? wring + (g & 3) * WST + (wave + 8 * i) * 512 : scratch + wave * 512;
            __builtin_amdgcn_global_load_lds((ch_gptr_t)src, (ch_lptr_t)dst, 16, 0, 0);
        }
    };

    f32x16 acc[2][NI];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    // halo row of this lane's pixel for the centre tap, per 32-pixel block mi: patch rows {2 wm + mi, 2 wm + mi + 8}
    int rc[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) rc[mi] = (2 * wm + mi + 8 * (r >> 4) + 1) * 18 + (r & 15) + 1;
    const int wswz = (r >> 1) & 7;
    const int wrow = (wn * 32 * NI + r) * 64;

    // ---- prologue: the whole halo of the first chunk (6 slots per wave), weight stages 0 and 1 ----------------------
#pragma unroll
    for (int i = 0; i < 6; ++i) issue_x(wave + 8 * i, cbeg, 0);
    issue_w(0);
    issue_w(1);
    issue_w(2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WP) : "memory");    // halo of chunk 0 and weight stage 0 (stages 1, 2 may fly)
    ctx_barrier();
    const int grp = wave >> 2;                                   // SIMD partners are waves w and w + 4
    if (grp == 1) ctx_barrier();                  // stagger: this group runs one barrier behind

    // A stage is two barrier-separated sections, R (fragment reads, DMA issue, counted wait) and M (16 MFMAs); the two wave
    // groups alternate, so on every SIMD one wave is in M while its partner is in R.  Slots are refilled two stages after
    // their last read (weight ring of 4, weights issued three stages ahead from the M section) and waited for in the R
    // section of the stage before their first read.
    for (int g = 0; g < nst; ++g) {
        const int ci = g / 9, t = g - ci * 9;                    // chunk index inside the slice, tap
        const f16 *hb = halo + (ci & 1) * HALO;
        const f16 *wb = wring + (g & 3) * WST + wrow;
        const int dy = t / 3 - 1, dx = t - (t / 3) * 3 - 1;
        const int toff = dy * 18 + dx;
        int xr[2], xs[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) { const int R = rc[mi] + toff; xr[mi] = R * 64; xs[mi] = (R >> 1) & 7; }
        f16x8 xf[4][2], wf[4][NI];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) xf[ks][mi] = *(const f16x8 *)(hb + xr[mi] + (((2 * ks + h) ^ xs[mi]) * 8));
#pragma unroll
            for (int j = 0; j < NI; ++j) wf[ks][j] = *(const f16x8 *)(wb + j * 32 * 64 + (((2 * ks + h) ^ wswz) * 8));
        }
        // stage g + 1 must be readable after this section's barrier: everything older than the newest group has landed
        if (g == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WP) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        ctx_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks][j], xf[ks][mi], acc[mi][j], 0, 0, 0);
            // this stage's DMA group rides in the MFMA section (an LDS-DMA piece costs ~60 issue cycles among MFMAs, 100-185
            // among ds_reads): one halo slot of the next chunk, the weights of stage g + 3 (ring slot of stage g - 1)
            if (ks == 0) issue_x(t * 8 + wave, cbeg + ci + 1, (ci + 1) & 1);
            if (ks == 1) issue_w(g + 3);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        ctx_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 0) ctx_barrier();                  // pairs with the stagger barrier of the other group
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // drain the dummy pieces before the workgroup retires

    // ---- epilogue: lane owns one pixel, registers walk 4 consecutive features -------------------------------------------
    const int nw = n0 + wn * 32 * NI;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int y = py0 + 2 * wm + mi + 8 * (r >> 4), x = px0 + (r & 15);
        const size_t m = ((size_t)b * a.H + y) * a.W + x;
        if (a.splitk > 1) {
            float *pb = a.part + ((size_t)slice * a.M + m) * a.N;
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int nn = nw + j * 32 + 8 * g4 + 4 * h;
                    if (nn < a.N) *(f32x4 *)(pb + nn) = (f32x4){acc[mi][j][4 * g4], acc[mi][j][4 * g4 + 1], acc[mi][j][4 * g4 + 2], acc[mi][j][4 * g4 + 3]};
                }
            continue;
        }
        const int bidx = a.rowbias ? (int)(m / a.rows_per_batch) : 0;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int nn = nw + j * 32 + 8 * g4 + 4 * h;
                if (nn >= a.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[mi][j][4 * g4 + e];
                if (a.bias) {
                    f16x4 bb = *(const f16x4 *)(a.bias + nn);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)bb[e];
                }
                if (a.rowbias) {
                    f16x4 bb = *(const f16x4 *)(a.rowbias + (size_t)bidx * a.ldrb + nn);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)bb[e];
                }
                if (a.residual) {
                    f16x4 bb = *(const f16x4 *)(a.residual + m * a.ldr + nn);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)bb[e];
                }
                f16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (f16)v[e];
                *(f16x4 *)(a.out + m * a.ldc + nn) = o;
            }
    }
}

// Launch when the problem fits (returns 1): stride 1, no upsample, H and W multiples of 16, Cin multiple of 64, epi 0.
int ctx_conv_halo_try(GemmArgs &a, int ni, hipStream_t s)
{
    if (a.stride != 1 || a.ups != 0 || (a.H & 15) || (a.W & 15) || a.Cin % 64 != 0 || a.epi != 0 || a.N % 8 != 0) return 0;
    const int B = a.M / (a.H * a.W);
    const int BN = 64 * ni;
    a.ntm = B * (a.H >> 4) * (a.W >> 4);
    a.ntn = cdiv(a.N, BN);
    const int S = (a.splitk > 1 && a.part) ? a.splitk : 1;
    if (S > a.Cin / 64) return 0;
    a.splitk = S;
    const size_t lds = (size_t)(2 * CH_HP * 64 + 4 * BN * 64 + 8 * 512) * sizeof(f16);
    static bool attr[2] = {false, false};
    auto go = [&](auto kern, int which) {
        if (!attr[which]) {
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr[which] = true;
        }
        if (ctx_prof_on()) {
            hipEvent_t e0, e1;
            ctx_prof_events(0, &e0, &e1);
            hipExtLaunchKernelGGL(kern, dim3(a.ntm * a.ntn * S), dim3(512), lds, s, e0, e1, 0, a);
        } else
            hipLaunchKernelGGL(kern, dim3(a.ntm * a.ntn * S), dim3(512), lds, s, a);
    };
    if (ni == 2) go(k_conv_halo<2>, 0); else go(k_conv_halo<1>, 1);
    return 1;
}
