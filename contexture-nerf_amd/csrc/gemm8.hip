// 256x256x64 fp16 MFMA GEMM / implicit-GEMM 3x3 convolution for the large UNet layers (gfx950): 8 waves in two
// staggered groups, half-tile LDS-DMA staging that stays in flight across barriers.
//
//   out[M,N] = X[M,K] . Wt[N,K]^T  (+bias)(+rowbias)(+residual) | GEGLU | fp32 split-K partials
//
// Replaces the cuBLAS / cuDNN calls under diffusers' UNet2DConditionModel (reference call site
// src/stable_diffusion_depth.py:422-423) for problems with >= ~150 tiles of 256x256; the smaller-tile kernel in gemm.hip
// takes the rest.  Why this shape: a CU can pull ~20-30 B/clk from L2 into LDS whatever the kernel does, so MFMA
// utilisation is set by flop per staged byte = 2*BM*BN/((BM+BN)*2); 256x256 doubles it over 128x128.
//
// Structure (one workgroup = one output tile, 512 threads):
//  * waves (wr, wc) = (wave>>2, wave&3): wave tile 128 tokens x 64 features = 8 x 4 accumulators of
//    v_mfma_f32_16x16x32_f16 (weights = A operand, activations = B operand: a lane owns one token and 4 consecutive
//    features per accumulator -> 8-byte packed stores, register-local bias / GEGLU).
//  * a K-tile (64 deep) is four half-tile slots of 16 KiB: X0/X1 = token halves (the first / second 64 tokens of BOTH
//    wave rows), W0/W1 = feature halves (the first / second 32 features of all four wave columns); two K-tiles of
//    slots = 128 KiB of LDS.  A slot is filled by 16 global_load_lds_dwordx4 pieces (8 rows x 128 B, two per wave);
//    the image is lane-linear and the 16-byte chunk index is XORed with (row>>1)&7 on the SOURCE address and on the
//    fragment read (conflict-free ds_read_b128 for the 16x16x32 operand map).
//  * a K-tile is four phases, one accumulator quadrant each: (X0,W0) (X0,W1) (X1,W1) (X1,W0).  A phase is
//        R: ds_read the fragments it needs (12 / 4 / 8 / 0), issue ONE half-tile refill, counted s_waitcnt vmcnt
//        s_barrier
//        M: 16 MFMAs
//        s_barrier
//    and the wr = 1 waves run one barrier behind the wr = 0 waves, so on every SIMD one wave is in M while its partner
//    is in R.  Refill order X0 W0 W1 X1, each slot refilled two phases after its last read (WAR) and waited for in the
//    R section of the phase BEFORE the one that reads it (RAW: own pieces by vmcnt, the other waves' by the barrier);
//    four half-tiles (64 KiB) are always in flight: vmcnt(8), never 0 until the tail.
#include "common.h"
#include "kernels.h"
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <type_traits>

typedef const __attribute__((address_space(1))) void *g8_gptr_t;
typedef __attribute__((address_space(3))) void *g8_lptr_t;
__device__ __attribute__((aligned(128))) f16 g8_zero[64];

__device__ __forceinline__ int g8_xcd_remap(int bid, int nwg)
{
    int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

__device__ __forceinline__ float g8_erf(float x)
{
    float ax = __builtin_fabsf(x);
    float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    float p = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    float e = 1.0f - p * __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    return __builtin_copysignf(e, x);
}
__device__ __forceinline__ float g8_gelu(float x) { return 0.5f * x * (1.0f + g8_erf(x * 0.70710678118654752f)); }

#define G8_WAIT(n)                                                              \
    do {                                                                        \
        if ((n) >= 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          \
        else if ((n) == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     \
        else if ((n) == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     \
        else if ((n) == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");     \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   \
    } while (0)

template <bool CONV>
__global__ __launch_bounds__(512, 2) void k_gemm8(GemmArgs a)
{
    extern __shared__ __attribute__((aligned(16))) f16 smem[];     // [2 K-tiles][X0 X1 W0 W1][128 rows][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: SGPR, scalar branches
    const int wr = wave >> 2, wc = wave & 3;
    const int r16 = lane & 15, kg = lane >> 4;

    const int ntiles = a.ntm * a.ntn;
    const int lin = g8_xcd_remap(blockIdx.x, ntiles * a.splitk);
    const int slice = lin / ntiles, bid = lin - slice * ntiles;
    const int tile_n = a.mfast ? bid / a.ntm : bid % a.ntn, tile_m = a.mfast ? bid % a.ntm : bid / a.ntn;
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    const int nkt_all = a.K / 64;
    const int kbeg = (int)((long)nkt_all * slice / a.splitk);
    const int nkt = (int)((long)nkt_all * (slice + 1) / a.splitk) - kbeg;
    const int nht = 4 * nkt;                                       // half-tiles to stage

    // ---- staging state: per half-tile kind (issue order X0 W0 W1 X1) two pieces per wave ------------------------------
    const int prow = lane >> 3, pc = lane & 7;
    const f16 *gp[4][2];
    int gst[4][2];
    // conv: per X piece the output pixel it gathers for
    int xb[2][2], xoy[2][2], xox[2][2];
    bool xok[2][2];
#pragma unroll
    for (int kind = 0; kind < 4; ++kind)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int s = 8 * (wave + 8 * i) + prow;                // slot row 0..127
            const int lc = (pc ^ ((s >> 1) & 7)) * 8;               // logical 16-byte chunk this lane fetches (in f16)
            if (kind == 0 || kind == 3) {
                const int h = kind == 0 ? 0 : 1;
                const int m = m0 + (s >> 6) * 128 + h * 64 + (s & 63);
                const bool ok = m < a.M;
                if (CONV) {
                    const int hw = a.Ho * a.Wo;
                    const int mm = ok ? m : 0;
                    const int b = mm / hw, p = mm - b * hw;
                    const int oy = p / a.Wo, ox = p - oy * a.Wo;
                    xb[h][i] = b * a.H * a.W * a.Cin + lc;
                    xoy[h][i] = oy * a.stride; xox[h][i] = ox * a.stride;
                    xok[h][i] = ok;
                    gp[kind][i] = g8_zero; gst[kind][i] = 0;
                } else {
                    gp[kind][i] = ok ? a.X + (size_t)m * a.K + (size_t)kbeg * 64 + lc : g8_zero + lc;
                    gst[kind][i] = ok ? 64 : 0;
                }
            } else {
                const int h = kind == 1 ? 0 : 1;
                const int n = n0 + (s >> 5) * 64 + h * 32 + (s & 31);
                const bool ok = n < a.N;
                gp[kind][i] = ok ? a.Wt + (size_t)n * a.K + (size_t)kbeg * 64 + lc : g8_zero + lc;
                gst[kind][i] = ok ? 64 : 0;
            }
        }

    int issued = 0;                                                 // half-tiles issued so far
    // issue half-tile `issued` (kind = issued & 3, K-tile = issued >> 2) if it exists
    auto issue = [&](const int kind, const bool steady = false) {
        if (!steady && issued >= nht) return;
        const int t = issued >> 2;
        const int slot = kind == 0 ? 0 : (kind == 3 ? 1 : (kind == 1 ? 2 : 3));      // LDS order X0 X1 W0 W1
        f16 *dst = smem + (t & 1) * 32768 + slot * 8192;
        if (CONV && (kind == 0 || kind == 3)) {
            const int h = kind == 0 ? 0 : 1;
            const int k0 = (kbeg + t) * 64;
            const int tap = k0 / a.Cin, c0 = k0 - tap * a.Cin;
            const int dy = tap / 3 - 1 + a.poff, dx = tap - (tap / 3) * 3 - 1 + a.poff;
            const int Hv = a.H << a.ups, Wv = a.W << a.ups;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int iy = xoy[h][i] + dy, ix = xox[h][i] + dx;
                const bool ok = xok[h][i] && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
                const f16 *src = ok ? a.X + xb[h][i] + (((iy >> a.ups) * a.W + (ix >> a.ups)) * a.Cin) + c0 : g8_zero;
                __builtin_amdgcn_global_load_lds((g8_gptr_t)src, (g8_lptr_t)(dst + (wave + 8 * i) * 512), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                __builtin_amdgcn_global_load_lds((g8_gptr_t)gp[kind][i], (g8_lptr_t)(dst + (wave + 8 * i) * 512), 16, 0, 0);
                gp[kind][i] += gst[kind][i];
            }
        }
        ++issued;
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (f16 units inside a slot): row s, chunk (4*ks + kg) ^ ((s>>1)&7); 16-row tiles keep the swizzle
    const int swz = (r16 >> 1) & 7;
    const int ck0 = ((kg) ^ swz) * 8, ck1 = ((4 + kg) ^ swz) * 8;
    const int xrow = (wr * 64 + r16) * 64, wrow = (wc * 32 + r16) * 64;

    // ---- prologue: X0 W0 W1 X1 of tile 0, X0 W0 of tile 1 ---------------------------------------------------------------
    issue(0); issue(1); issue(2); issue(3); issue(0); issue(1);
    { const int n = issued - 1 - 1; G8_WAIT(n); }                  // tile 0's X0, W0 (half-tiles 0, 1)
    ctx_barrier();
    if (wr == 1) ctx_barrier();                     // stagger: this group runs one barrier behind

    f16x8 xf[4][2], wf0[2][2], wf1[2][2];
    // `steady` (std::integral_constant): K-tiles during which every half-tile issue still exists and four half-tiles are in flight at
    // every wait: the counted waits are one immediate and the issues unconditional (the general form costs a dozen s_cbranch per tile)
    auto ktile = [&](const int t, auto steady) {
        constexpr bool ST = decltype(steady)::value;
        const f16 *sb = smem + (t & 1) * 32768;
        const f16 *X0 = sb + xrow, *X1 = sb + 8192 + xrow, *W0 = sb + 16384 + wrow, *W1 = sb + 24576 + wrow;
        // ---- phase 0: (X0, W0) ------------------------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < 2; ++j) { wf0[j][0] = *(const f16x8 *)(W0 + j * 1024 + ck0); wf0[j][1] = *(const f16x8 *)(W0 + j * 1024 + ck1); }
#pragma unroll
        for (int j = 0; j < 4; ++j) { xf[j][0] = *(const f16x8 *)(X0 + j * 1024 + ck0); xf[j][1] = *(const f16x8 *)(X0 + j * 1024 + ck1); }
        issue(2, ST);                                               // W1(t+1)
        { const int n = issued - 1 - (4 * t + 2); if (ST) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else G8_WAIT(n); }     // W1(t) for phase 1
        __builtin_amdgcn_sched_barrier(0);
        ctx_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0[j][ks], xf[i][ks], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        ctx_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase 1: (X0, W1) ------------------------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < 2; ++j) { wf1[j][0] = *(const f16x8 *)(W1 + j * 1024 + ck0); wf1[j][1] = *(const f16x8 *)(W1 + j * 1024 + ck1); }
        issue(3, ST);                                               // X1(t+1)
        { const int n = issued - 1 - (4 * t + 3); if (ST) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else G8_WAIT(n); }     // X1(t) for phase 2
        __builtin_amdgcn_sched_barrier(0);
        ctx_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1[j][ks], xf[i][ks], acc[i][2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        ctx_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase 2: (X1, W1) ------------------------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < 4; ++j) { xf[j][0] = *(const f16x8 *)(X1 + j * 1024 + ck0); xf[j][1] = *(const f16x8 *)(X1 + j * 1024 + ck1); }
        issue(0, ST);                                               // X0(t+2)
        __builtin_amdgcn_sched_barrier(0);
        ctx_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1[j][ks], xf[i][ks], acc[4 + i][2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        ctx_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase 3: (X1, W0) ------------------------------------------------------------------------------------
        issue(1, ST);                                               // W0(t+2)
        { const int n = issued - 1 - (4 * t + 5); if (ST) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else G8_WAIT(n); }     // X0, W0 of tile t+1 for its phase 0
        __builtin_amdgcn_sched_barrier(0);
        ctx_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0[j][ks], xf[i][ks], acc[4 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        ctx_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    {
        using T1 = std::integral_constant<bool, true>;
        using T0 = std::integral_constant<bool, false>;
        int t = 0;
        for (; t + 2 < nkt; ++t) ktile(t, T1{});
        for (; t < nkt; ++t) ktile(t, T0{});
    }
    if (wr == 0) ctx_barrier();                     // pairs with the stagger barrier of the other group

    // ---- epilogue: lane owns token m (column r16 of each 16x16 block), registers walk 4 consecutive features --------------
    const int mb = m0 + wr * 128, nb = n0 + wc * 64;
    if (a.splitk > 1) {
        float *pb = a.part + (size_t)slice * a.M * a.N;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = mb + 16 * i + r16;
            if (m >= a.M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int nn = nb + 16 * j + 4 * kg;
                if (nn < a.N) *(f32x4 *)(pb + (size_t)m * a.N + nn) = acc[i][j];
            }
        }
        return;
    }
    if (a.epi == 1) {
        // GEGLU: packed weight rows hold [32 value | 32 gate] features per 64: blocks j (value) and j + 2 (gate)
        const int fbase = nb / 2;
        f16x4 bv[2], bg[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nn = nb + 16 * j + 4 * kg;
            bv[j] = (a.bias && nn < a.N) ? *(const f16x4 *)(a.bias + nn) : (f16x4){0, 0, 0, 0};
            bg[j] = (a.bias && nn < a.N) ? *(const f16x4 *)(a.bias + nn + 32) : (f16x4){0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = mb + 16 * i + r16;
            if (m >= a.M) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nn = nb + 16 * j + 4 * kg;
                if (nn >= a.N) continue;
                f16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float xv = acc[i][j][e] + (float)bv[j][e], gv = acc[i][2 + j][e] + (float)bg[j][e];
                    o[e] = (f16)(xv * g8_gelu(gv));
                }
                *(f16x4 *)(a.out + (size_t)m * a.ldc + fbase + 16 * j + 4 * kg) = o;
            }
        }
        return;
    }
    f16x4 bs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nn = nb + 16 * j + 4 * kg;
        bs[j] = (a.bias && nn < a.N) ? *(const f16x4 *)(a.bias + nn) : (f16x4){0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = mb + 16 * i + r16;
        if (m >= a.M) continue;
        const int bidx = a.rowbias ? m / a.rows_per_batch : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
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

// Launch if the problem suits the 256x256 tile (returns 1), else 0 and the caller falls back to gemm.hip's tiles.
int ctx_gemm8_try(GemmArgs &a, bool conv, bool force, hipStream_t s)
{
    // CTX_GEMM8: 0 off, 1 auto (default), 2 force whenever the kernel is applicable (tests); read per call on purpose
    const char *e = getenv("CTX_GEMM8");
    int en = e ? atoi(e) : 1;
    if (force) en = 2;
    const char *tt = getenv("CTX_GEMM8_MIN_TILES");
    const int min_tiles = tt ? atoi(tt) : 180;
    if (!en) return 0;
    if (a.K % 64 != 0 || (conv && a.Cin % 64 != 0) || a.N % 8 != 0) return 0;
    if (a.epi == 1 && a.N % 64 != 0) return 0;
    const int ntm = cdiv(a.M, 256), ntn = cdiv(a.N, 256);
    const int S = (a.splitk > 1 && a.part) ? a.splitk : 1;
    if (en != 2) {
        // enough tiles to fill the chip, little masked waste, and a K loop long enough to amortise prologue + epilogue
        const double useful = (double)a.M * a.N / ((double)ntm * ntn * 65536.0);
        if (ntm * ntn * S < min_tiles || useful < 0.8 || a.K / 64 / S < 4) return 0;
    }
    a.ntm = ntm; a.ntn = ntn; a.splitk = S;
    const double wbytes = (double)a.N * a.K, xbytes = (double)a.M * (conv ? a.Cin : a.K);
    a.mfast = wbytes > xbytes ? 1 : 0;
    const size_t lds = 131072;
    static bool attr[2] = {false, false};
    auto go = [&](auto kern, int which) {
        if (!attr[which]) {
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr[which] = true;
        }
        if (ctx_prof_on()) {
            hipEvent_t e0, e1;
            ctx_prof_events(0, &e0, &e1);
            hipExtLaunchKernelGGL(kern, dim3(ntm * ntn * S), dim3(512), lds, s, e0, e1, 0, a);
        } else
            hipLaunchKernelGGL(kern, dim3(ntm * ntn * S), dim3(512), lds, s, a);
    };
    if (conv) go(k_gemm8<true>, 1); else go(k_gemm8<false>, 0);
    return 1;
}
