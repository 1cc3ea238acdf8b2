// Flash-style fused attention  O = softmax(Q K^T * scale) V  for the UNet transformer blocks
// (diffusers Attention / AttnProcessor2_0 under src/stable_diffusion_depth.py:422), head_dim 64, fp16
// operands, fp32 scores / running max / running sum / output accumulator.
//
// 4 waves per workgroup, 32 query rows per wave; K/V tiles of 64 keys staged through LDS and shared
// by the 4 waves.  Scores are computed TRANSPOSED (S^T = K . Q^T: K rows are the MFMA A operand, Q
// the B operand) so each lane owns one query column: the softmax row reductions are in-register plus a
// single cross-half shuffle, and the exponentiated tile, converted to f16 in place, is already the B
// operand of the second product O^T = V^T . P^T (no LDS round trip for P).  V arrives pre-transposed
// ([B, heads, 64, Skv_pad], keys contiguous) so its fragment is two 8-byte LDS reads.
#include "common.h"
#include "kernels.h"
#include <hip/hip_ext.h>
#include <math.h>
#include <stdlib.h>

#define AT_KB 64                // keys per tile
#define AT_KROW 72              // f16 per K row in LDS (144 B: conflict-free ds_read_b128)
#define AT_VROW 68              // f16 per V^T row in LDS (136 B: conflict-free ds_read_b64)


struct AttnArgs {
    const f16 *Q, *K, *V;
    f16 *O;
    int Sq, Skv, heads;
    int q_stride, kv_stride, o_stride;
    float scale_log2e;
    float lazy;            // log2 units a row's maximum may run ahead of the subtracted one before the accumulators are rescaled (0: never)
};

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant (default): K and V^T tiles arrive by global_load_lds into a ring of AT_NS stages (1 KiB pieces of
// 8 rows x 128 B, chunk swizzle c ^ ((row>>1)&7) on the source address and on the fragment reads), prefetch distance
// AT_NS-1 tiles, one raw s_barrier + counted vmcnt per tile — the same pipeline as k_gemm_pipe.
__device__ __attribute__((aligned(16))) f16 g_attn_zero[64];
typedef const __attribute__((address_space(1))) void *agptr_t;
typedef __attribute__((address_space(3))) void *alptr_t;


// One 64-key tile of the online-softmax recurrence for this wave's 32 queries (MASK only for the last, ragged tile).
// V stays [key][d] in LDS exactly as it lies in memory (no transpose pass): the PV product's A operand (V^T: row d,
// 8 keys) comes from two transposing reads ds_read_b64_tr_b16, each a 4-key x 16-d block per 16-lane group: lane
// (q = (l&15)>>2, p = l&3) points at key row q, d columns 4p..4p+3 and receives d column l&15 of the 4 keys — exactly
// the keys 4h + (0..3) (then 8 + 4h + (0..3)) that element j of the P fragment holds.  The 16-byte chunk index of the V
// image is XORed with 4 on key rows 2, 3 (mod 4), which spreads the four rows of a block over all 64 banks.
// The row sums l = sum_k p ride on the matrix pipe: one extra MFMA per k-step with an all-ones A operand accumulates, in
// every row of `ls`, the sum of exactly the fp16-rounded P the PV product uses; the VALU is this kernel's critical resource.
typedef short at_s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
__device__ __forceinline__ f16x4 at_tr16(const f16 *p)
{
    at_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) at_s16x4 *)p);
    return __builtin_bit_cast(f16x4, v);
}
struct AtNoIssue { __device__ __forceinline__ void operator()(int) const {} };
// NISSUE > 0: `iss(i)` launches DMA piece i of a later tile right behind the i-th QK MFMA.  The four waves of a workgroup leave the
// tile's barrier together, and 16 KB of global_load_lds issued in one burst queue at the CU's one vector-memory port: timestamps
// taken inside the loop (s_memtime around each section, round 3) showed ~10 % of a wave's tile period stalled at the issue of
// its 4 pieces.
template <bool MASK, int NISSUE = 0, class Issue = AtNoIssue>
__device__ __forceinline__ void attn_tile(const f16 *__restrict__ Ks, const f16 *__restrict__ Vs, int k0, int Skv, int r, int h,
                                          int swz, int vlane, int vfq, float c, const f16x8 (&qf)[4], f32x16 (&o)[2], f32x16 &ls,
                                          float &m_run, float lazy, Issue iss = Issue())
{
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int q = 0; q < 16; ++q) s[kb][q] = 0.f;
        const f16 *kr = Ks + (kb * 32 + r) * 64;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f16x8 kf = *(const f16x8 *)(kr + ((2 * ks + h) ^ swz) * 8);
            s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[kb], 0, 0, 0);
            if (kb * 4 + ks < NISSUE) iss(kb * 4 + ks);
        }
    }
    if (MASK) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                int key = k0 + kb * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (key >= Skv) s[kb][q] = -INFINITY;
            }
    }
    // 32 scores -> three-input maxima in two chains (v_max3_f32), then the other half's lane
    float mxa = fmaxf(s[0][0], s[0][1]), mxb = fmaxf(s[1][0], s[1][1]);
#pragma unroll
    for (int q = 2; q < 16; q += 2) {
        mxa = __builtin_fmaxf(__builtin_fmaxf(mxa, s[0][q]), s[0][q + 1]);
        mxb = __builtin_fmaxf(__builtin_fmaxf(mxb, s[1][q]), s[1][q + 1]);
    }
    float mx = fmaxf(mxa, mxb);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c;
    // The subtracted maximum is any per-row constant: it follows the true running maximum only when some row of the wave has moved by
    // more than `lazy` (P then stays <= 2^lazy, exact in fp16's range), which skips most of the 33-multiply rescales of O and l
    // (lazy = 0: every change, the textbook recurrence; first tile: m_run = -inf always moves)
    const float m_cand = fmaxf(m_run, mx);
    const bool move = __any(m_cand > m_run + lazy);
    const float m_new = move ? m_cand : m_run;
    // s * c - m_new two scores per instruction (v_pk_fma_f32: the same fused operation per element), then the exponentials
    typedef float at_f2 __attribute__((ext_vector_type(2)));
    const at_f2 c2 = {c, c}, nm2 = {-m_new, -m_new};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            const at_f2 t = __builtin_elementwise_fma((at_f2){s[kb][q], s[kb][q + 1]}, c2, nm2);
            s[kb][q] = __builtin_amdgcn_exp2f(t[0]);
            s[kb][q + 1] = __builtin_amdgcn_exp2f(t[1]);
        }
    // rescale only when some row's max moved (wave-uniform branch).  The multiplies are inline asm with tied operands:
    // written as C++ the compiler multiplies out of place and then copies all 48 accumulator registers on the
    // fall-through path of every tile, which costs more than multiplying unconditionally.
    if (move) {
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);      // 0 on the first tile (m_run = -inf)
        // s_nop: alpha comes straight from v_exp_f32 (transcendental -> VALU use needs a wait state hipcc cannot see into)
        asm volatile("s_nop 1\n\tv_mul_f32 %0, %0, %1" : "+v"(ls[0]) : "v"(alpha));
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int q = 0; q < 16; ++q) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(o[d][q]) : "v"(alpha));
    }
    m_run = m_new;
    const f16x8 ones = {1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            f16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (f16)s[kb][8 * st + j];
            const f16 *vk = Vs + (32 * kb + 16 * st + 4 * h) * 64 + vlane;       // key block of this lane half
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int co = ((4 * d) ^ vfq) * 8;                              // d block, swizzled with the lane's key row
                f16x4 v0 = at_tr16(vk + co), v1 = at_tr16(vk + 8 * 64 + co);
                f16x8 vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[d], 0, 0, 0);
            }
            ls = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, pf, ls, 0, 0, 0);
        }
}

// NW waves per workgroup (32 queries each): 4 = 128 queries, 8 = 256 queries per staged K/V tile (half the L2 -> LDS bytes per
// FLOP: the staging path is this chip's scarce resource, ~28 B/clk per CU, and three 4-wave workgroups per CU ask ~20 of it)
template <int AT_NS, int NW, bool SPREAD = true>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 1) void k_attention_dma(AttnArgs a)
{
    __shared__ __attribute__((aligned(16))) f16 ring[AT_NS * 2 * 64 * 64];    // per stage: K [64][64] then V^T [64][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: SGPR, scalar branches
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, hd = blockIdx.y;
    const int q0 = blockIdx.x * (32 * NW) + wave * 32;
    const int qrow = q0 + r;
    const bool qok = qrow < a.Sq;
    constexpr int PI = 8 / NW;                       // K pieces (and V pieces) per wave per tile: the tile is 8 + 8 pieces of 1 KiB
    constexpr int G = 2 * PI;                        // DMA pieces per wave per tile

    f16x8 qf[4];
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    {
        const f16 *qp = a.Q + ((size_t)b * a.Sq + (qok ? qrow : 0)) * a.q_stride + hd * 64 + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = qok ? *(const f16x8 *)(qp + ks * 16) : zero8;
    }
    f32x16 o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int q = 0; q < 16; ++q) o[d][q] = 0.f;
    f32x16 ls;
#pragma unroll
    for (int q = 0; q < 16; ++q) ls[q] = 0.f;
    float m_run = -INFINITY;

    const int ntiles = (a.Skv + AT_KB - 1) / AT_KB;
    // issue-side state: this wave moves pieces {wave, wave+4} of the K tile and of the V^T tile
    const int lr = lane >> 3, pc = lane & 7;
    const f16 *kp[PI], *vp[PI];
    int krow[PI];
#pragma unroll
    for (int i = 0; i < PI; ++i) {
        int row = 8 * (wave + NW * i) + lr;
        int lc = (pc ^ ((row >> 1) & 7)) * 8;
        krow[i] = row;
        kp[i] = a.K + ((size_t)b * a.Skv + row) * a.kv_stride + hd * 64 + lc;
        vp[i] = a.V + ((size_t)b * a.Skv + row) * a.kv_stride + hd * 64 + ((pc ^ (((row >> 1) & 1) << 2)) * 8);
    }
    int issued = 0;
    auto issue = [&](int buf) {
        f16 *Ks = ring + buf * (2 * 64 * 64);
        f16 *Vs = Ks + 64 * 64;
        const int k0 = issued * AT_KB;
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const f16 *src = (k0 + krow[i] < a.Skv) ? kp[i] : g_attn_zero;
            __builtin_amdgcn_global_load_lds((agptr_t)src, (alptr_t)(Ks + (wave + NW * i) * 512), 16, 0, 0);
            kp[i] += (size_t)AT_KB * a.kv_stride;
        }
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const f16 *src = (k0 + krow[i] < a.Skv) ? vp[i] : g_attn_zero;
            __builtin_amdgcn_global_load_lds((agptr_t)src, (alptr_t)(Vs + (wave + NW * i) * 512), 16, 0, 0);
            vp[i] += (size_t)AT_KB * a.kv_stride;
        }
        ++issued;
    };
    // one piece of tile `issued` (idx < PI: K piece idx, else V piece idx - PI); the caller bumps `issued` after the last one
    auto issue_piece = [&](int buf, int idx) {
        f16 *Ks = ring + buf * (2 * 64 * 64);
        f16 *Vs = Ks + 64 * 64;
        const int k0 = issued * AT_KB;
        const int i = idx < PI ? idx : idx - PI;
        const f16 *&gp = idx < PI ? kp[i] : vp[i];
        const f16 *src = (k0 + krow[i] < a.Skv) ? gp : g_attn_zero;
        __builtin_amdgcn_global_load_lds((agptr_t)src, (alptr_t)((idx < PI ? Ks : Vs) + (wave + NW * i) * 512), 16, 0, 0);
        gp += (size_t)AT_KB * a.kv_stride;
    };
#pragma unroll
    for (int p = 0; p < AT_NS - 1; ++p)
        if (p < ntiles) issue(p);

    const int swz = (r >> 1) & 7;
    // V reads: lane (q = (l&15)>>2, p = l&3, a = (l>>4)&1) -> key row q, d columns 16a + 4p.. of a 32-d block
    const int vq = (lane & 15) >> 2, vfq = ((vq >> 1) & 1) << 2;
    const int vlane = vq * 64 + (((2 * ((lane >> 4) & 1) + ((lane & 3) >> 1)) ^ 0) * 8) + (lane & 1) * 4;
    const float c = a.scale_log2e;
    // one copy of the tile body in the loop (runtime ring slot): full tiles first, the ragged tile peeled off the end, so
    // the accumulators keep one fixed register block (unrolled / two-variant bodies made hipcc shuffle all 48 of them)
    const int nfull = a.Skv / AT_KB;
    int slot = 0;
    int t = 0;
    // steady tiles: a tile AT_NS-1 ahead is still to be issued, so exactly AT_NS-2 newer tiles are in flight at the wait (no
    // wave-uniform branches in the body)
    for (; t < nfull && t + AT_NS - 1 < ntiles; ++t) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AT_NS - 2) * G) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        ctx_barrier();
        const int ibuf = slot == 0 ? AT_NS - 1 : slot - 1;
        const f16 *Ks = ring + slot * (2 * 64 * 64);
        if (SPREAD) {
            attn_tile<false, G>(Ks, Ks + 64 * 64, t * AT_KB, a.Skv, r, h, swz, vlane, vfq, c, qf, o, ls, m_run, a.lazy, [&](int idx) { issue_piece(ibuf, idx); });
            ++issued;
        } else {
            issue(ibuf);
            attn_tile<false>(Ks, Ks + 64 * 64, t * AT_KB, a.Skv, r, h, swz, vlane, vfq, c, qf, o, ls, m_run, a.lazy);
        }
        slot = slot + 1 == AT_NS ? 0 : slot + 1;
    }
    for (; t < nfull; ++t) {
        const int newer = issued - 1 - t;
        if (AT_NS >= 4 && newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
        else if (newer >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // tile t-1's slot is refilled right after the barrier and the barrier does not wait for LDS reads in flight: the tile's
        // MFMAs (which wait for their K / V fragments) must all stay above it
        __builtin_amdgcn_sched_barrier(0);
        ctx_barrier();
        if (issued < ntiles) issue(slot == 0 ? AT_NS - 1 : slot - 1);          // the slot tile t-1 used
        const f16 *Ks = ring + slot * (2 * 64 * 64);
        attn_tile<false>(Ks, Ks + 64 * 64, t * AT_KB, a.Skv, r, h, swz, vlane, vfq, c, qf, o, ls, m_run, a.lazy);
        slot = slot + 1 == AT_NS ? 0 : slot + 1;
    }
    if (nfull < ntiles) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ctx_barrier();
        const f16 *Ks = ring + slot * (2 * 64 * 64);
        attn_tile<true>(Ks, Ks + 64 * 64, nfull * AT_KB, a.Skv, r, h, swz, vlane, vfq, c, qf, o, ls, m_run, a.lazy);
    }
    if (qok) {
        float inv = 1.0f / ls[0];
        f16 *op = a.O + ((size_t)b * a.Sq + qrow) * a.o_stride + hd * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f16x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (f16)(o[d][4 * g + j] * inv);
                *(f16x4 *)(op + d * 32 + 8 * g + 4 * h) = v;
            }
    }
}

int ctx_attention_core(const f16 *Q, const f16 *K, const f16 *V, int B, int Sq, int Skv, int heads, int q_stride,
                       int kv_stride, float scale, f16 *O, int o_stride, hipStream_t s)
{
    AttnArgs a;
    a.Q = Q; a.K = K; a.V = V; a.O = O; a.Sq = Sq; a.Skv = Skv; a.heads = heads;
    a.q_stride = q_stride; a.kv_stride = kv_stride; a.o_stride = o_stride;
    a.scale_log2e = scale * 1.4426950408889634f;
    static float lazy = -1.f;           // CTX_ATTN_LAZY: threshold in log2 units (default 8 = a factor 256; 0 = rescale on every change)
    if (lazy < 0.f) { const char *e = getenv("CTX_ATTN_LAZY"); lazy = e ? (float)atof(e) : 8.0f; if (!(lazy >= 0.f && lazy <= 12.f)) lazy = 8.0f; }
    a.lazy = lazy;
    static int ns = -1, nw8 = -1;
    if (ns < 0) { const char *e = getenv("CTX_ATTN_NS"); ns = e ? atoi(e) : 3; }
    // 8-wave workgroups (two per CU at 128 VGPRs) measured against three 4-wave ones (148 VGPRs): 72 vs 77 us at 2304 tokens
    // x 10 heads, 411 vs 371 us at 9216 x 5 — so only the mid-size self-attention takes them (CTX_ATTN_NW8: 0 never, 1 always)
    if (nw8 < 0) { const char *e = getenv("CTX_ATTN_NW8"); nw8 = e ? atoi(e) : -1; }
    const bool w8 = nw8 == 1 || (nw8 < 0 && Sq >= 1024 && Sq < 4096 && Skv >= 1024);
    static int spread = -1;             // CTX_ATTN_SPREAD=0: all DMA pieces of a tile right after the barrier (the A/B switch)
    if (spread < 0) { const char *e = getenv("CTX_ATTN_SPREAD"); spread = e ? atoi(e) : 1; }
    auto kern = w8 ? (ns == 2 ? k_attention_dma<2, 8> : (ns == 4 ? k_attention_dma<4, 8> : (spread ? k_attention_dma<3, 8> : k_attention_dma<3, 8, false>)))
                   : (ns == 2 ? k_attention_dma<2, 4> : (ns == 4 ? k_attention_dma<4, 4> : (spread ? k_attention_dma<3, 4> : k_attention_dma<3, 4, false>)));
    const int nthr = w8 ? 512 : 256, qpw = w8 ? 256 : 128;
    static int dbg = -1;
    if (dbg < 0) {
        const char *e = getenv("CTX_ATTN_DEBUG"); dbg = e ? atoi(e) : 0;
        if (dbg) {
            int nb = 0;
            hipFuncAttributes fa;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, nthr, 0);
            (void)hipFuncGetAttributes(&fa, (const void *)kern);
            fprintf(stderr, "[ctx] attention: %d blocks/CU by the occupancy API, %d VGPRs, %zu B static LDS\n", nb, fa.numRegs, fa.sharedSizeBytes);
        }
    }
    if (ctx_prof_on()) {
        hipEvent_t e0, e1;
        ctx_prof_events(1, &e0, &e1);
        hipExtLaunchKernelGGL(kern, dim3(cdiv(Sq, qpw), heads, B), dim3(nthr), 0, s, e0, e1, 0, a);
    } else
        hipLaunchKernelGGL(kern, dim3(cdiv(Sq, qpw), heads, B), dim3(nthr), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ctx_set_error("attention launch failed: %s", hipGetErrorString(e));
        return CTX_E_LAUNCH;
    }
    return CTX_OK;
}

extern "C" int64_t ctx_attention_ws_bytes(int32_t B, int32_t Skv, int32_t heads)
{
    (void)B; (void)Skv; (void)heads;
    return 256;                                   // no workspace is needed any more (V is consumed untransposed); kept for callers
}

extern "C" int32_t ctx_attention_f16(const void *Q, const void *K, const void *V, int32_t B, int32_t Sq, int32_t Skv,
                                     int32_t heads, int32_t q_stride, int32_t kv_stride, float scale, void *O,
                                     int32_t o_stride, void *vt_ws, ctx_stream_t stream)
{
    (void)vt_ws;
    CTX_REQUIRE(Q && K && V && O, "attention: null pointer");
    CTX_REQUIRE(B > 0 && Sq > 0 && Skv > 0 && heads > 0 && q_stride % 8 == 0 && kv_stride % 8 == 0 && o_stride % 4 == 0 &&
                    q_stride >= heads * 64 && kv_stride >= heads * 64 && o_stride >= heads * 64,
                "attention: bad strides/sizes (head_dim is fixed at 64)");
    return ctx_attention_core((const f16 *)Q, (const f16 *)K, (const f16 *)V, B, Sq, Skv, heads, q_stride, kv_stride, scale, (f16 *)O,
                              o_stride, (hipStream_t)stream);
}
