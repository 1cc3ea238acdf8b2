// Fused texture field:  uv -> Fourier embed -> NeRF2D MLP -> raw rgb (-> (tanh+1)/2 atlas).
// Replaces get_embedder + NeRF2D.forward (src/run_nerf_helpers.py:15-135) as called from
// TexturedMeshModel.get_texture_map (src/models/textured_mesh.py:266-301).
//
// The reference computes this in fp32, so the contraction runs on the exact-f32 matrix pipe
// (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain, 157 TFLOP/s peak on MI355X).
//
// One workgroup = 64 texels x W hidden units; wave w owns hidden columns [64w, 64w+64) as 2x2
// accumulator tiles of 32x32.  Activations never leave the CU: they live in LDS as
// act[64][STRIDE] floats with the (zero-padded to 48) embedding in columns [0,48) and the hidden
// vector in [48,48+W); the skip layer simply reads columns [0,48+W).  Weights are pre-packed so a
// lane's four k-steps of an 8-wide k-block are one 16-byte global load (L2-resident: 1.9 MB).
#include "common.h"
#include <math.h>
#include <stdlib.h>

#define UVM_MAX_LAYERS 16
#define UVM_EPAD 48       // padded embedding width of the 2-D field (42 -> 48)
#define UVM_EPAD3 64      // ... of the 3-D field (63 -> 64); the plan carries the one in use
#define UVM_TM 64         // texels per workgroup

// Weight fragments of the NEXT k-block are fetched while the current one feeds the matrix pipe.  hipcc re-materialises a plain
// C++ prefetch (it proves the carried value equals a load of this iteration's address and re-loads it at the loop top), so the
// prefetched block index is laundered through an empty asm: the value then has to be carried in registers.  The K loops
// are unrolled by two over two register sets so that carrying costs no copies.
__device__ __forceinline__ int uvm_opaque(int k)
{
    asm volatile("" : "+s"(k));
    return k;
}

struct UvmLayer {
    int col0;       // first LDS column read
    int kp;         // padded K (multiple of 8)
    int64_t w_off;  // float offset of packed weights
    int64_t b_off;  // float offset of bias
};
struct UvmPlan {
    int n_hidden;            // D
    int W;
    int in_ch;               // dims*(1+2L)
    int epad;                // in_ch padded: 48 or 64
    int dims;                // input dimensionality (2: uv, 3: xyz); set by the forward entry
    int out_ch;
    int64_t out_w_off, out_b_off;
    UvmLayer layer[UVM_MAX_LAYERS];
    int64_t wt_off[UVM_MAX_LAYERS];   // layer i >= 1: W_i[:, hidden part]^T packed as an MFMA B operand (backward chain)
    int64_t w16_off[UVM_MAX_LAYERS];  // layer i: the same weights as two fp16 planes (hi, lo x 2^11) in 32x32x16 B-fragment order, or -1
    int64_t wt16_off[UVM_MAX_LAYERS]; // layer i >= 1: wt_off's operand as two fp16 planes (backward chain of k_uvmlp_dgrad16), or -1
};

static int uvm_build_plan(int D, int W, int input_ch, int output_ch, int skip, UvmPlan &p, int64_t &total)
{
    if (D < 1 || D > UVM_MAX_LAYERS || W % 64 != 0 || W > 256 || input_ch < 1 || input_ch > UVM_EPAD3 || output_ch < 1 || output_ch > 4) return -1;
    const int EP = input_ch <= UVM_EPAD ? UVM_EPAD : UVM_EPAD3;
    p.n_hidden = D; p.W = W; p.in_ch = input_ch; p.out_ch = output_ch; p.epad = EP; p.dims = 2;
    int64_t off = 0;
    for (int i = 0; i < D; ++i) {
        UvmLayer &l = p.layer[i];
        if (i == 0) { l.col0 = 0; l.kp = EP; }
        else if (i == skip + 1) { l.col0 = 0; l.kp = EP + W; }
        else { l.col0 = EP; l.kp = W; }
        l.w_off = off; off += (int64_t)l.kp * W;
        l.b_off = off; off += W;
    }
    p.out_w_off = off; off += (int64_t)output_ch * W;
    p.out_b_off = off; off += 4;
    p.wt_off[0] = -1;
    for (int i = 1; i < D; ++i) { p.wt_off[i] = off; off += (int64_t)W * W; }
    // split-fp16 copies for the fast forward (k_uvmlp_fwd16): kp x W elements x (2 + 2) bytes = kp x W floats of space
    for (int i = 0; i < D; ++i) {
        if (W == 256 && EP == UVM_EPAD && p.layer[i].kp % 16 == 0) { p.w16_off[i] = off; off += (int64_t)p.layer[i].kp * W; }
        else p.w16_off[i] = -1;
    }
    for (int i = 0; i < D; ++i) {
        if (i >= 1 && W == 256 && EP == UVM_EPAD) { p.wt16_off[i] = off; off += (int64_t)W * W; }
        else p.wt16_off[i] = -1;
    }
    total = off;
    return 0;
}

extern "C" int64_t ctx_uvmlp_packed_bytes(int32_t D, int32_t W, int32_t input_ch, int32_t output_ch, int32_t skip)
{
    UvmPlan p; int64_t total = 0;
    if (uvm_build_plan(D, W, input_ch, output_ch, skip, p, total)) return -1;
    return total * 4;
}

// src: nn.Linear weight [W][kin]; dst packed [(kb*(W/32)+nb)*64+lane][4] with
// lane=(r,h): value j = weight[nb*32+r][map(kb*8+4h+j)].
__global__ void k_uvm_pack(const float *__restrict__ w, const float *__restrict__ b, int W, int kin, int kp,
                           int in_ch, int epad, int mode /*0 first,1 skip,2 plain*/, float *__restrict__ dw, float *__restrict__ db)
{
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t total = (int64_t)kp * W;
    if (idx < total) {
        int j = idx & 3;
        int lane = (idx >> 2) & 63;
        int64_t blk = idx >> 8;
        int nb = blk % (W / 32);
        int kb = blk / (W / 32);
        int r = lane & 31, h = lane >> 5;
        int k = kb * 8 + 4 * h + j;
        int n = nb * 32 + r;
        int src_k;
        if (mode == 2) src_k = k;
        else if (mode == 0) src_k = k < in_ch ? k : -1;
        else src_k = k < in_ch ? k : (k < epad ? -1 : k - epad + in_ch);
        dw[idx] = (src_k >= 0 && src_k < kin) ? w[(int64_t)n * kin + src_k] : 0.0f;
    }
    if (idx < W) db[idx] = b[idx];
}

// Backward-chain operand: dA_{i-1}[t][k] = sum_n dZ_i[t][n] * w[n][off + k]  ->  B[n][k]; packed like k_uvm_pack with the
// contraction index n in the place of k:  dst[(kb*(W/32)+nb)*64+lane][j] = w[kb*8+4h+j][off + nb*32 + r].
__global__ void k_uvm_pack_t(const float *__restrict__ w, int W, int kin, int off, float *__restrict__ dw)
{
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)W * W) return;
    int j = idx & 3;
    int lane = (idx >> 2) & 63;
    int64_t blk = idx >> 8;
    int nb = blk % (W / 32);
    int kb = blk / (W / 32);
    int r = lane & 31, h = lane >> 5;
    int n = kb * 8 + 4 * h + j;
    int k = nb * 32 + r;
    dw[idx] = w[(int64_t)n * kin + off + k];
}

// fp16 split of the layer weights for k_uvmlp_fwd16: w = hi + lo * 2^-11 with hi = rn16(w), lo = rn16((w - hi) * 2^11): 22 bits of
// mantissa in two fp16 planes.  Block (kb, nb) = hi[64 lanes][8] then lo[64 lanes][8]; lane (r, h) element j = w[nb*32 + r][map(kb*16 + 8h + j)]
// — the B fragment of v_mfma_f32_32x32x16_f16, one 16-byte load per lane and plane.
__global__ void k_uvm_pack16(const float *__restrict__ w, int W, int kin, int kp, int in_ch, int epad, int mode, f16 *__restrict__ dw)
{
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)kp * W) return;
    const int j = idx & 7, lane = (idx >> 3) & 63;
    const int64_t blk = idx >> 9;
    const int nb = blk % (W / 32), kb = blk / (W / 32);
    const int r = lane & 31, h = lane >> 5;
    const int k = kb * 16 + 8 * h + j, n = nb * 32 + r;
    int src_k;
    if (mode == 2) src_k = k;
    else if (mode == 0) src_k = k < in_ch ? k : -1;
    else src_k = k < in_ch ? k : (k < epad ? -1 : k - epad + in_ch);
    const float v = (src_k >= 0 && src_k < kin) ? w[(int64_t)n * kin + src_k] : 0.0f;
    const f16 hi = (f16)v;
    const f16 lo = (f16)((v - (float)hi) * 2048.0f);
    dw[blk * 1024 + lane * 8 + j] = hi;
    dw[blk * 1024 + 512 + lane * 8 + j] = lo;
}

// backward-chain operand of k_uvmlp_dgrad16: block (kb over the contraction index n, nb over the output column k) = hi[64][8] | lo[64][8],
// lane (r, h) element j = w[kb*16 + 8h + j][off + nb*32 + r]
__global__ void k_uvm_pack16_t(const float *__restrict__ w, int W, int kin, int off, f16 *__restrict__ dw)
{
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)W * W) return;
    const int j = idx & 7, lane = (idx >> 3) & 63;
    const int64_t blk = idx >> 9;
    const int nb = blk % (W / 32), kb = blk / (W / 32);
    const int r = lane & 31, h = lane >> 5;
    const float v = w[(int64_t)(kb * 16 + 8 * h + j) * kin + off + nb * 32 + r];
    const f16 hi = (f16)v;
    dw[blk * 1024 + lane * 8 + j] = hi;
    dw[blk * 1024 + 512 + lane * 8 + j] = (f16)((v - (float)hi) * 2048.0f);
}

__global__ void k_uvm_pack_out(const float *__restrict__ w, const float *__restrict__ b, int W, int out_ch,
                               float *__restrict__ dw, float *__restrict__ db)
{
    int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < out_ch * W) dw[idx] = w[idx];
    if (idx < 4) db[idx] = idx < out_ch ? b[idx] : 0.0f;
}

extern "C" int32_t ctx_uvmlp_pack(const float *const *ws, const float *const *bs, int32_t D, int32_t W,
                                  int32_t input_ch, int32_t output_ch, int32_t skip, void *packed, ctx_stream_t stream)
{
    UvmPlan p; int64_t total = 0;
    CTX_REQUIRE(ws && bs && packed, "uvmlp_pack: null pointer");
    CTX_REQUIRE(uvm_build_plan(D, W, input_ch, output_ch, skip, p, total) == 0,
                "uvmlp_pack: unsupported D=%d W=%d input_ch=%d output_ch=%d", D, W, input_ch, output_ch);
    float *dst = (float *)packed;
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < D; ++i) {
        int mode = i == 0 ? 0 : (i == skip + 1 ? 1 : 2);
        int kin = i == 0 ? input_ch : (i == skip + 1 ? input_ch + W : W);
        int64_t n = (int64_t)p.layer[i].kp * W;
        hipLaunchKernelGGL(k_uvm_pack, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, ws[i], bs[i], W, kin, p.layer[i].kp,
                           input_ch, p.epad, mode, dst + p.layer[i].w_off, dst + p.layer[i].b_off);
        if (i >= 1)
            hipLaunchKernelGGL(k_uvm_pack_t, dim3((unsigned)cdiv64((int64_t)W * W, 256)), dim3(256), 0, s, ws[i], W, kin,
                               i == skip + 1 ? input_ch : 0, dst + p.wt_off[i]);
        if (p.wt16_off[i] >= 0)
            hipLaunchKernelGGL(k_uvm_pack16_t, dim3((unsigned)cdiv64((int64_t)W * W, 256)), dim3(256), 0, s, ws[i], W, kin, i == skip + 1 ? input_ch : 0,
                               (f16 *)(dst + p.wt16_off[i]));
        if (p.w16_off[i] >= 0)
            hipLaunchKernelGGL(k_uvm_pack16, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, ws[i], W, kin, p.layer[i].kp, input_ch, p.epad, mode,
                               (f16 *)(dst + p.w16_off[i]));
    }
    hipLaunchKernelGGL(k_uvm_pack_out, dim3(cdiv(output_ch * W, 256)), dim3(256), 0, s, ws[D], bs[D], W, output_ch,
                       dst + p.out_w_off, dst + p.out_b_off);
    CTX_CHECK_LAUNCH("uvmlp_pack");
    return CTX_OK;
}

template <int W, int EP>
__global__ __launch_bounds__(W) void k_uvmlp_fwd(const float *__restrict__ uv, const float *__restrict__ emb, int64_t N, int res, int L,
                                                 const float *__restrict__ packed, UvmPlan plan,
                                                 float *__restrict__ raw, float *__restrict__ tex, float *__restrict__ saved)
{
    // 2-D field (EP = 48): embedding in columns [0,48), hidden vector in [48,48+W), 78.8 KB -> two workgroups per CU.
    // 3-D field (EP = 64): that layout would be 82.9 KB (one workgroup per CU), so the hidden vector overlays the embedding
    // ([0,W), 66.6 KB) and the skip layer re-creates the embedding in place after its hidden-part k-blocks.
    constexpr bool RC = EP > UVM_EPAD;
    constexpr int HOFF = RC ? 0 : EP;
    constexpr int STRIDE = HOFF + W + 4;   // 308 / 260 for W=256: 4 x odd -> conflict-free ds_read_b128 rows
    static_assert((STRIDE / 4) % 2 == 1, "row stride must be 4 x odd");
    static_assert(W >= EP || !RC, "the overlay needs W >= 64");
    extern __shared__ __attribute__((aligned(16))) float act[];   // [UVM_TM][STRIDE]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t n0 = (int64_t)blockIdx.x * UVM_TM;

    // ---- Fourier embedding into columns [0,EP) -------------------------------------------------
    auto put_embedding = [&]() {
        int row = tid & 63;
        int64_t n = n0 + row;
        const int d = plan.dims;
        float x0 = 0.f, x1 = 0.f, x2 = 0.f;
        if (n < N && !emb) {
            if (uv) { x0 = uv[n * d + 0]; x1 = uv[n * d + 1]; if (d > 2) x2 = uv[n * d + 2]; }
            else {
                // torch.linspace(0,1,res): start + i*step for the first half, end - (res-1-i)*step after
                int i = (int)(n / res), j = (int)(n % res);
                float step = 1.0f / (float)(res - 1);
                x0 = (j < res / 2) ? (float)j * step : 1.0f - (float)(res - 1 - j) * step;
                x1 = (i < res / 2) ? (float)i * step : 1.0f - (float)(res - 1 - i) * step;
            }
        }
        for (int e = wave; e < EP; e += W / 64) {
            float val = 0.f;
            if (emb) val = (e < plan.in_ch && n < N) ? emb[n * plan.in_ch + e] : 0.f;
            else if (e < d) val = e == 0 ? x0 : (e == 1 ? x1 : x2);
            else if (e < d * (1 + 2 * L)) {
                int l = (e - d) / (2 * d), q = (e - d) % (2 * d);      // [sin(2^l x) (d), cos(2^l x) (d)] per frequency
                int c = q % d;
                float a = (c == 0 ? x0 : (c == 1 ? x1 : x2)) * (float)(1 << l);
                val = (q >= d) ? cosf(a) : sinf(a);
            }
            act[row * STRIDE + e] = val;
        }
    };
    put_embedding();
    __syncthreads();
    if (saved) {   // training: keep the (padded) embedding for the weight gradients of layer 0 and the skip layer
        for (int i = tid; i < UVM_TM * (EP / 4); i += W) {
            int row = i / (EP / 4), c4 = i % (EP / 4);
            if (n0 + row < N) *(float4 *)(saved + (n0 + row) * EP + c4 * 4) = *(const float4 *)(act + row * STRIDE + c4 * 4);
        }
    }

    // ---- hidden layers on the f32 matrix pipe --------------------------------------------------
    for (int li = 0; li < plan.n_hidden; ++li) {
        const UvmLayer ly = plan.layer[li];
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
        const float4 *wp = (const float4 *)(packed + ly.w_off) + ((size_t)(wave * 2) * 64 + lane);
        constexpr size_t KBS = (size_t)(W / 32) * 64;   // float4 per k-block
        // k-blocks [kb0, kb1) of the layer's packed weights against LDS columns lc0 + 8 (kb - kb0) ...
        auto run_k = [&](int kb0, int kb1, int lc0) {
            const float *arow0 = act + r * STRIDE + lc0 + 4 * h - kb0 * 8;
            const float *arow1 = arow0 + 32 * STRIDE;
            auto kstep = [&](int kb, const float4 &b0, const float4 &b1) {
                float4 a0 = *(const float4 *)(arow0 + kb * 8);
                float4 a1 = *(const float4 *)(arow1 + kb * 8);
                const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
                const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[j], bv0[j], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[j], bv1[j], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[j], bv0[j], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[j], bv1[j], acc[1][1], 0, 0, 0);
                }
            };
            const float4 *p0 = wp + (size_t)kb0 * KBS;
            float4 wa0 = p0[0], wa1 = p0[64], wb0, wb1;
            for (int kb = kb0; kb < kb1; kb += 2) {          // every k-block range here has an even length
                const float4 *pb = wp + (size_t)uvm_opaque(kb + 1) * KBS;
                wb0 = pb[0]; wb1 = pb[64];
                __builtin_amdgcn_sched_barrier(0);           // keep the prefetch above the MFMAs it hides under
                kstep(kb, wa0, wa1);
                const float4 *pa = wp + (size_t)uvm_opaque(kb + 2 < kb1 ? kb + 2 : kb) * KBS;
                wa0 = pa[0]; wa1 = pa[64];
                __builtin_amdgcn_sched_barrier(0);
                kstep(kb + 1, wb0, wb1);
            }
        };
        if (!RC) run_k(0, ly.kp / 8, ly.col0);
        else if (ly.kp == EP + W) {                          // skip layer of the overlaid layout: hidden part, then embedding
            run_k(EP / 8, (EP + W) / 8, 0);
            __syncthreads();
            put_embedding();
            __syncthreads();
            run_k(0, EP / 8, 0);
        } else run_k(0, ly.kp / 8, 0);
        __syncthreads();   // everyone has finished reading this layer's input
        const float *bias = packed + ly.b_off;
        unsigned long long relu_bits = 0;   // bit (nb*2+mb)*16+q of this lane: its accumulator element is > 0
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            int col = wave * 64 + nb * 32 + r;
            float bv = bias[col];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    int row = mb * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                    float v = acc[mb][nb][q] + bv;
                    relu_bits |= (unsigned long long)(v > 0.f) << ((nb * 2 + mb) * 16 + q);
                    act[row * STRIDE + HOFF + col] = v > 0.f ? v : 0.f;
                }
        }
        if (saved) {   // the ReLU pattern in the accumulator layout the backward chain's tiles have: [layer][tile][thread] u64
            unsigned long long *mk = (unsigned long long *)(saved + N * (int64_t)(EP + plan.n_hidden * W));
            mk[((int64_t)li * gridDim.x + blockIdx.x) * W + tid] = relu_bits;
        }
        __syncthreads();
        if (saved) {   // post-ReLU activations [layer][texel][W], whole rows per texel
            float *dst = saved + N * EP + (int64_t)li * N * W;
            for (int i = tid; i < UVM_TM * (W / 4); i += W) {
                int row = i / (W / 4), c4 = i % (W / 4);
                if (n0 + row < N) *(float4 *)(dst + (n0 + row) * W + c4 * 4) = *(const float4 *)(act + row * STRIDE + HOFF + c4 * 4);
            }
        }
    }

    // ---- output layer (W -> out_ch <= 4) on the VALU, 4 lanes per texel ------------------------
    {
        constexpr int PARTS = W / 64;            // threads per texel
        int row = tid / PARTS, part = tid % PARTS;
        const float *a = act + row * STRIDE + HOFF + part * 64;
        const float *ow = packed + plan.out_w_off + part * 64;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < 64; ++k) {
            float x = a[k];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < plan.out_ch) s[c] += x * ow[c * W + k];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            for (int o = 1; o < PARTS; o <<= 1) s[c] += __shfl_xor(s[c], o, 64);
        int64_t n = n0 + row;
        if (part == 0 && n < N) {
            for (int c = 0; c < plan.out_ch; ++c) {
                float v = s[c] + packed[plan.out_b_off + c];
                raw[n * plan.out_ch + c] = v;
                if (tex) tex[(int64_t)c * N + n] = (tanhf(v) + 1.0f) / 2.0f;
            }
        }
    }
}


// =====================================================================================================================
// Fast forward of the 2-D texture field (W = 256): the same network on the 16-bit matrix pipe with SPLIT operands.
// Every activation and weight is carried as two fp16 numbers, x = hi + lo * 2^-11 (hi = rn16(x), lo = rn16((x - hi) * 2^11)): 22 bits
// of mantissa, i.e. fp32-grade operands.  A product a.w = ah.wh + 2^-11 (ah.wl + al.wh) + 2^-22 al.wl: the first three terms are three
// v_mfma_f32_32x32x16_f16 passes into TWO fp32 accumulator sets (main, cross), the last (relative 2^-22) is dropped.  Per 16-deep k-step
// that is 3 x 32 cycles against 8 x 64 for v_mfma_f32_32x32x2_f32: 5.3x the matrix rate at the f32 path's accuracy (measured against
// the reference's vectors at the same tolerances; CTX_UVMLP_EXACT_F32=1 selects the exact-f32 kernel).
// One workgroup = 128 texels, 4 waves; wave w owns hidden columns [64w, 64w+64) as 4 x 2 accumulator tiles of 32 x 32 (x 2 sets), one
// wave per SIMD, so a weight fragment fetched from L2 serves 128 texels.  Activations live in LDS as two fp16 planes [128][312].
// `saved` gets what k_uvmlp_fwd leaves (embedding, post-ReLU activations as hi + lo * 2^-11, ReLU bit masks per 64-texel tile).
#define UVM16_TM 64
#define UVM16_LO 312              // halves: a row's lo plane sits this far behind its hi plane (48 + 256 + 8)
#define UVM16_STRIDE 632          // halves per row PAIR (hi | lo): 1264 bytes = 16 x odd -> conflict-free ds_read_b128 over 16 rows, and
                                  // the lo plane within the 16-bit immediate offset of every DS access to the hi plane
__global__ __launch_bounds__(256, 2) void k_uvmlp_fwd16(const float *__restrict__ uv, const float *__restrict__ emb, int64_t N, int res, int L,
                                                     const float *__restrict__ packed, UvmPlan plan,
                                                     float *__restrict__ raw, float *__restrict__ tex, float *__restrict__ saved)
{
    constexpr int W = 256, EP = UVM_EPAD, STRIDE = UVM16_STRIDE;
    extern __shared__ __attribute__((aligned(16))) f16 pl[];      // [128 rows][hi 312 | lo 312 | pad 8] halves
    f16 *phi = pl, *plo = pl + UVM16_LO;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t n0 = (int64_t)blockIdx.x * UVM16_TM;
    auto put = [&](int row, int col, float v) {
        const f16 hi = (f16)v;
        phi[row * STRIDE + col] = hi;
        plo[row * STRIDE + col] = (f16)((v - (float)hi) * 2048.0f);
    };
    auto get = [&](int row, int col) { return (float)phi[row * STRIDE + col] + (float)plo[row * STRIDE + col] * (1.0f / 2048.0f); };

    // ---- Fourier embedding into columns [0,48): four threads per texel, 12 columns each ------------------------------
    {
        const int row = tid >> 2, half = tid & 3;
        const int64_t n = n0 + row;
        const int d = plan.dims;
        float x0 = 0.f, x1 = 0.f;
        if (n < N && !emb) {
            if (uv) { x0 = uv[n * d + 0]; x1 = uv[n * d + 1]; }
            else {
                int i = (int)(n / res), j = (int)(n % res);
                float step = 1.0f / (float)(res - 1);
                x0 = (j < res / 2) ? (float)j * step : 1.0f - (float)(res - 1 - j) * step;
                x1 = (i < res / 2) ? (float)i * step : 1.0f - (float)(res - 1 - i) * step;
            }
        }
        for (int e = half * (EP / 4); e < (half + 1) * (EP / 4); ++e) {
            float val = 0.f;
            if (emb) val = (e < plan.in_ch && n < N) ? emb[n * plan.in_ch + e] : 0.f;
            else if (e < d) val = e == 0 ? x0 : x1;
            else if (e < d * (1 + 2 * L)) {
                int l = (e - d) / (2 * d), q = (e - d) % (2 * d);
                float a = ((q % d) == 0 ? x0 : x1) * (float)(1 << l);
                val = (q >= d) ? cosf(a) : sinf(a);
            }
            put(row, e, val);
        }
    }
    __syncthreads();
    if (saved) {
        for (int i = tid; i < UVM16_TM * EP; i += 256) {
            int row = i / EP, c = i % EP;
            if (n0 + row < N) saved[(n0 + row) * EP + c] = get(row, c);
        }
    }

    for (int li = 0; li < plan.n_hidden; ++li) {
        const UvmLayer ly = plan.layer[li];
        const int nkb = ly.kp / 16;
        const f16 *ah_base = phi + r * STRIDE + ly.col0 + 8 * h;
        const f16 *al_base = plo + r * STRIDE + ly.col0 + 8 * h;
        const float *bias = packed + ly.b_off;
        unsigned long long bits0 = 0;              // ReLU pattern of the tile, in k_uvmlp_fwd's bit layout
        uint32_t outv[2][2][16];                   // this layer's outputs (hi | lo << 16) wait in registers until every wave has read its inputs
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {           // the wave's two 32-column blocks one after the other: 64 accumulator registers
            f32x16 acc[2], acx[2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int q = 0; q < 16; ++q) { acc[a][q] = 0.f; acx[a][q] = 0.f; }
            // weight fragments: block (kb, nbg) at ((kb * 8 + nbg) * 1024) halves, hi[64][8] then lo[64][8]
            const f16 *wbase = (const f16 *)(packed + plan.w16_off[li]) + (size_t)(wave * 2 + nb) * 1024 + lane * 8;
            f16x8 b0h, b0l, b1h, b1l, b2h, b2l;    // k-steps kb, kb+1, kb+2: two steps of prefetch (one wave per SIMD hides nothing)
            auto load_b = [&](int kb, f16x8 &xh, f16x8 &xl) {
                const f16 *p = wbase + (size_t)uvm_opaque(kb < nkb ? kb : nkb - 1) * 8 * 1024;
                xh = *(const f16x8 *)(p); xl = *(const f16x8 *)(p + 512);
            };
            load_b(0, b0h, b0l); load_b(1, b1h, b1l);
            f16x8 ah[2], al[2], nah[2], nal[2];
            auto load_a = [&](int kb, f16x8 (&xh)[2], f16x8 (&xl)[2]) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    xh[mb] = *(const f16x8 *)(ah_base + mb * 32 * STRIDE + kb * 16);
                    xl[mb] = *(const f16x8 *)(al_base + mb * 32 * STRIDE + kb * 16);
                }
            };
            load_a(0, ah, al);
            // One k-step: the next step's activation fragments (LDS) and the weights two steps ahead (L2) are requested before this
            // step's MFMAs issue — with one wave per SIMD nothing else covers their latency.  The loop walks two steps per trip over
            // the two activation register sets (no copies); only the three small weight sets rotate.
            auto step = [&](int kb, f16x8 (&ch)[2], f16x8 (&cl)[2], f16x8 (&nh)[2], f16x8 (&nl)[2]) {
                load_b(kb + 2, b2h, b2l);
                load_a(kb + 1 < nkb ? kb + 1 : kb, nh, nl);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch[mb], b0h, acc[mb], 0, 0, 0);
                    acx[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch[mb], b0l, acx[mb], 0, 0, 0);
                    acx[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl[mb], b0h, acx[mb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                b0h = b1h; b0l = b1l; b1h = b2h; b1l = b2l;
            };
            int kb = 0;
            for (; kb + 2 <= nkb; kb += 2) {
                step(kb, ah, al, nah, nal);
                step(kb + 1, nah, nal, ah, al);
            }
            if (kb < nkb) step(kb, ah, al, nah, nal);       // K = 48 and 304 are an odd number of 16-deep steps
            const float bv = bias[wave * 64 + nb * 32 + r];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float v = acc[mb][q] + acx[mb][q] * (1.0f / 2048.0f) + bv;
                    bits0 |= (unsigned long long)(v > 0.f) << ((nb * 2 + mb) * 16 + q);
                    const float y = v > 0.f ? v : 0.f;
                    const f16 yh = (f16)y;
                    const f16 yl = (f16)((y - (float)yh) * 2048.0f);
                    outv[nb][mb][q] = (uint32_t)__builtin_bit_cast(unsigned short, yh) | ((uint32_t)__builtin_bit_cast(unsigned short, yl) << 16);
                }
        }
        __syncthreads();                       // everyone has finished reading this layer's input
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int col = wave * 64 + nb * 32 + r;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int o = (mb * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * STRIDE + EP + col;
                    ((unsigned short *)phi)[o] = (unsigned short)(outv[nb][mb][q] & 0xffffu);
                    ((unsigned short *)plo)[o] = (unsigned short)(outv[nb][mb][q] >> 16);
                }
        }
        if (saved) {
            unsigned long long *mk = (unsigned long long *)(saved + N * (int64_t)(EP + plan.n_hidden * W));
            mk[((int64_t)li * gridDim.x + blockIdx.x) * W + tid] = bits0;
        }
        __syncthreads();
        if (saved) {   // post-ReLU activations [layer][texel][W] as the value the next layer consumes (hi + lo 2^-11)
            float *dst = saved + N * EP + (int64_t)li * N * W;
            for (int i = tid; i < UVM16_TM * (W / 4); i += 256) {
                const int row = i / (W / 4), c4 = i % (W / 4);
                if (n0 + row < N) {
                    const f16x4 vh = *(const f16x4 *)(phi + row * STRIDE + EP + c4 * 4), vl = *(const f16x4 *)(plo + row * STRIDE + EP + c4 * 4);
                    float4 o;
                    o.x = (float)vh[0] + (float)vl[0] * (1.0f / 2048.0f); o.y = (float)vh[1] + (float)vl[1] * (1.0f / 2048.0f);
                    o.z = (float)vh[2] + (float)vl[2] * (1.0f / 2048.0f); o.w = (float)vh[3] + (float)vl[3] * (1.0f / 2048.0f);
                    *(float4 *)(dst + (n0 + row) * W + c4 * 4) = o;
                }
            }
        }
    }

    // ---- output layer (256 -> out_ch <= 4) on the VALU, 4 threads per texel ------------------------------------------
    {
        const int row = tid >> 2, part = tid & 3;
        const float *ow = packed + plan.out_w_off + part * 64;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < 64; k += 8) {
            const f16x8 vh = *(const f16x8 *)(phi + row * STRIDE + EP + part * 64 + k), vl = *(const f16x8 *)(plo + row * STRIDE + EP + part * 64 + k);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = (float)vh[j] + (float)vl[j] * (1.0f / 2048.0f);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < plan.out_ch) s[c] += x * ow[c * W + k + j];
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) { s[c] += __shfl_xor(s[c], 1, 64); s[c] += __shfl_xor(s[c], 2, 64); }
        const int64_t n = n0 + row;
        if (part == 0 && n < N) {
            for (int c = 0; c < plan.out_ch; ++c) {
                float v = s[c] + packed[plan.out_b_off + c];
                raw[n * plan.out_ch + c] = v;
                if (tex) tex[(int64_t)c * N + n] = (tanhf(v) + 1.0f) / 2.0f;
            }
        }
    }
}

static inline int uvm_epad(int input_ch) { return input_ch <= UVM_EPAD ? UVM_EPAD : UVM_EPAD3; }

extern "C" int64_t ctx_uvmlp_saved_bytes(int64_t N, int32_t D, int32_t W, int32_t input_ch)
{
    if (N <= 0 || D < 1 || D > UVM_MAX_LAYERS || W % 64 != 0 || W > 256 || input_ch < 1 || input_ch > UVM_EPAD3) return -1;
    return N * (int64_t)(uvm_epad(input_ch) + D * W) * 4 + (int64_t)D * cdiv64(N, UVM_TM) * W * 8;   // + ReLU bit masks
}

extern "C" int32_t ctx_uvmlp_fwd_save(const float *uv, const float *emb, int64_t N, int32_t res, const void *packed, int32_t D, int32_t W,
                                      int32_t dims, int32_t L, int32_t output_ch, int32_t skip, float *raw, float *tex_chw, void *saved_v,
                                      ctx_stream_t stream)
{
    float *saved = (float *)saved_v;
    UvmPlan p; int64_t total = 0;
    CTX_REQUIRE(packed && raw && N > 0, "uvmlp_fwd: bad args");
    CTX_REQUIRE(dims == 2 || dims == 3, "uvmlp_fwd: dims=%d (2: uv, 3: xyz)", dims);
    CTX_REQUIRE(uv || emb || (dims == 2 && res > 1 && (int64_t)res * res == N),
                "uvmlp_fwd: no inputs needs dims == 2 and N == res*res (N=%lld res=%d)", (long long)N, res);
    int input_ch = dims * (1 + 2 * L);
    CTX_REQUIRE(uvm_build_plan(D, W, input_ch, output_ch, skip, p, total) == 0,
                "uvmlp_fwd: unsupported D=%d W=%d dims=%d L=%d output_ch=%d", D, W, dims, L, output_ch);
    p.dims = dims;
    hipStream_t s = (hipStream_t)stream;
    unsigned grid = (unsigned)cdiv64(N, UVM_TM);
    size_t lds = (size_t)UVM_TM * ((p.epad == UVM_EPAD ? UVM_EPAD : 0) + W + 4) * 4;   // the 3-D layout overlays the embedding
    const float *pk = (const float *)packed;
    {
        // the split-fp16 kernel for the reference's texture field (2-D, W = 256) unless the exact-f32 pipe is asked for
        const char *ex = getenv("CTX_UVMLP_EXACT_F32");
        bool fast = dims == 2 && W == 256 && p.epad == UVM_EPAD && !(ex && ex[0] == '1');
        for (int i = 0; i < D && fast; ++i) fast = p.w16_off[i] >= 0;
        if (fast) {
            const size_t lds16 = (size_t)UVM16_TM * UVM16_STRIDE * sizeof(f16);
            (void)hipFuncSetAttribute((const void *)k_uvmlp_fwd16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
            hipLaunchKernelGGL(k_uvmlp_fwd16, dim3((unsigned)cdiv64(N, UVM16_TM)), dim3(256), lds16, s, uv, emb, N, res, L, pk, p, raw, tex_chw, saved);
            CTX_CHECK_LAUNCH("uvmlp_fwd16");
            return CTX_OK;
        }
    }
#define UVM_FWD(WW, EE)                                                                                                     \
    do {                                                                                                                    \
        (void)hipFuncSetAttribute((const void *)k_uvmlp_fwd<WW, EE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_uvmlp_fwd<WW, EE>), dim3(grid), dim3(WW), lds, s, uv, emb, N, res, L, pk, p, raw, tex_chw, saved); \
    } while (0)
    if (p.epad == UVM_EPAD) {
        if (W == 256) UVM_FWD(256, UVM_EPAD); else if (W == 128) UVM_FWD(128, UVM_EPAD); else UVM_FWD(64, UVM_EPAD);
    } else {
        if (W == 256) UVM_FWD(256, UVM_EPAD3); else if (W == 128) UVM_FWD(128, UVM_EPAD3); else UVM_FWD(64, UVM_EPAD3);
    }
#undef UVM_FWD
    CTX_CHECK_LAUNCH("uvmlp_fwd");
    return CTX_OK;
}

extern "C" int32_t ctx_uvmlp_fwd(const float *uv, const float *emb, int64_t N, int32_t res, const void *packed, int32_t D, int32_t W,
                                 int32_t L, int32_t output_ch, int32_t skip, float *raw, float *tex_chw,
                                 ctx_stream_t stream)
{
    return ctx_uvmlp_fwd_save(uv, emb, N, res, packed, D, W, 2, L, output_ch, skip, raw, tex_chw, nullptr, stream);
}

// =====================================================================================================================
// Backward of the texture field (autograd of NeRF2D.forward, src/run_nerf_helpers.py:106-135, as driven by the SDS loop
// src/training/trainer.py:644-907: atlas gradient -> tanh -> MLP -> parameter gradients; uv is not trainable).
//
// Two phases, both on the exact-f32 matrix pipe:
//   k_uvmlp_dgrad : per 64-texel tile, the chain dA_7 = draw . Wout; dZ_i = dA_i * (A_i > 0); dA_{i-1} = dZ_i . W_i[:, hidden]
//                   with dA resident in LDS; every dZ_i goes to HBM [layer][texel][W] (whole rows), and the output layer's
//                   weight / bias gradients are accumulated on the VALU on the way.
//   k_uvmlp_wgrad : dW_i = dZ_i^T . In_i as a split-K GEMM over the texels: a workgroup owns the whole W x cols gradient
//                   in registers (256 accumulator VGPRs per lane for 256 x 256) and walks its texel range two texels per
//                   MFMA k-step; both operands are texel-major, so a lane's operand is ONE dword of a 128-byte row segment
//                   straight from global memory (no LDS, no transposes).  db_i = column sums of dZ_i ride along.
//                   Partial gradients land in per-workgroup slabs and are summed in a fixed order (deterministic).
// =====================================================================================================================
#define UVM_WG_GROUPS 256       // workgroups (= texel ranges) of the weight-gradient GEMMs
#define UVM_WG_GROUPS_EMB 768   // the 48-column embedding gradients run 3 workgroups per CU
#define UVM_DGRAD_GRID 512      // persistent dgrad workgroups (2 per CU)

template <int W>
__global__ __launch_bounds__(W, 2) void k_uvmlp_dgrad(const float *__restrict__ grad_raw, const float *__restrict__ grad_tex,
                                                   const float *__restrict__ raw, int64_t N, const float *__restrict__ packed,
                                                   UvmPlan plan, const float *__restrict__ saved, float *__restrict__ dz,
                                                   float *__restrict__ part_w /*[grid][4][W]*/, float *__restrict__ part_b /*[grid][4]*/)
{
    constexpr int STRIDE = W + 4;               // 4 x odd
    constexpr int C4N = W / 4;                  // float4 per row; W threads -> 4 rows per pass, 16 passes per tile
    extern __shared__ __attribute__((aligned(16))) float g[];   // [64][STRIDE] dA / dZ, then dr[64][4]
    float *dr = g + UVM_TM * STRIDE;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int c4 = tid % C4N, rg = tid / C4N;
    const int D = plan.n_hidden;
    const float *acts = saved + N * plan.epad;
    const unsigned long long *masks = (const unsigned long long *)(saved + N * (int64_t)(plan.epad + D * W));

    float wo[4][4], gwo[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            wo[c][j] = c < plan.out_ch ? packed[plan.out_w_off + c * W + c4 * 4 + j] : 0.f;
            gwo[c][j] = 0.f;
        }
    float gbo = 0.f;                            // this thread's channel is tid & 3

    const int64_t ntiles = (N + UVM_TM - 1) / UVM_TM;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t n0 = tile * UVM_TM;
        // ---- d loss / d raw for the tile's 64 texels --------------------------------------------
        for (int i = tid; i < UVM_TM * 4; i += W) {
            int t = i >> 2, c = i & 3;
            int64_t n = n0 + t;
            float v = 0.f;
            if (n < N && c < plan.out_ch) {
                if (grad_raw) v = grad_raw[n * plan.out_ch + c];
                if (grad_tex) {
                    float y = tanhf(raw[n * plan.out_ch + c]);
                    v += grad_tex[(int64_t)c * N + n] * 0.5f * (1.0f - y * y);
                }
            }
            dr[i] = v;
            gbo += v;
        }
        __syncthreads();
        // ---- output layer: dA = draw . Wout, dWout += draw^T . A, dZ = dA * (A > 0) -----------
        {
            const float *a_top = acts + (int64_t)(D - 1) * N * W;
            float *dz_top = dz + (int64_t)(D - 1) * N * W;
#pragma unroll 4
            for (int p = 0; p < UVM_TM / 4; ++p) {
                int t = p * 4 + rg;
                int64_t n = n0 + t;
                int64_t nc = n < N ? n : N - 1;
                float4 a = *(const float4 *)(a_top + nc * W + c4 * 4);
                float4 d = *(const float4 *)(dr + t * 4);
                const float av[4] = {a.x, a.y, a.z, a.w}, dv[4] = {d.x, d.y, d.z, d.w};
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float da = dv[0] * wo[0][j];
                    da = fmaf(dv[1], wo[1][j], da);
                    da = fmaf(dv[2], wo[2][j], da);
                    da = fmaf(dv[3], wo[3][j], da);
                    o[j] = av[j] > 0.f ? da : 0.f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) gwo[c][j] = fmaf(dv[c], av[j], gwo[c][j]);
                }
                float4 ov = make_float4(o[0], o[1], o[2], o[3]);
                *(float4 *)(g + t * STRIDE + c4 * 4) = ov;
                if (n < N) *(float4 *)(dz_top + n * W + c4 * 4) = ov;
            }
        }
        __syncthreads();
        // ---- hidden layers, last to second ------------------------------------------------------
        for (int li = D - 1; li >= 1; --li) {
            float *dz_prev = dz + (int64_t)(li - 1) * N * W;
            const unsigned long long relu_bits = masks[((int64_t)(li - 1) * ntiles + tile) * W + tid];
            f32x16 acc[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
            const float4 *wp = (const float4 *)(packed + plan.wt_off[li]) + ((size_t)(wave * 2) * 64 + lane);
            const float *arow0 = g + r * STRIDE + 4 * h;
            const float *arow1 = arow0 + 32 * STRIDE;
            auto kstep = [&](int kb, const float4 &b0, const float4 &b1) {
                float4 a0 = *(const float4 *)(arow0 + kb * 8);
                float4 a1 = *(const float4 *)(arow1 + kb * 8);
                const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
                const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[j], bv0[j], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[j], bv1[j], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[j], bv0[j], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[j], bv1[j], acc[1][1], 0, 0, 0);
                }
            };
            constexpr size_t KBS = (size_t)(W / 32) * 64;
            float4 wa0 = wp[0], wa1 = wp[64], wb0, wb1;
            for (int kb = 0; kb < W / 8; kb += 2) {
                const float4 *pb = wp + (size_t)uvm_opaque(kb + 1) * KBS;
                wb0 = pb[0]; wb1 = pb[64];
                __builtin_amdgcn_sched_barrier(0);
                kstep(kb, wa0, wa1);
                const float4 *pa = wp + (size_t)uvm_opaque(kb + 2 < W / 8 ? kb + 2 : kb) * KBS;
                wa0 = pa[0]; wa1 = pa[64];
                __builtin_amdgcn_sched_barrier(0);
                kstep(kb + 1, wb0, wb1);
            }
            __syncthreads();                     // all fragment reads of dZ_li (and the row copies of it below) are done
            // dZ_{li-1} = dA_{li-1} * relu'(layer li-1): the forward left the pattern in this very accumulator layout
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                int col = wave * 64 + nb * 32 + r;
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        int row = mb * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                        bool on = (relu_bits >> ((nb * 2 + mb) * 16 + q)) & 1ull;
                        g[row * STRIDE + col] = on ? acc[mb][nb][q] : 0.f;
                    }
            }
            __syncthreads();
            // whole rows of dZ_{li-1} to HBM; the next layer's fragment reads of g run alongside (both only read)
#pragma unroll 4
            for (int p = 0; p < UVM_TM / 4; ++p) {
                int t = p * 4 + rg;
                int64_t n = n0 + t;
                if (n < N) *(float4 *)(dz_prev + n * W + c4 * 4) = *(const float4 *)(g + t * STRIDE + c4 * 4);
            }
        }
        __syncthreads();                         // the last row copies read g before the next tile overwrites it
    }
    // ---- output-layer gradients of this workgroup: fold the 4 row groups, one partial row per channel ----
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) g[(rg * 4 + c) * W + c4 * 4 + j] = gwo[c][j];
    __syncthreads();
    for (int c = 0; c < 4; ++c) {
        float s = g[(0 * 4 + c) * W + tid];
        s += g[(1 * 4 + c) * W + tid];
        s += g[(2 * 4 + c) * W + tid];
        s += g[(3 * 4 + c) * W + tid];
        part_w[((int64_t)blockIdx.x * 4 + c) * W + tid] = s;
    }
    dr[tid] = gbo;
    __syncthreads();
    if (tid < 4) {
        float s = 0.f;
        for (int i = tid; i < W; i += 4) s += dr[i];
        part_b[(int64_t)blockIdx.x * 4 + tid] = s;
    }
}

// ---- the dZ chain on the 16-bit matrix pipe with split operands (the backward twin of k_uvmlp_fwd16) ---------------------------------
// Same phases and outputs as k_uvmlp_dgrad<256>; the chain's GEMMs dA_{i-1} = dZ_i . W_i[:, hidden] run as three v_mfma_f32_32x32x16_f16
// passes over (hi, lo) planes of dZ (LDS) and of the weights (L2).  Gradients have no natural scale (1e-8 under a mean-reduced loss):
// every op of the chain is linear, so the tile's draw is multiplied by 2^e with e from a device reduction of max|grad| (the largest
// draw lands near 16: room for a 4 000-fold growth through the layers before fp16 overflows, 2^-35 of the maximum still resolved), and
// dZ goes to HBM multiplied by 2^-e (exact).
struct UvmCtl { uint32_t absmax_raw, absmax_tex; int32_t e; int32_t pad; };
__global__ __launch_bounds__(256) void k_uvm_absmax(const float *__restrict__ a, int64_t na, const float *__restrict__ b, int64_t nb_, UvmCtl *__restrict__ ctl)
{
    uint32_t ma = 0, mb = 0;
    const int64_t stride = (int64_t)gridDim.x * 256, i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a) for (int64_t i = i0; i < na; i += stride) ma = max(ma, __float_as_uint(a[i]) & 0x7fffffffu);
    if (b) for (int64_t i = i0; i < nb_; i += stride) mb = max(mb, __float_as_uint(b[i]) & 0x7fffffffu);
    for (int o = 32; o; o >>= 1) { ma = max(ma, (uint32_t)__shfl_xor((int)ma, o)); mb = max(mb, (uint32_t)__shfl_xor((int)mb, o)); }
    __shared__ uint32_t s_m[2][4];                 // one atomic pair per workgroup (atomics on one address serialise)
    if ((threadIdx.x & 63) == 0) { s_m[0][threadIdx.x >> 6] = ma; s_m[1][threadIdx.x >> 6] = mb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ma = max(max(s_m[0][0], s_m[0][1]), max(s_m[0][2], s_m[0][3]));
        mb = max(max(s_m[1][0], s_m[1][1]), max(s_m[1][2], s_m[1][3]));
        if (ma) atomicMax(&ctl->absmax_raw, ma);
        if (mb) atomicMax(&ctl->absmax_tex, mb);
    }
}
__global__ void k_uvm_scale(UvmCtl *ctl)
{
    // |draw| <= |grad_raw| + 0.5 |grad_tex| (tanh' <= 1); non-finite or zero gradients: no scaling (they propagate as they are)
    const float bound = __uint_as_float(ctl->absmax_raw) + 0.5f * __uint_as_float(ctl->absmax_tex);
    int e = 0;
    if (bound > 0.f && bound < INFINITY) { int x; (void)frexpf(bound, &x); e = 4 - x; }
    ctl->e = e;
}

#define UVM16B_LO 264             // halves: dZ's lo plane behind its hi plane (256 + 8)
#define UVM16B_STRIDE 536         // halves per row pair: 1072 bytes = 16 x odd
__global__ __launch_bounds__(256, 2) void k_uvmlp_dgrad16(const float *__restrict__ grad_raw, const float *__restrict__ grad_tex,
                                                          const float *__restrict__ raw, int64_t N, const float *__restrict__ packed,
                                                          UvmPlan plan, const float *__restrict__ saved, float *__restrict__ dz,
                                                          float *__restrict__ part_w, float *__restrict__ part_b, const UvmCtl *__restrict__ ctl)
{
    constexpr int W = 256, STRIDE = UVM16B_STRIDE, C4N = W / 4;
    extern __shared__ __attribute__((aligned(16))) f16 gp[];       // [64][hi 264 | lo 264 | 8] halves, then dr[64][4] floats
    f16 *ghi = gp, *glo = gp + UVM16B_LO;
    float *dr = (float *)(gp + UVM_TM * STRIDE);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int c4 = tid % C4N, rg = tid / C4N;
    const int D = plan.n_hidden;
    const float *acts = saved + N * plan.epad;
    const unsigned long long *masks = (const unsigned long long *)(saved + N * (int64_t)(plan.epad + D * W));
    const int e = ctl->e;

    float wo[4][4], gwo[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            wo[c][j] = c < plan.out_ch ? packed[plan.out_w_off + c * W + c4 * 4 + j] : 0.f;
            gwo[c][j] = 0.f;
        }
    float gbo = 0.f;

    const int64_t ntiles = (N + UVM_TM - 1) / UVM_TM;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t n0 = tile * UVM_TM;
        for (int i = tid; i < UVM_TM * 4; i += 256) {
            int t = i >> 2, c = i & 3;
            int64_t n = n0 + t;
            float v = 0.f;
            if (n < N && c < plan.out_ch) {
                if (grad_raw) v = grad_raw[n * plan.out_ch + c];
                if (grad_tex) {
                    float y = tanhf(raw[n * plan.out_ch + c]);
                    v += grad_tex[(int64_t)c * N + n] * 0.5f * (1.0f - y * y);
                }
            }
            dr[i] = v;
            gbo += v;
        }
        __syncthreads();
        // ---- output layer on the VALU (f32): dA = draw . Wout, dWout += draw^T . A, dZ = dA * (A > 0) -> HBM as is, LDS scaled and split
        {
            const float *a_top = acts + (int64_t)(D - 1) * N * W;
            float *dz_top = dz + (int64_t)(D - 1) * N * W;
#pragma unroll 4
            for (int p = 0; p < UVM_TM / 4; ++p) {
                int t = p * 4 + rg;
                int64_t n = n0 + t;
                int64_t nc = n < N ? n : N - 1;
                float4 a = *(const float4 *)(a_top + nc * W + c4 * 4);
                float4 d = *(const float4 *)(dr + t * 4);
                const float av[4] = {a.x, a.y, a.z, a.w}, dv[4] = {d.x, d.y, d.z, d.w};
                float o[4];
                f16x4 oh, ol;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float da = dv[0] * wo[0][j];
                    da = fmaf(dv[1], wo[1][j], da);
                    da = fmaf(dv[2], wo[2][j], da);
                    da = fmaf(dv[3], wo[3][j], da);
                    o[j] = av[j] > 0.f ? da : 0.f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) gwo[c][j] = fmaf(dv[c], av[j], gwo[c][j]);
                    const float sv = ldexpf(o[j], e);
                    oh[j] = (f16)sv;
                    ol[j] = (f16)((sv - (float)oh[j]) * 2048.0f);
                }
                *(f16x4 *)(ghi + t * STRIDE + c4 * 4) = oh;
                *(f16x4 *)(glo + t * STRIDE + c4 * 4) = ol;
                if (n < N) *(float4 *)(dz_top + n * W + c4 * 4) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
        __syncthreads();
        // ---- hidden layers, last to second ------------------------------------------------------------------------------
        for (int li = D - 1; li >= 1; --li) {
            float *dz_prev = dz + (int64_t)(li - 1) * N * W;
            const unsigned long long relu_bits = masks[((int64_t)(li - 1) * ntiles + tile) * W + tid];
            const f16 *ah_base = ghi + r * STRIDE + 8 * h, *al_base = glo + r * STRIDE + 8 * h;
            uint32_t outv[2][2][16];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                f32x16 acc[2], acx[2];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int q = 0; q < 16; ++q) { acc[a][q] = 0.f; acx[a][q] = 0.f; }
                const f16 *wbase = (const f16 *)(packed + plan.wt16_off[li]) + (size_t)(wave * 2 + nb) * 1024 + lane * 8;
                constexpr int nkb = W / 16;
                f16x8 b0h, b0l, b1h, b1l, b2h, b2l;
                auto load_b = [&](int kb, f16x8 &xh, f16x8 &xl) {
                    const f16 *p = wbase + (size_t)uvm_opaque(kb < nkb ? kb : nkb - 1) * 8 * 1024;
                    xh = *(const f16x8 *)(p); xl = *(const f16x8 *)(p + 512);
                };
                load_b(0, b0h, b0l); load_b(1, b1h, b1l);
                f16x8 ah[2], al[2], nah[2], nal[2];
                auto load_a = [&](int kb, f16x8 (&xh)[2], f16x8 (&xl)[2]) {
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        xh[mb] = *(const f16x8 *)(ah_base + mb * 32 * STRIDE + kb * 16);
                        xl[mb] = *(const f16x8 *)(al_base + mb * 32 * STRIDE + kb * 16);
                    }
                };
                load_a(0, ah, al);
                auto step = [&](int kb, f16x8 (&ch)[2], f16x8 (&cl)[2], f16x8 (&nh)[2], f16x8 (&nl)[2]) {
                    load_b(kb + 2, b2h, b2l);
                    load_a(kb + 1 < nkb ? kb + 1 : kb, nh, nl);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch[mb], b0h, acc[mb], 0, 0, 0);
                        acx[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch[mb], b0l, acx[mb], 0, 0, 0);
                        acx[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl[mb], b0h, acx[mb], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    b0h = b1h; b0l = b1l; b1h = b2h; b1l = b2l;
                };
                for (int kb = 0; kb < nkb; kb += 2) {
                    step(kb, ah, al, nah, nal);
                    step(kb + 1, nah, nal, ah, al);
                }
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const bool on = (relu_bits >> ((nb * 2 + mb) * 16 + q)) & 1ull;
                        const float y = on ? acc[mb][q] + acx[mb][q] * (1.0f / 2048.0f) : 0.f;
                        const f16 yh = (f16)y;
                        const f16 yl = (f16)((y - (float)yh) * 2048.0f);
                        outv[nb][mb][q] = (uint32_t)__builtin_bit_cast(unsigned short, yh) | ((uint32_t)__builtin_bit_cast(unsigned short, yl) << 16);
                    }
            }
            __syncthreads();                     // all fragment reads of dZ_li (and the row copies of it below) are done
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int col = wave * 64 + nb * 32 + r;
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int o = (mb * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * STRIDE + col;
                        ((unsigned short *)ghi)[o] = (unsigned short)(outv[nb][mb][q] & 0xffffu);
                        ((unsigned short *)glo)[o] = (unsigned short)(outv[nb][mb][q] >> 16);
                    }
            }
            __syncthreads();
            // whole rows of dZ_{li-1} to HBM, unscaled; the next layer's fragment reads run alongside (both only read)
#pragma unroll 4
            for (int p = 0; p < UVM_TM / 4; ++p) {
                int t = p * 4 + rg;
                int64_t n = n0 + t;
                if (n < N) {
                    const f16x4 vh = *(const f16x4 *)(ghi + t * STRIDE + c4 * 4), vl = *(const f16x4 *)(glo + t * STRIDE + c4 * 4);
                    float4 o;
                    o.x = ldexpf((float)vh[0] + (float)vl[0] * (1.0f / 2048.0f), -e); o.y = ldexpf((float)vh[1] + (float)vl[1] * (1.0f / 2048.0f), -e);
                    o.z = ldexpf((float)vh[2] + (float)vl[2] * (1.0f / 2048.0f), -e); o.w = ldexpf((float)vh[3] + (float)vl[3] * (1.0f / 2048.0f), -e);
                    *(float4 *)(dz_prev + n * W + c4 * 4) = o;
                }
            }
        }
        __syncthreads();                         // the last row copies read the planes before the next tile overwrites them
    }
    // ---- output-layer gradients of this workgroup: fold the 4 row groups, one partial row per channel ----
    float *gf = (float *)gp;                     // 16 x 256 floats
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) gf[(rg * 4 + c) * W + c4 * 4 + j] = gwo[c][j];
    __syncthreads();
    for (int c = 0; c < 4; ++c) {
        float s = gf[(0 * 4 + c) * W + tid];
        s += gf[(1 * 4 + c) * W + tid];
        s += gf[(2 * 4 + c) * W + tid];
        s += gf[(3 * 4 + c) * W + tid];
        part_w[((int64_t)blockIdx.x * 4 + c) * W + tid] = s;
    }
    dr[tid] = gbo;
    __syncthreads();
    if (tid < 4) {
        float s = 0.f;
        for (int i = tid; i < W; i += 4) s += dr[i];
        part_b[(int64_t)blockIdx.x * 4 + tid] = s;
    }
}

// dW[rows x cols] = dZ^T . In over the texel range of this workgroup.  Waves WR x WC, wave tile (32 NI) x (32 NJ).
// Row strides are compile-time (LDZ = W = rows, LDIN = W or 48) so the operand loads of a whole k-step group hang off two
// running pointers with immediate offsets; only the last, ragged texel range takes the bounds-checked path.
template <int WR, int WC, int NI, int NJ, int LDZ, int LDIN, int INCOLS>
__global__ __launch_bounds__(64 * WR * WC)
void k_uvmlp_wgrad(const float *__restrict__ dz, const float *__restrict__ in, int64_t N, int64_t chunk, float *__restrict__ slab,
                   float *__restrict__ bslab)
{
    constexpr int ROWS = WR * NI * 32, COLS = WC * NJ * 32;
    static_assert(ROWS == LDZ && COLS >= INCOLS, "workgroup owns the whole gradient");
    constexpr int U = 4;                         // k-steps (2 texels each) per buffer
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wr = wave / WC, wc = wave % WC;
    const int row0 = wr * NI * 32, col0 = wc * NJ * 32;
    const int64_t t_begin = (int64_t)blockIdx.x * chunk;
    const int64_t t_end = t_begin + chunk < N ? t_begin + chunk : N;

    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    float bs[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) bs[i] = 0.f;
    int bcol[NJ]; bool bok[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        int cc = col0 + j * 32 + r;
        bok[j] = cc < INCOLS;
        bcol[j] = bok[j] ? cc : INCOLS - 1;
    }
    float a0[U][NI], b0[U][NJ], a1[U][NI], b1[U][NJ];
    auto mma = [&](const float (&a)[U][NI], const float (&b)[U][NJ]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
                bs[i] += a[u][i];
            }
        }
    };
    if (t_begin + chunk <= N) {
        // ---- whole range: no bounds checks.  Three operand buffers of U k-steps: while buffer p feeds the matrix pipe, the
        // buffer consumed one phase ago is refilled (8 loads after each k-step's 16 MFMAs) and is used one phase later, so a
        // whole phase (U x 1024 MFMA cycles) covers the memory latency and the waits are vmcnt(one buffer), never 0.
        const float *zp = dz + (t_begin + h) * LDZ + row0 + r;
        const float *ip = in + (t_begin + h) * LDIN;
        float a2[U][NI], b2[U][NJ];
        auto load_step = [&](int u, float (&a)[U][NI], float (&b)[U][NJ]) {
#pragma unroll
            for (int i = 0; i < NI; ++i) a[u][i] = zp[u * 2 * LDZ + i * 32];
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[u][j] = ip[u * 2 * LDIN + bcol[j]];
        };
        auto phase = [&](const float (&a)[U][NI], const float (&b)[U][NJ], float (&na)[U][NI], float (&nb)[U][NJ]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float bv[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) bv[j] = (INCOLS == COLS || bok[j]) ? b[u][j] : 0.f;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], bv[j], acc[i][j], 0, 0, 0);
                    bs[i] += a[u][i];
                }
                load_step(u, na, nb);
                __builtin_amdgcn_sched_barrier(0);
            }
            zp += 2 * U * LDZ; ip += 2 * U * LDIN;
        };
        // prologue: buffers 0 and 1; zp / ip then point at the rows of the buffer to refill next
#pragma unroll
        for (int u = 0; u < U; ++u) load_step(u, a0, b0);
        zp += 2 * U * LDZ; ip += 2 * U * LDIN;
#pragma unroll
        for (int u = 0; u < U; ++u) load_step(u, a1, b1);
        zp += 2 * U * LDZ; ip += 2 * U * LDIN;
        __builtin_amdgcn_sched_barrier(0);
        // chunk is a multiple of 3 buffers (48 texels); the refills of the last two phases read (in-bounds or next-range)
        // rows that are never used: keep them in bounds by stepping back at the end
        const int rounds = (int)(chunk / (6 * U));
        const float *zlast = dz + (N - 2 * U + h) * LDZ + row0 + r;     // last whole buffer of the array
        const float *ilast = in + (N - 2 * U + h) * LDIN;
        for (int it = 0; it < rounds; ++it) {
            const bool tail = it + 1 == rounds;
            phase(a0, b0, a2, b2);
            if (tail) { zp = zlast; ip = ilast; }
            phase(a1, b1, a0, b0);
            if (tail) { zp = zlast; ip = ilast; }
            phase(a2, b2, a1, b1);
        }
    } else {
        auto fetch = [&](int64_t t, float (&a)[U][NI], float (&b)[U][NJ]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int64_t tt = t + 2 * u + h;
                bool ok = tt < t_end;
                int64_t tc = tt < N ? tt : N - 1;
                const float *zp = dz + tc * LDZ + row0 + r;
                const float *ip = in + tc * LDIN;
#pragma unroll
                for (int i = 0; i < NI; ++i) { float v = zp[i * 32]; a[u][i] = ok ? v : 0.f; }
#pragma unroll
                for (int j = 0; j < NJ; ++j) { float v = ip[bcol[j]]; b[u][j] = bok[j] ? v : 0.f; }
            }
        };
        fetch(t_begin, a0, b0);
        for (int64_t t = t_begin; t < t_end; t += 4 * U) {
            fetch(t + 2 * U, a1, b1);            // past-the-end fetches are clamped and zeroed
            mma(a0, b0);
            fetch(t + 4 * U, a0, b0);
            mma(a1, b1);
        }
    }
    float *sl = slab + (int64_t)blockIdx.x * ROWS * COLS;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                int n = row0 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                sl[n * COLS + col0 + j * 32 + r] = acc[i][j][q];
            }
    if (wc == 0 && bslab) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float v = bs[i] + __shfl_xor(bs[i], 32, 64);
            if (h == 0) bslab[(int64_t)blockIdx.x * ROWS + row0 + i * 32 + r] = v;
        }
    }
}

// out[n][col_off + k] = sum_g slab[g][n][k]  (fixed order), k < out_cols.  One thread per (4 columns, quarter of the groups):
// 16-byte loads, the four quarters folded through LDS in a fixed order.  cols_pad % 4 == 0.
__global__ __launch_bounds__(256) void k_uvm_reduce(const float *__restrict__ slab, int G, int64_t stride, int rows, int cols_pad,
                                                    int out_cols, float *__restrict__ out, int ld_out, int col_off)
{
    __shared__ float4 part[4][64];
    const int q = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int idx4 = blockIdx.x * 64 + l;                    // float4 index within rows x cols_pad
    const int total4 = rows * cols_pad / 4;
    const int gq = (G + 3) / 4;
    const int g0 = q * gq, g1 = (g0 + gq < G ? g0 + gq : G);
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (idx4 < total4) {
        const float4 *p = (const float4 *)slab + idx4;
        const int64_t st4 = stride / 4;
        int gi = g0;
        for (; gi + 2 <= g1; gi += 2) {
            float4 a = p[gi * st4], b = p[(gi + 1) * st4];
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
        }
        if (gi < g1) { float4 a = p[gi * st4]; s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w; }
    }
    part[q][l] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    __syncthreads();
    if (q == 0 && idx4 < total4) {
        float4 a = part[0][l], b = part[1][l], c = part[2][l], d = part[3][l];
        const float v[4] = {(a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w)};
        const int n = idx4 * 4 / cols_pad, k = idx4 * 4 % cols_pad;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k + j < out_cols) out[(int64_t)n * ld_out + col_off + k + j] = v[j];
    }
}

static inline int64_t uvm_align(int64_t x) { return (x + 255) & ~(int64_t)255; }

extern "C" int64_t ctx_uvmlp_bwd_ws_bytes(int64_t N, int32_t D, int32_t W)
{
    if (N <= 0 || D < 1 || D > UVM_MAX_LAYERS || W % 64 != 0 || W > 256) return -1;
    int64_t b = uvm_align((int64_t)D * N * W * 4);                       // dZ
    b += uvm_align((int64_t)UVM_WG_GROUPS * W * (W > 64 ? W : 64) * 4);  // weight-gradient slabs
    b += uvm_align((int64_t)UVM_WG_GROUPS_EMB * W * 4);                  // bias slabs
    b += uvm_align((int64_t)UVM_DGRAD_GRID * 4 * W * 4);                 // output-layer weight partials
    b += uvm_align((int64_t)UVM_DGRAD_GRID * 4 * 4);                     // output-layer bias partials
    b += 256;                                                            // control words of the split-fp16 chain (gradient scale)
    return b;
}

template <int WR, int WC, int NI, int NJ, int LDZ, int LDIN, int INCOLS>
static void uvm_launch_wgrad(int G, const float *dz, const float *in, int64_t N, int64_t chunk, float *slab, float *bslab, hipStream_t s)
{
    hipLaunchKernelGGL((k_uvmlp_wgrad<WR, WC, NI, NJ, LDZ, LDIN, INCOLS>), dim3(G), dim3(64 * WR * WC), 0, s, dz, in, N, chunk, slab,
                       bslab);
}

extern "C" int32_t ctx_uvmlp_bwd(const float *grad_raw, const float *grad_tex, const float *raw, int64_t N, const void *packed,
                                 int32_t D, int32_t W, int32_t dims, int32_t L, int32_t output_ch, int32_t skip, const void *saved_v, void *ws,
                                 float *const *gws, float *const *gbs, ctx_stream_t stream)
{
    UvmPlan p; int64_t total = 0;
    CTX_REQUIRE(packed && saved_v && ws && gws && gbs && N > 0, "uvmlp_bwd: bad args");
    CTX_REQUIRE(grad_raw || grad_tex, "uvmlp_bwd: need grad_raw and / or grad_tex");
    CTX_REQUIRE(!grad_tex || raw, "uvmlp_bwd: grad_tex needs raw (the tanh argument)");
    CTX_REQUIRE(dims == 2 || dims == 3, "uvmlp_bwd: dims=%d (2: uv, 3: xyz)", dims);
    int input_ch = dims * (1 + 2 * L);
    CTX_REQUIRE(uvm_build_plan(D, W, input_ch, output_ch, skip, p, total) == 0,
                "uvmlp_bwd: unsupported D=%d W=%d dims=%d L=%d output_ch=%d", D, W, dims, L, output_ch);
    p.dims = dims;
    CTX_REQUIRE(skip >= 0 && skip + 1 < D, "uvmlp_bwd: skip=%d outside [0, D-2]", skip);
    for (int i = 0; i <= D; ++i) CTX_REQUIRE(gws[i] && gbs[i], "uvmlp_bwd: null gradient pointer for layer %d", i);
    hipStream_t s = (hipStream_t)stream;
    const float *pk = (const float *)packed;
    const float *saved = (const float *)saved_v;
    char *wp = (char *)ws;
    float *dz = (float *)wp;          wp += uvm_align((int64_t)D * N * W * 4);
    float *slab = (float *)wp;        wp += uvm_align((int64_t)UVM_WG_GROUPS * W * (W > 64 ? W : 64) * 4);
    float *bslab = (float *)wp;       wp += uvm_align((int64_t)UVM_WG_GROUPS_EMB * W * 4);
    float *part_w = (float *)wp;      wp += uvm_align((int64_t)UVM_DGRAD_GRID * 4 * W * 4);
    float *part_b = (float *)wp;      wp += uvm_align((int64_t)UVM_DGRAD_GRID * 4 * 4);
    UvmCtl *ctl = (UvmCtl *)wp;

    // ---- phase 1: the dZ chain ----
    int64_t ntiles = cdiv64(N, UVM_TM);
    int dg = (int)(ntiles < UVM_DGRAD_GRID ? ntiles : UVM_DGRAD_GRID);
    size_t lds = (size_t)(UVM_TM * (W + 4) + UVM_TM * 4) * 4;
    bool fast = dims == 2 && W == 256 && p.epad == UVM_EPAD;
    {
        const char *ex = getenv("CTX_UVMLP_EXACT_F32");
        if (ex && ex[0] == '1') fast = false;
        for (int i = 1; i < D && fast; ++i) fast = p.wt16_off[i] >= 0;
    }
    if (fast) {
        // split-fp16 chain: scale from max|grad| on the device, then the same phases as the f32 kernel
        (void)hipMemsetAsync(ctl, 0, sizeof(UvmCtl), s);
        hipLaunchKernelGGL(k_uvm_absmax, dim3(512), dim3(256), 0, s, grad_raw, grad_raw ? N * output_ch : 0, grad_tex, grad_tex ? N * output_ch : 0, ctl);
        hipLaunchKernelGGL(k_uvm_scale, dim3(1), dim3(1), 0, s, ctl);
        const size_t lds16 = (size_t)UVM_TM * UVM16B_STRIDE * sizeof(f16) + UVM_TM * 4 * 4;
        (void)hipFuncSetAttribute((const void *)k_uvmlp_dgrad16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
        hipLaunchKernelGGL(k_uvmlp_dgrad16, dim3(dg), dim3(256), lds16, s, grad_raw, grad_tex, raw, N, pk, p, saved, dz, part_w, part_b, ctl);
    } else if (W == 256) {
        (void)hipFuncSetAttribute((const void *)k_uvmlp_dgrad<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_uvmlp_dgrad<256>, dim3(dg), dim3(256), lds, s, grad_raw, grad_tex, raw, N, pk, p, saved, dz, part_w, part_b);
    } else if (W == 128) {
        hipLaunchKernelGGL(k_uvmlp_dgrad<128>, dim3(dg), dim3(128), lds, s, grad_raw, grad_tex, raw, N, pk, p, saved, dz, part_w, part_b);
    } else {
        hipLaunchKernelGGL(k_uvmlp_dgrad<64>, dim3(dg), dim3(64), lds, s, grad_raw, grad_tex, raw, N, pk, p, saved, dz, part_w, part_b);
    }
    CTX_CHECK_LAUNCH("uvmlp_dgrad");
    hipLaunchKernelGGL(k_uvm_reduce, dim3(cdiv(output_ch * W / 4, 64)), dim3(256), 0, s, part_w, dg, (int64_t)4 * W, output_ch, W, W,
                       gws[D], W, 0);
    hipLaunchKernelGGL(k_uvm_reduce, dim3(1), dim3(256), 0, s, part_b, dg, (int64_t)4, 1, 4, output_ch, gbs[D], 4, 0);
    CTX_CHECK_LAUNCH("uvmlp_reduce_out");

    // ---- phase 2: weight / bias gradients ----
    int64_t chunk = cdiv64(N, UVM_WG_GROUPS);
    chunk = (chunk + 47) / 48 * 48;              // whole rounds of both loop forms (3 buffers x 4 k-steps x 2 texels; 16)
    int G = (int)cdiv64(N, chunk);
    int64_t chunk_e = cdiv64(N, UVM_WG_GROUPS_EMB);
    chunk_e = (chunk_e + 47) / 48 * 48;
    int Ge = (int)cdiv64(N, chunk_e);
    const float *emb = saved;
    const float *acts = saved + N * p.epad;
    static const int wg8 = [] { const char *e = getenv("CTX_UVM_WG8"); return e ? atoi(e) : 0; }();
    for (int li = D - 1; li >= 0; --li) {
        const float *dzl = dz + (int64_t)li * N * W;
        const int kin = li == 0 ? input_ch : (li == skip + 1 ? input_ch + W : W);
        const bool has_emb = li == 0 || li == skip + 1;
        const bool has_hid = li != 0;
        if (has_hid) {
            const float *in = acts + (int64_t)(li - 1) * N * W;
            if (W == 256) {
                if (wg8) uvm_launch_wgrad<2, 4, 4, 2, 256, 256, 256>(G, dzl, in, N, chunk, slab, bslab, s);
                else uvm_launch_wgrad<2, 2, 4, 4, 256, 256, 256>(G, dzl, in, N, chunk, slab, bslab, s);
            } else if (W == 128) uvm_launch_wgrad<2, 2, 2, 2, 128, 128, 128>(G, dzl, in, N, chunk, slab, bslab, s);
            else uvm_launch_wgrad<2, 2, 1, 1, 64, 64, 64>(G, dzl, in, N, chunk, slab, bslab, s);
            CTX_CHECK_LAUNCH("uvmlp_wgrad");
            hipLaunchKernelGGL(k_uvm_reduce, dim3(cdiv(W * W / 4, 64)), dim3(256), 0, s, slab, G, (int64_t)W * W, W, W, W, gws[li], kin,
                               has_emb ? input_ch : 0);
            hipLaunchKernelGGL(k_uvm_reduce, dim3(cdiv(W / 4, 64)), dim3(256), 0, s, bslab, G, (int64_t)W, 1, W, W, gbs[li], W, 0);
        }
        if (has_emb) {
            float *bsl = has_hid ? nullptr : bslab;
            if (p.epad == UVM_EPAD) {
                if (W == 256) uvm_launch_wgrad<4, 1, 2, 2, 256, UVM_EPAD, UVM_EPAD>(Ge, dzl, emb, N, chunk_e, slab, bsl, s);
                else if (W == 128) uvm_launch_wgrad<2, 1, 2, 2, 128, UVM_EPAD, UVM_EPAD>(Ge, dzl, emb, N, chunk_e, slab, bsl, s);
                else uvm_launch_wgrad<1, 1, 2, 2, 64, UVM_EPAD, UVM_EPAD>(Ge, dzl, emb, N, chunk_e, slab, bsl, s);
            } else {
                if (W == 256) uvm_launch_wgrad<4, 1, 2, 2, 256, UVM_EPAD3, UVM_EPAD3>(Ge, dzl, emb, N, chunk_e, slab, bsl, s);
                else if (W == 128) uvm_launch_wgrad<2, 1, 2, 2, 128, UVM_EPAD3, UVM_EPAD3>(Ge, dzl, emb, N, chunk_e, slab, bsl, s);
                else uvm_launch_wgrad<1, 1, 2, 2, 64, UVM_EPAD3, UVM_EPAD3>(Ge, dzl, emb, N, chunk_e, slab, bsl, s);
            }
            CTX_CHECK_LAUNCH("uvmlp_wgrad_emb");
            hipLaunchKernelGGL(k_uvm_reduce, dim3(cdiv(W * 64 / 4, 64)), dim3(256), 0, s, slab, Ge, (int64_t)W * 64, W, 64, input_ch, gws[li], kin, 0);
            if (!has_hid)
                hipLaunchKernelGGL(k_uvm_reduce, dim3(cdiv(W / 4, 64)), dim3(256), 0, s, bslab, Ge, (int64_t)W, 1, W, W, gbs[li], W, 0);
        }
        CTX_CHECK_LAUNCH("uvmlp_wgrad_reduce");
    }
    return CTX_OK;
}
