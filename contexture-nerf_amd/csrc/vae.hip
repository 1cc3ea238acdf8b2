// VAE engine: diffusers `AutoencoderKL.decode` (post_quant_conv + Decoder) as used by
// StableDiffusion.decode_latents (src/stable_diffusion_depth.py:976-990): latents [B,4,h,w] -> image [B,3,8h,8w], and
// `AutoencoderKL.encode` (Encoder + quant_conv) behind StableDiffusion.encode_imgs (:971-975): image [B,3,8h,8w] ->
// moments [B,8,h,w] (mean | logvar of the latent distribution; sampling stays with the caller's RNG).
// Reuses the UNet's fp16 MFMA conv / GEMM and GroupNorm kernels; the single-head (dim 512) mid-block attention is run
// as two GEMMs around a row softmax (scores are materialised: it is one call per painted view, not per denoise step).
// Parameter names follow the diffusers AutoencoderKL state_dict ("decoder.up_blocks.2.resnets.0.conv1.weight", ...).
#include "common.h"
#include "kernels.h"
#include <algorithm>
#include <string>
#include <vector>

struct VParam { std::string name; int ndim; int64_t shape[4]; int kind; size_t dst; int a, b;   // kind: 0 copy, 1 conv3, 2 convin
                // second pack for the encoder's backward (input gradients): kind2 1 = conv3 [Cout,Cin,3,3] -> [Cin][2-ky][2-kx][pad2 >= Cout]
                // (the data-gradient of a 3x3 convolution is a 3x3 convolution with this matrix), 2 = [out,in] -> [in][ld2] at column col2
                int kind2 = 0; size_t dst2 = 0; int pad2 = 0, ld2 = 0, col2 = 0; };
struct VRes { int cin, cout; size_t n1g, n1b, c1w, c1b, n2g, n2b, c2w, c2b, scw, scb; size_t c1wT = 0, c2wT = 0, scwT = 0; };
struct VAttn { size_t ng, nb, qkv, qkvb, ow, ob; size_t qkvT = 0, owT = 0; };

struct ctx_vae {
    ctx_vae_config_t cfg;
    std::vector<VParam> params;
    size_t wtop = 0;
    size_t pqw, pqb, ciw, cib, cng, cnb, cow, cob;
    VAttn att;
    VRes mid[2];
    // encoder
    size_t e_ciw, e_cib, e_cng, e_cnb, e_cow, e_cob, qw, qb;
    VAttn e_att;
    VRes e_mid[2];
    std::vector<std::vector<VRes>> down;
    std::vector<size_t> dnw, dnb;
    std::vector<std::vector<VRes>> up;
    std::vector<size_t> upw, upb;
    std::vector<int> upc;
    f16 *W = nullptr; char *ws = nullptr; size_t ws_cap = 0, top = 0, peak = 0;
    bool dry = false; hipStream_t s = nullptr; int rc = 0;
    double flops = 0;
    int n_dec_params = 0;
    // encoder backward: transposed packs + the tape of the last training forward (pointers into the workspace)
    std::vector<size_t> dnwT;
    size_t e_cowT = 0;
    struct ResTape { const f16 *x, *h; };
    struct Tape {
        bool valid = false; int B = 0, H = 0, W = 0; size_t top = 0;
        const f16 *conv_in_out = nullptr;
        std::vector<ResTape> res;          // in forward order: down[i][j]..., mid0, mid1
        const f16 *attn_in = nullptr, *qkv = nullptr, *norm_out_in = nullptr;
    } tape;
    bool train = false;

    size_t walloc(size_t n) { size_t o = wtop; wtop += (n + 127) / 128 * 128; return o; }
    size_t add(const std::string &name, std::vector<int64_t> shp, int kind, size_t dst, int a = 0, int b = 0)
    {
        VParam p; p.name = name; p.ndim = (int)shp.size(); p.kind = kind; p.dst = dst; p.a = a; p.b = b;
        for (int i = 0; i < 4; ++i) p.shape[i] = i < p.ndim ? shp[i] : 1;
        params.push_back(p);
        return dst;
    }
    size_t vec(const std::string &n, int c) { return add(n, {c}, 0, walloc(c)); }
    void *alloc(size_t bytes)
    {
        size_t o = (top + 255) / 256 * 256;
        top = o + bytes;
        if (top > peak) peak = top;
        if (!dry && top > ws_cap) { rc = CTX_E_STATE; ctx_set_error("vae: workspace too small (%zu > %zu)", top, ws_cap); return ws; }
        return dry ? nullptr : (void *)(ws + o);
    }
    f16 *allocH(size_t n) { return (f16 *)alloc(n * 2); }
};

static void vadd_bwd_conv3(ctx_vae *v, size_t &dstT, int cout, int cin, int pad)
{
    dstT = v->walloc((size_t)cin * 9 * pad);
    VParam &q = v->params.back();
    q.kind2 = 1; q.dst2 = dstT; q.pad2 = pad;
}
static void vadd_bwd_mat(ctx_vae *v, size_t dstT, int ld, int col)
{
    VParam &q = v->params.back();
    q.kind2 = 2; q.dst2 = dstT; q.ld2 = ld; q.col2 = col;
}

static void vadd_res(ctx_vae *v, const std::string &p, int cin, int cout, VRes &r, bool bwd = false)
{
    r.cin = cin; r.cout = cout;
    r.n1g = v->vec(p + ".norm1.weight", cin); r.n1b = v->vec(p + ".norm1.bias", cin);
    r.c1w = v->add(p + ".conv1.weight", {cout, cin, 3, 3}, 1, v->walloc((size_t)cout * cin * 9), cout, cin);
    if (bwd) vadd_bwd_conv3(v, r.c1wT, cout, cin, cout);
    r.c1b = v->vec(p + ".conv1.bias", cout);
    r.n2g = v->vec(p + ".norm2.weight", cout); r.n2b = v->vec(p + ".norm2.bias", cout);
    r.c2w = v->add(p + ".conv2.weight", {cout, cout, 3, 3}, 1, v->walloc((size_t)cout * cout * 9), cout, cout);
    if (bwd) vadd_bwd_conv3(v, r.c2wT, cout, cout, cout);
    r.c2b = v->vec(p + ".conv2.bias", cout);
    if (cin != cout) {
        r.scw = v->add(p + ".conv_shortcut.weight", {cout, cin, 1, 1}, 0, v->walloc((size_t)cout * cin));
        if (bwd) { r.scwT = v->walloc((size_t)cin * cout); vadd_bwd_mat(v, r.scwT, cout, 0); }
        r.scb = v->vec(p + ".conv_shortcut.bias", cout);
    } else r.scw = r.scb = 0;
}

static void vadd_attn(ctx_vae *v, const std::string &ap, int top, VAttn &a, bool bwd = false)
{
    a.ng = v->vec(ap + ".group_norm.weight", top); a.nb = v->vec(ap + ".group_norm.bias", top);
    a.qkv = v->walloc((size_t)3 * top * top); a.qkvb = v->walloc((size_t)3 * top);
    if (bwd) a.qkvT = v->walloc((size_t)3 * top * top);           // [top (in)][3 top (q | k | v out)]
    const char *qkv[3] = {"to_q", "to_k", "to_v"};
    for (int k = 0; k < 3; ++k) {
        v->add(ap + "." + qkv[k] + ".weight", {top, top}, 0, a.qkv + (size_t)k * top * top);
        if (bwd) vadd_bwd_mat(v, a.qkvT, 3 * top, k * top);
        v->add(ap + "." + qkv[k] + ".bias", {top}, 0, a.qkvb + (size_t)k * top);
    }
    a.ow = v->add(ap + ".to_out.0.weight", {top, top}, 0, v->walloc((size_t)top * top));
    if (bwd) { a.owT = v->walloc((size_t)top * top); vadd_bwd_mat(v, a.owT, top, 0); }
    a.ob = v->vec(ap + ".to_out.0.bias", top);
}

extern "C" ctx_vae_t *ctx_vae_create(const ctx_vae_config_t *cfg)
{
    if (!cfg || cfg->n_levels < 1 || cfg->n_levels > 4 || cfg->latent_channels > 8 || cfg->out_channels > 4 || cfg->groups > 64 ||
        cfg->layers_per_block < 1 || cfg->layers_per_block > 3) { ctx_set_error("vae_create: unsupported config"); return nullptr; }
    for (int i = 0; i < cfg->n_levels; ++i)
        if (cfg->block_out_channels[i] % 64 || cfg->block_out_channels[i] % cfg->groups) { ctx_set_error("vae_create: channels must be multiples of 64 and of groups"); return nullptr; }
    ctx_vae *v = new ctx_vae();
    v->cfg = *cfg;
    const int n = cfg->n_levels, L = cfg->latent_channels;
    const int *ch = cfg->block_out_channels;
    const int top = ch[n - 1];
    if (top % 64) { delete v; return nullptr; }
    v->pqw = v->add("post_quant_conv.weight", {L, L, 1, 1}, 0, v->walloc((size_t)L * L));
    v->pqb = v->vec("post_quant_conv.bias", L);
    v->ciw = v->add("decoder.conv_in.weight", {top, L, 3, 3}, 2, v->walloc((size_t)top * 72), top, L);
    v->cib = v->vec("decoder.conv_in.bias", top);
    vadd_res(v, "decoder.mid_block.resnets.0", top, top, v->mid[0]);
    vadd_attn(v, "decoder.mid_block.attentions.0", top, v->att);
    vadd_res(v, "decoder.mid_block.resnets.1", top, top, v->mid[1]);
    v->up.resize(n); v->upw.assign(n, 0); v->upb.assign(n, 0); v->upc.assign(n, 0);
    int out = top;
    for (int i = 0; i < n; ++i) {
        int prev = out; out = ch[n - 1 - i];
        std::string p = "decoder.up_blocks." + std::to_string(i);
        v->up[i].resize(cfg->layers_per_block + 1);
        for (int j = 0; j <= cfg->layers_per_block; ++j) vadd_res(v, p + ".resnets." + std::to_string(j), j == 0 ? prev : out, out, v->up[i][j]);
        if (i != n - 1) {
            v->upc[i] = out;
            v->upw[i] = v->add(p + ".upsamplers.0.conv.weight", {out, out, 3, 3}, 1, v->walloc((size_t)out * out * 9), out, out);
            v->upb[i] = v->vec(p + ".upsamplers.0.conv.bias", out);
        }
    }
    v->cng = v->vec("decoder.conv_norm_out.weight", ch[0]); v->cnb = v->vec("decoder.conv_norm_out.bias", ch[0]);
    v->cow = v->add("decoder.conv_out.weight", {cfg->out_channels, ch[0], 3, 3}, 1, v->walloc((size_t)cfg->out_channels * ch[0] * 9), cfg->out_channels, ch[0]);
    v->cob = v->vec("decoder.conv_out.bias", cfg->out_channels);
    // ---- encoder (diffusers Encoder: conv_in, DownEncoderBlock2D x n, UNetMidBlock2D, GroupNorm-SiLU-conv_out) + quant_conv.
    // Registered after the decoder so a decoder-only checkpoint still fills a prefix of the table.
    v->n_dec_params = (int)v->params.size();
    v->e_ciw = v->add("encoder.conv_in.weight", {ch[0], cfg->out_channels, 3, 3}, 2, v->walloc((size_t)ch[0] * 72), ch[0], cfg->out_channels);
    v->e_cib = v->vec("encoder.conv_in.bias", ch[0]);
    v->down.resize(n); v->dnw.assign(n, 0); v->dnb.assign(n, 0);
    int cur = ch[0];
    for (int i = 0; i < n; ++i) {
        std::string p = "encoder.down_blocks." + std::to_string(i);
        v->down[i].resize(cfg->layers_per_block);
        for (int j = 0; j < cfg->layers_per_block; ++j) { vadd_res(v, p + ".resnets." + std::to_string(j), j == 0 ? cur : ch[i], ch[i], v->down[i][j], true); }
        cur = ch[i];
        if (i != n - 1) {
            v->dnw[i] = v->add(p + ".downsamplers.0.conv.weight", {cur, cur, 3, 3}, 1, v->walloc((size_t)cur * cur * 9), cur, cur);
            v->dnwT.resize(n, 0);
            vadd_bwd_conv3(v, v->dnwT[i], cur, cur, cur);
            v->dnb[i] = v->vec(p + ".downsamplers.0.conv.bias", cur);
        }
    }
    vadd_res(v, "encoder.mid_block.resnets.0", top, top, v->e_mid[0], true);
    vadd_attn(v, "encoder.mid_block.attentions.0", top, v->e_att, true);
    vadd_res(v, "encoder.mid_block.resnets.1", top, top, v->e_mid[1], true);
    v->e_cng = v->vec("encoder.conv_norm_out.weight", top); v->e_cnb = v->vec("encoder.conv_norm_out.bias", top);
    v->e_cow = v->add("encoder.conv_out.weight", {2 * L, top, 3, 3}, 1, v->walloc((size_t)2 * L * top * 9), 2 * L, top);
    vadd_bwd_conv3(v, v->e_cowT, 2 * L, top, 64);                 // its 2L output channels padded to one 64-deep K stage
    v->e_cob = v->vec("encoder.conv_out.bias", 2 * L);
    v->qw = v->add("quant_conv.weight", {2 * L, 2 * L, 1, 1}, 0, v->walloc((size_t)4 * L * L));
    v->qb = v->vec("quant_conv.bias", 2 * L);
    return v;
}

extern "C" void ctx_vae_destroy(ctx_vae_t *v) { delete v; }
extern "C" int32_t ctx_vae_param_count(const ctx_vae_t *v) { return v ? (int32_t)v->params.size() : 0; }
/* the first ctx_vae_decoder_param_count entries are post_quant_conv + decoder, the rest encoder + quant_conv */
extern "C" int32_t ctx_vae_decoder_param_count(const ctx_vae_t *v) { return v ? v->n_dec_params : 0; }
extern "C" const char *ctx_vae_param_name(const ctx_vae_t *v, int32_t i) { return (v && i >= 0 && i < (int)v->params.size()) ? v->params[i].name.c_str() : ""; }
extern "C" int32_t ctx_vae_param_shape(const ctx_vae_t *v, int32_t i, int64_t shape4[4])
{
    if (!v || i < 0 || i >= (int)v->params.size()) return 0;
    for (int k = 0; k < 4; ++k) shape4[k] = v->params[i].shape[k];
    return v->params[i].ndim;
}
extern "C" int64_t ctx_vae_weight_bytes(const ctx_vae_t *v) { return v ? (int64_t)v->wtop * 2 + 256 : 0; }
extern "C" int32_t ctx_vae_bind(ctx_vae_t *v, void *weights, void *workspace, int64_t workspace_bytes)
{
    CTX_REQUIRE(v && weights && workspace && workspace_bytes > 0, "vae_bind: bad args");
    v->W = (f16 *)weights; v->ws = (char *)workspace; v->ws_cap = (size_t)workspace_bytes;
    return CTX_OK;
}

__global__ void k_vpack_copy(const float *__restrict__ s, int64_t n, f16 *__restrict__ d)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d[i] = (f16)s[i];
}
__global__ void k_vpack_conv3(const float *__restrict__ s, int Cout, int Cin, int Cinp, f16 *__restrict__ d)
{
    int64_t n = (int64_t)Cout * 9 * Cinp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int c = (int)(i % Cinp), tap = (int)((i / Cinp) % 9), o = (int)(i / ((int64_t)Cinp * 9));
        d[i] = c < Cin ? (f16)s[((int64_t)o * Cin + c) * 9 + tap] : (f16)0.f;
    }
}
// backward packs: conv3 [Cout,Cin,3,3] -> [Cin][t' = 3 (2-ky) + (2-kx)][pad] (zero beyond Cout); matrix [out,in] -> [in][ld] at column col
__global__ void k_vpack_conv3_T(const float *__restrict__ s, int Cout, int Cin, int pad, f16 *__restrict__ d)
{
    int64_t n = (int64_t)Cin * 9 * pad;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int o = (int)(i % pad), tp = (int)((i / pad) % 9), c = (int)(i / ((int64_t)pad * 9));
        int ky = 2 - tp / 3, kx = 2 - tp % 3;
        d[i] = o < Cout ? (f16)s[(((int64_t)o * Cin + c) * 3 + ky) * 3 + kx] : (f16)0.f;
    }
}
__global__ void k_vpack_mat_T(const float *__restrict__ s, int out, int in, int ld, int col, f16 *__restrict__ d)
{
    int64_t n = (int64_t)out * in;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int c = (int)(i % in), o = (int)(i / in);
        d[(int64_t)c * ld + col + o] = (f16)s[i];
    }
}

extern "C" int32_t ctx_vae_set_param(ctx_vae_t *v, int32_t i, const float *src, ctx_stream_t stream)
{
    CTX_REQUIRE(v && v->W && src && i >= 0 && i < (int)v->params.size(), "vae_set_param: bad args / not bound");
    const VParam &p = v->params[i];
    int64_t n = 1;
    for (int k = 0; k < p.ndim; ++k) n *= p.shape[k];
    unsigned nb = (unsigned)(cdiv64(n, 256) > 4096 ? 4096 : cdiv64(n, 256));
    hipStream_t s = (hipStream_t)stream;
    if (p.kind == 0) hipLaunchKernelGGL(k_vpack_copy, dim3(nb), dim3(256), 0, s, src, n, v->W + p.dst);
    else hipLaunchKernelGGL(k_vpack_conv3, dim3(nb), dim3(256), 0, s, src, p.a, p.b, p.kind == 2 ? 8 : p.b, v->W + p.dst);
    if (p.kind2 == 1) hipLaunchKernelGGL(k_vpack_conv3_T, dim3(nb), dim3(256), 0, s, src, p.a, p.b, p.pad2, v->W + p.dst2);
    else if (p.kind2 == 2) hipLaunchKernelGGL(k_vpack_mat_T, dim3(nb), dim3(256), 0, s, src, (int)p.shape[0], (int)p.shape[1], p.ld2, p.col2, v->W + p.dst2);
    CTX_CHECK_LAUNCH("vae_set_param");
    return CTX_OK;
}

// post_quant_conv: 1x1 conv over <= 8 latent channels, f32 NCHW in/out (16 MACs per pixel)
__global__ __launch_bounds__(256) void k_pointwise_small(const float *__restrict__ x, const f16 *__restrict__ w, const f16 *__restrict__ b,
                                                         int B, int C, int64_t HW, float *__restrict__ y)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)B * HW; i += (int64_t)gridDim.x * 256) {
        int bb = (int)(i / HW); int64_t p = i % HW;
        float in[8];
        for (int c = 0; c < C; ++c) in[c] = x[((int64_t)bb * C + c) * HW + p];
        for (int o = 0; o < C; ++o) {
            float acc = (float)b[o];
            for (int c = 0; c < C; ++c) acc += in[c] * (float)w[o * C + c];
            y[((int64_t)bb * C + o) * HW + p] = acc;
        }
    }
}

// row softmax of f16 scores [rows, n] * scale -> f16 probabilities (fp32 math), one workgroup per row
__global__ __launch_bounds__(256) void k_softmax_rows(const f16 *__restrict__ s, int n, float scale_log2e, f16 *__restrict__ p)
{
    const f16 *row = s + (size_t)blockIdx.x * n;
    f16 *out = p + (size_t)blockIdx.x * n;
    __shared__ float red[4];
    float mx = -INFINITY;
    for (int i = threadIdx.x * 8; i < n; i += 2048) {
        f16x8 v = *(const f16x8 *)(row + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) mx = fmaxf(mx, (float)v[j]);
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale_log2e;
    __syncthreads();
    float sum = 0.f;
    for (int i = threadIdx.x * 8; i < n; i += 2048) {
        f16x8 v = *(const f16x8 *)(row + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += __builtin_amdgcn_exp2f((float)v[j] * scale_log2e - mx);
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
    for (int i = threadIdx.x * 8; i < n; i += 2048) {
        f16x8 v = *(const f16x8 *)(row + i);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)(__builtin_amdgcn_exp2f((float)v[j] * scale_log2e - mx) * inv);
        *(f16x8 *)(out + i) = o;
    }
}

#define VRUN(expr) do { if (!v->dry && v->rc == 0) { int r__ = (expr); if (r__ != 0) v->rc = r__; } } while (0)

static void vgemm(ctx_vae *v, const f16 *X, const f16 *Wt, const f16 *bias, const f16 *res, int M, int N, int K, f16 *out)
{
    GemmArgs a = {};
    a.X = X; a.Wt = Wt; a.bias = bias; a.residual = res; a.out = out; a.M = M; a.N = N; a.K = K; a.ldc = N; a.ldr = N;
    a.rows_per_batch = 1; a.ldrb = N; a.epi = 0;
    v->flops += 2.0 * M * N * K;
    size_t mark = v->top;
    ctx_gemm_plan(a, false);
    if (a.splitk > 1) a.part = (float *)v->alloc((size_t)a.splitk * M * N * 4);
    VRUN(ctx_gemm_dispatch(a, false, v->s));
    v->top = mark;
}
static void vconv(ctx_vae *v, const f16 *x, size_t w, size_t bias, const f16 *res, int B, int H, int W, int Cin, int Cout, int ups, f16 *out,
                  int down = 0)
{
    GemmArgs a = {};
    a.Ho = down ? H / 2 : H << ups; a.Wo = down ? W / 2 : W << ups;
    a.X = x; a.Wt = v->W + w; a.bias = v->W + bias; a.residual = res; a.out = out;
    a.M = B * a.Ho * a.Wo; a.N = Cout; a.K = 9 * Cin; a.ldc = Cout; a.ldr = Cout; a.rows_per_batch = a.Ho * a.Wo; a.ldrb = Cout;
    a.H = H; a.W = W; a.Cin = Cin; a.stride = down ? 2 : 1; a.ups = ups; a.poff = down ? 1 : 0;
    v->flops += 2.0 * a.M * a.N * a.K;
    size_t mark = v->top;
    ctx_gemm_plan(a, true);
    if (a.splitk > 1) a.part = (float *)v->alloc((size_t)a.splitk * a.M * a.N * 4);
    VRUN(ctx_gemm_dispatch(a, true, v->s));
    v->top = mark;
}
static void vgn(ctx_vae *v, const f16 *x, size_t g, size_t b, int B, int HW, int C, int silu, f16 *y, void *stats)
{
    VRUN(ctx_groupnorm_f16(x, v->W + g, v->W + b, B, HW, C, v->cfg.groups, 1e-6f, silu, y, stats, v->s));
}
static void vres(ctx_vae *v, const VRes &r, const f16 *x, int B, int H, int W, f16 *out, void *stats)
{
    const size_t M = (size_t)B * H * W;
    f16 *hkeep = nullptr;
    if (v->train) {                                  // training forward: the two GroupNorm inputs (x, h) stay on the tape
        hkeep = v->allocH(M * r.cout);
        v->tape.res.push_back({x, hkeep});
    }
    size_t mark = v->top;
    f16 *t1 = v->allocH(M * r.cin);
    vgn(v, x, r.n1g, r.n1b, B, H * W, r.cin, 1, t1, stats);
    f16 *h = hkeep ? hkeep : v->allocH(M * r.cout);
    vconv(v, t1, r.c1w, r.c1b, nullptr, B, H, W, r.cin, r.cout, 0, h);
    f16 *t2 = v->allocH(M * r.cout);
    vgn(v, h, r.n2g, r.n2b, B, H * W, r.cout, 1, t2, stats);
    const f16 *sc = x;
    if (r.cin != r.cout) {
        f16 *s2 = v->allocH(M * r.cout);
        vgemm(v, x, v->W + r.scw, v->W + r.scb, nullptr, (int)M, r.cout, r.cin, s2);
        sc = s2;
    }
    vconv(v, t2, r.c2w, r.c2b, sc, B, H, W, r.cout, r.cout, 0, out);
    v->top = mark;
}

// single-head attention of the mid block, dim = top: q,k,v GEMM -> per-batch scores GEMM -> softmax -> P.V GEMM -> out proj
// (+residual o); the result lands in x.
static int vattn(ctx_vae *v, const VAttn &at, const f16 *o, f16 *x, int B, int h, int w, int top, void *stats)
{
        const int S = h * w, M = B * S;
        if (S % 64) { ctx_set_error("vae: latent h*w must be a multiple of 64 (got %d)", S); return CTX_E_ARG; }
        f16 *qkeep = nullptr;
        if (v->train) { qkeep = v->allocH((size_t)M * 3 * top); v->tape.attn_in = o; v->tape.qkv = qkeep; }
        size_t mark = v->top;
        f16 *g = v->allocH((size_t)M * top);
        vgn(v, o, at.ng, at.nb, B, S, top, 0, g, stats);
        f16 *qkv = qkeep ? qkeep : v->allocH((size_t)M * 3 * top);
        vgemm(v, g, v->W + at.qkv, v->W + at.qkvb, nullptr, M, 3 * top, top, qkv);
        f16 *att = v->allocH((size_t)M * top);
        f16 *sc = v->allocH((size_t)S * S), *pr = v->allocH((size_t)S * S), *vt = v->allocH((size_t)top * S);
        f16 *qb = v->allocH((size_t)S * top), *kb = v->allocH((size_t)S * top);
        for (int b = 0; b < B; ++b) {
            const f16 *base = qkv ? qkv + (size_t)b * S * 3 * top : nullptr;
            // de-interleave q and k rows (row stride 3*top) into dense [S, top] operands; V^T through the head-transpose kernel
            if (!v->dry) {
                (void)hipMemcpy2DAsync(qb, (size_t)top * 2, base, (size_t)3 * top * 2, (size_t)top * 2, S, hipMemcpyDeviceToDevice, v->s);
                (void)hipMemcpy2DAsync(kb, (size_t)top * 2, base + top, (size_t)3 * top * 2, (size_t)top * 2, S, hipMemcpyDeviceToDevice, v->s);
            }
            VRUN(ctx_transpose_v_f16(base + 2 * top, 1, S, 3 * top, top / 64, S, 0, vt, v->s));
            vgemm(v, qb, kb, nullptr, nullptr, S, S, top, sc);
            if (!v->dry) hipLaunchKernelGGL(k_softmax_rows, dim3(S), dim3(256), 0, v->s, sc, S, 1.4426950408889634f / sqrtf((float)top), pr);
            vgemm(v, pr, vt, nullptr, nullptr, S, top, S, att ? att + (size_t)b * S * top : nullptr);
        }
        f16 *o2 = v->allocH((size_t)M * top);
        vgemm(v, att, v->W + at.ow, v->W + at.ob, o, M, top, top, o2);
        // o2 lives above the mark: copy down into x (free since its previous content is dead)
        if (!v->dry) (void)hipMemcpyAsync(x, o2, (size_t)M * top * 2, hipMemcpyDeviceToDevice, v->s);
        v->top = mark;
        return 0;
}

static int vae_run(ctx_vae *v, const float *z, int B, int H, int W, float *img)
{
    const ctx_vae_config_t &c = v->cfg;
    const int n = c.n_levels, top = c.block_out_channels[n - 1], L = c.latent_channels;
    v->top = 0; v->peak = 0; v->rc = 0; v->flops = 0;
    void *stats = v->alloc((size_t)ctx_groupnorm_ws_bytes(B, c.groups));
    float *zq = (float *)v->alloc((size_t)B * L * H * W * 4);
    if (!v->dry) hipLaunchKernelGGL(k_pointwise_small, dim3((unsigned)cdiv64((int64_t)B * H * W, 256)), dim3(256), 0, v->s, z, v->W + v->pqw,
                                    v->W + v->pqb, B, L, (int64_t)H * W, zq);
    int h = H, w = W;
    f16 *x = v->allocH((size_t)B * h * w * top);
    VRUN(ctx_conv_in_f16(zq, v->W + v->ciw, v->W + v->cib, B, L, h, w, top, x, v->s));
    f16 *o = v->allocH((size_t)B * h * w * top);
    vres(v, v->mid[0], x, B, h, w, o, stats);
    { int r = vattn(v, v->att, o, x, B, h, w, top, stats); if (r) return r; }
    vres(v, v->mid[1], x, B, h, w, o, stats);
    f16 *cur = o;
    int cc = top;
    for (int i = 0; i < n; ++i) {
        for (size_t j = 0; j < v->up[i].size(); ++j) {
            int cout = v->up[i][j].cout;
            f16 *nx = v->allocH((size_t)B * h * w * cout);
            vres(v, v->up[i][j], cur, B, h, w, nx, stats);
            cur = nx; cc = cout;
        }
        if (i != n - 1) {
            f16 *nx = v->allocH((size_t)B * (2 * h) * (2 * w) * cc);
            vconv(v, cur, v->upw[i], v->upb[i], nullptr, B, h, w, cc, cc, 1, nx);
            cur = nx; h *= 2; w *= 2;
        }
    }
    f16 *y = v->allocH((size_t)B * h * w * cc);
    vgn(v, cur, v->cng, v->cnb, B, h * w, cc, 1, y, stats);
    VRUN(ctx_conv_out_f16(y, v->W + v->cow, v->W + v->cob, B, h, w, cc, c.out_channels, img, v->s));
    if (!v->dry && v->rc == 0) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { ctx_set_error("vae_decode: launch failed: %s", hipGetErrorString(e)); return CTX_E_LAUNCH; }
    }
    return v->rc;
}

extern "C" int64_t ctx_vae_workspace_bytes(const ctx_vae_t *cv, int32_t B, int32_t H, int32_t W)
{
    ctx_vae *v = const_cast<ctx_vae *>(cv);
    if (!v || B < 1 || H < 1 || W < 1 || (H * W) % 64) return -1;
    v->dry = true; v->train = false;
    vae_run(v, nullptr, B, H, W, nullptr);
    v->dry = false;
    return (int64_t)v->peak + 4096;
}

extern "C" int32_t ctx_vae_decode(ctx_vae_t *v, const float *latents, int32_t B, int32_t H, int32_t W, float *image, ctx_stream_t stream)
{
    CTX_REQUIRE(v && latents && image && v->W && v->ws, "vae_decode: null pointer / not bound");
    CTX_REQUIRE(B >= 1 && H >= 1 && W >= 1 && (H * W) % 64 == 0, "vae_decode: need h*w %% 64 == 0 (B=%d H=%d W=%d)", B, H, W);
    v->s = (hipStream_t)stream; v->dry = false; v->train = false;
    return vae_run(v, latents, B, H, W, image);
}

// quant_conv (1x1 over the 2L moment channels) on the encoder's f16 NHWC output -> f32 NCHW moments
__global__ __launch_bounds__(256) void k_quant_moments(const f16 *__restrict__ x, const f16 *__restrict__ w, const f16 *__restrict__ b,
                                                       int B, int C, int64_t HW, float *__restrict__ y)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)B * HW; i += (int64_t)gridDim.x * 256) {
        int bb = (int)(i / HW); int64_t p = i % HW;
        float in[16];
        for (int c = 0; c < C; ++c) in[c] = (float)x[i * C + c];
        for (int o = 0; o < C; ++o) {
            float acc = (float)b[o];
            for (int c = 0; c < C; ++c) acc += in[c] * (float)w[o * C + c];
            y[((int64_t)bb * C + o) * HW + p] = acc;
        }
    }
}

// image f32 NCHW [B,3,H,W] (H, W multiples of 2^(n-1)) -> moments f32 NCHW [B,2L,H>>(n-1),W>>(n-1)]
static int vae_encode_run(ctx_vae *v, const float *img, int B, int H, int W, float *moments)
{
    const ctx_vae_config_t &c = v->cfg;
    const int n = c.n_levels, top = c.block_out_channels[n - 1], L2 = 2 * c.latent_channels;
    v->top = 0; v->peak = 0; v->rc = 0; v->flops = 0;
    v->tape = ctx_vae::Tape();
    void *stats = v->alloc((size_t)ctx_groupnorm_ws_bytes(B, c.groups));
    int h = H, w = W, cc = c.block_out_channels[0];
    f16 *cur = v->allocH((size_t)B * h * w * cc);
    VRUN(ctx_conv_in_f16(img, v->W + v->e_ciw, v->W + v->e_cib, B, c.out_channels, h, w, cc, cur, v->s));
    v->tape.conv_in_out = cur;
    for (int i = 0; i < n; ++i) {
        for (size_t j = 0; j < v->down[i].size(); ++j) {
            int cout = v->down[i][j].cout;
            f16 *nx = v->allocH((size_t)B * h * w * cout);
            vres(v, v->down[i][j], cur, B, h, w, nx, stats);
            cur = nx; cc = cout;
        }
        if (i != n - 1) {
            f16 *nx = v->allocH((size_t)B * (h / 2) * (w / 2) * cc);
            vconv(v, cur, v->dnw[i], v->dnb[i], nullptr, B, h, w, cc, cc, 0, nx, 1);
            cur = nx; h /= 2; w /= 2;
        }
    }
    f16 *o = v->allocH((size_t)B * h * w * top), *x = v->allocH((size_t)B * h * w * top);
    vres(v, v->e_mid[0], cur, B, h, w, o, stats);
    { int r = vattn(v, v->e_att, o, x, B, h, w, top, stats); if (r) return r; }
    if (v->train) o = v->allocH((size_t)B * h * w * top);        // the attention's input stays on the tape: do not overwrite it
    vres(v, v->e_mid[1], x, B, h, w, o, stats);
    v->tape.norm_out_in = o;
    if (v->train) { v->tape.valid = !v->dry; v->tape.B = B; v->tape.H = H; v->tape.W = W; v->tape.top = v->top; }
    f16 *y = v->allocH((size_t)B * h * w * top);
    vgn(v, o, v->e_cng, v->e_cnb, B, h * w, top, 1, y, stats);
    f16 *m16 = v->allocH((size_t)B * h * w * L2);
    vconv(v, y, v->e_cow, v->e_cob, nullptr, B, h, w, top, L2, 0, m16);
    if (!v->dry && v->rc == 0)
        hipLaunchKernelGGL(k_quant_moments, dim3((unsigned)cdiv64((int64_t)B * h * w, 256)), dim3(256), 0, v->s, m16, v->W + v->qw, v->W + v->qb, B,
                           L2, (int64_t)h * w, moments);
    if (!v->dry && v->rc == 0) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { ctx_set_error("vae_encode: launch failed: %s", hipGetErrorString(e)); return CTX_E_LAUNCH; }
    }
    return v->rc;
}

static bool vae_encode_dims_ok(const ctx_vae *v, int B, int H, int W)
{
    const int f = 1 << (v->cfg.n_levels - 1);
    return B >= 1 && H >= f && W >= f && H % f == 0 && W % f == 0 && ((H / f) * (W / f)) % 64 == 0 && v->cfg.latent_channels * 2 % 8 == 0;
}

extern "C" int64_t ctx_vae_encode_workspace_bytes(const ctx_vae_t *cv, int32_t B, int32_t H, int32_t W)
{
    ctx_vae *v = const_cast<ctx_vae *>(cv);
    if (!v || !vae_encode_dims_ok(v, B, H, W)) return -1;
    v->dry = true; v->train = false;
    vae_encode_run(v, nullptr, B, H, W, nullptr);
    v->dry = false;
    return (int64_t)v->peak + 4096;
}

extern "C" int32_t ctx_vae_encode(ctx_vae_t *v, const float *image, int32_t B, int32_t H, int32_t W, float *moments, ctx_stream_t stream)
{
    CTX_REQUIRE(v && image && moments && v->W && v->ws, "vae_encode: null pointer / not bound");
    CTX_REQUIRE(vae_encode_dims_ok(v, B, H, W), "vae_encode: need H, W multiples of %d with (H/%d)*(W/%d) %% 64 == 0 and 2*latent_channels %% 8 == 0 (B=%d H=%d W=%d)",
                1 << (v->cfg.n_levels - 1), 1 << (v->cfg.n_levels - 1), 1 << (v->cfg.n_levels - 1), B, H, W);
    v->s = (hipStream_t)stream; v->dry = false; v->train = false;
    return vae_encode_run(v, image, B, H, W, moments);
}

// =====================================================================================================================
// Encoder backward (input gradients only: the VAE is frozen): the link that lets the reference's SDS loss reach the texture
// (`loss.backward()` through `vae.encode(rendered_grid)`, src/training/trainer.py:732, 866).  Every layer's data gradient runs on
// the forward's kernels: conv dgrad = the implicit-GEMM conv on the transposed / flipped weight pack (stride-2 downsamplers: the
// zero-inserted grid, GemmArgs.zins), linear dgrad = the GEMM on W^T, attention = the five products of softmax attention's
// backward as GEMMs around a row kernel (P is recomputed from the taped q, k).  GroupNorm(+SiLU) backward is two reductions
// (statistics of x, then sum(du) and sum(du x^)) and one apply pass, all deterministic (fixed-order partial sums).
// Gradients are fp16 with a caller-chosen scale `gscale` (every op is linear in the gradient; the result is divided by it).
#define GNB_MAX_SPLITS 128

// MODE 0: per-group partial (sum x, sum x^2); MODE 1: partial (sum du, sum du x^) with du = dy silu'(u) gamma, u = gamma x^ + beta
template <int MODE>
__global__ __launch_bounds__(256) void k_gnb_reduce(const f16 *__restrict__ x, const f16 *__restrict__ dy, const f16 *__restrict__ gamma,
                                                    const f16 *__restrict__ beta, const float *__restrict__ mr, int HW, int C, int G, int NS,
                                                    int silu, float *__restrict__ part)
{
    extern __shared__ float sm[];                    // [PL][C][2] then [C][2]
    const int c8n = C / 8, PL = 256 / c8n;
    const int b = blockIdx.y, sp = blockIdx.x;
    const int c8 = threadIdx.x % c8n, pl = threadIdx.x / c8n;
    const int per = (HW + NS - 1) / NS, p0 = sp * per, p1 = min(HW, p0 + per), cg = C / G;
    float s[8], q[8], a[8], b0[8], ga[8], mu[8], rs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s[j] = 0.f; q[j] = 0.f;
        if (MODE == 1) {
            const int c = c8 * 8 + j, g = c / cg;
            mu[j] = mr[((size_t)b * G + g) * 2]; rs[j] = mr[((size_t)b * G + g) * 2 + 1];
            ga[j] = (float)gamma[c]; a[j] = rs[j] * ga[j]; b0[j] = (float)beta[c] - mu[j] * a[j];
        }
    }
    if (pl < PL)
        for (int p = p0 + pl; p < p1; p += PL) {
            const size_t off = ((size_t)b * HW + p) * C + c8 * 8;
            const f16x8 xv = *(const f16x8 *)(x + off);
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { float f = (float)xv[j]; s[j] += f; q[j] += f * f; }
            } else {
                const f16x8 dv = *(const f16x8 *)(dy + off);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xf = (float)xv[j], u = xf * a[j] + b0[j];
                    float d = (float)dv[j];
                    if (silu) { const float sg = 1.0f / (1.0f + __expf(-u)); d *= sg * (1.0f + u * (1.0f - sg)); }
                    const float du = d * ga[j];
                    s[j] += du; q[j] += du * ((xf - mu[j]) * rs[j]);
                }
            }
        }
    if (pl < PL) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { sm[((size_t)pl * C + c8 * 8 + j) * 2] = s[j]; sm[((size_t)pl * C + c8 * 8 + j) * 2 + 1] = q[j]; }
    }
    __syncthreads();
    float *ch = sm + (size_t)PL * C * 2;
    for (int c = threadIdx.x; c < C; c += 256) {
        float ss = 0.f, qq = 0.f;
        for (int l = 0; l < PL; ++l) { ss += sm[((size_t)l * C + c) * 2]; qq += sm[((size_t)l * C + c) * 2 + 1]; }
        ch[c * 2] = ss; ch[c * 2 + 1] = qq;
    }
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += 256) {
        float ss = 0.f, qq = 0.f;
        for (int c = g * cg; c < (g + 1) * cg; ++c) { ss += ch[c * 2]; qq += ch[c * 2 + 1]; }
        part[(((size_t)b * NS + sp) * G + g) * 2] = ss; part[(((size_t)b * NS + sp) * G + g) * 2 + 1] = qq;
    }
}
// MODE 0 -> (mean, rstd); MODE 1 -> (sum du / n, sum du x^ / n)
template <int MODE>
__global__ void k_gnb_finalize(const float *__restrict__ part, int G, int NS, float n, float eps, float *__restrict__ out)
{
    const int b = blockIdx.x;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        float ss = 0.f, qq = 0.f;
        for (int k = 0; k < NS; ++k) { ss += part[(((size_t)b * NS + k) * G + g) * 2]; qq += part[(((size_t)b * NS + k) * G + g) * 2 + 1]; }
        if (MODE == 0) {
            const float mean = ss / n;
            out[((size_t)b * G + g) * 2] = mean; out[((size_t)b * G + g) * 2 + 1] = rsqrtf(fmaxf(qq / n - mean * mean, 0.f) + eps);
        } else { out[((size_t)b * G + g) * 2] = ss / n; out[((size_t)b * G + g) * 2 + 1] = qq / n; }
    }
}
// dx = rstd (du - c1 - x^ c2) (+ add)
__global__ __launch_bounds__(256) void k_gnb_apply(const f16 *__restrict__ x, const f16 *__restrict__ dy, const f16 *__restrict__ gamma,
                                                   const f16 *__restrict__ beta, const float *__restrict__ mr, const float *__restrict__ cc,
                                                   const f16 *__restrict__ add, int HW, int C, int G, int silu, f16 *__restrict__ dx)
{
    const int b = blockIdx.y, c8n = C / 8, cg = C / G;
    const size_t total = (size_t)HW * c8n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c0 = (int)(i % c8n) * 8;
        const size_t off = (size_t)b * HW * C + i * 8;
        const f16x8 xv = *(const f16x8 *)(x + off), dv = *(const f16x8 *)(dy + off);
        f16x8 av = {0, 0, 0, 0, 0, 0, 0, 0};
        if (add) av = *(const f16x8 *)(add + off);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c0 + j, g = c / cg;
            const float mu = mr[((size_t)b * G + g) * 2], rs = mr[((size_t)b * G + g) * 2 + 1];
            const float c1 = cc[((size_t)b * G + g) * 2], c2 = cc[((size_t)b * G + g) * 2 + 1];
            const float ga = (float)gamma[c], xh = ((float)xv[j] - mu) * rs, u = xh * ga + (float)beta[c];
            float d = (float)dv[j];
            if (silu) { const float sg = 1.0f / (1.0f + __expf(-u)); d *= sg * (1.0f + u * (1.0f - sg)); }
            o[j] = (f16)(rs * (d * ga - c1 - xh * c2) + (float)av[j]);
        }
        *(f16x8 *)(dx + off) = o;
    }
}

// GroupNorm(+SiLU) backward: dx = d(loss)/dx (+ add).  ws: 2 * B * NS * G * 2 + 2 * B * G * 2 floats.
static int gn_bwd(ctx_vae *v, const f16 *x, const f16 *dy, size_t g, size_t b, const f16 *add, int B, int HW, int C, int silu, f16 *dx, float *ws)
{
    const int G = v->cfg.groups, c8n = C / 8;
    if (C % 8 || 256 % c8n || C % G) { ctx_set_error("vae backward: GroupNorm with C=%d groups=%d is outside the kernel's envelope", C, G); return CTX_E_ARG; }
    const int PL = 256 / c8n;
    int NS = HW / (PL * 4); if (NS < 1) NS = 1; if (NS > GNB_MAX_SPLITS) NS = GNB_MAX_SPLITS;
    float *part = ws, *mr = ws + (size_t)B * GNB_MAX_SPLITS * G * 2, *cc = mr + (size_t)B * G * 2;
    const size_t lds = ((size_t)PL * C * 2 + (size_t)C * 2) * sizeof(float);
    const float n = (float)HW * (float)(C / G);
    hipLaunchKernelGGL(k_gnb_reduce<0>, dim3(NS, B), dim3(256), lds, v->s, x, (const f16 *)nullptr, v->W + g, v->W + b, (const float *)nullptr, HW, C, G, NS, silu, part);
    hipLaunchKernelGGL(k_gnb_finalize<0>, dim3(B), dim3(64), 0, v->s, part, G, NS, n, 1e-6f, mr);
    hipLaunchKernelGGL(k_gnb_reduce<1>, dim3(NS, B), dim3(256), lds, v->s, x, dy, v->W + g, v->W + b, mr, HW, C, G, NS, silu, part);
    hipLaunchKernelGGL(k_gnb_finalize<1>, dim3(B), dim3(64), 0, v->s, part, G, NS, n, 0.f, cc);
    const size_t total = (size_t)HW * c8n;
    unsigned nb = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_gnb_apply, dim3(nb, B), dim3(256), 0, v->s, x, dy, v->W + g, v->W + b, mr, cc, add, HW, C, G, silu, dx);
    return 0;
}
static size_t gn_bwd_ws_floats(int B, int G) { return (size_t)B * GNB_MAX_SPLITS * G * 2 + (size_t)B * G * 4; }

// dS = P * (dP - rowsum(dP * P)) * scale, one workgroup per row
__global__ __launch_bounds__(256) void k_softmax_bwd_rows(const f16 *__restrict__ P, const f16 *__restrict__ dP, int n, float scale, f16 *__restrict__ dS)
{
    const f16 *pr = P + (size_t)blockIdx.x * n, *dr = dP + (size_t)blockIdx.x * n;
    f16 *out = dS + (size_t)blockIdx.x * n;
    __shared__ float red[4];
    float dot = 0.f;
    for (int i = threadIdx.x * 8; i < n; i += 2048) {
        f16x8 a = *(const f16x8 *)(pr + i), b = *(const f16x8 *)(dr + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) dot += (float)a[j] * (float)b[j];
    }
    dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    dot = (red[0] + red[1]) + (red[2] + red[3]);
    for (int i = threadIdx.x * 8; i < n; i += 2048) {
        f16x8 a = *(const f16x8 *)(pr + i), b = *(const f16x8 *)(dr + i), o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)((float)a[j] * ((float)b[j] - dot) * scale);
        *(f16x8 *)(out + i) = o;
    }
}

// quant_conv backward: g f32 NCHW [B,C,hw] -> d(m16) f16 NHWC padded to 64 channels, times gscale
__global__ __launch_bounds__(256) void k_quant_bwd(const float *__restrict__ g, const f16 *__restrict__ w, int B, int C, int64_t HW, float gscale,
                                                   f16 *__restrict__ d)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)B * HW; i += (int64_t)gridDim.x * 256) {
        const int bb = (int)(i / HW); const int64_t p = i % HW;
        float in[16];
        for (int o = 0; o < C; ++o) in[o] = g[((int64_t)bb * C + o) * HW + p] * gscale;
        for (int c8 = 0; c8 < 8; ++c8) {
            f16x8 o8 = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int j = 0; j < 8; ++j) {
                const int c = c8 * 8 + j;
                if (c < C) { float acc = 0.f; for (int o = 0; o < C; ++o) acc += in[o] * (float)w[o * C + c]; o8[j] = (f16)acc; }
            }
            *(f16x8 *)(d + i * 64 + c8 * 8) = o8;
        }
    }
}

// conv_in backward: dy f16 NHWC [B,H,W,C] -> d(image) f32 NCHW [B,Cimg,H,W] / gscale; w = the forward pack [C][3][3][8]
__global__ __launch_bounds__(256) void k_conv_in_bwd(const f16 *__restrict__ dy, const f16 *__restrict__ w, int B, int H, int W, int C, int Cimg,
                                                     float inv_gscale, float *__restrict__ dimg)
{
    extern __shared__ float s_w[];                   // [9][Cimg(<=4)][C]
    for (int i = threadIdx.x; i < 9 * 4 * C; i += 256) {
        const int c = i % C, ci = (i / C) % 4, t = i / (4 * C);
        s_w[i] = ci < Cimg ? (float)w[((size_t)c * 9 + t) * 8 + ci] : 0.f;
    }
    __syncthreads();
    const int64_t npix = (int64_t)B * H * W;
    for (int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * 256) {
        const int b = (int)(pix / ((int64_t)H * W)), p = (int)(pix % ((int64_t)H * W));
        const int iy = p / W, ix = p % W;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < 9; ++t) {
            // y[oy,ox] += x[oy+ky-1, ox+kx-1] w[ky,kx]  =>  dx[iy,ix] += dy[iy-ky+1, ix-kx+1] w[ky,kx]
            const int oy = iy - t / 3 + 1, ox = ix - t % 3 + 1;
            if (oy < 0 || oy >= H || ox < 0 || ox >= W) continue;
            const f16 *dp = dy + (((size_t)b * H + oy) * W + ox) * C;
            for (int c = 0; c < C; c += 8) {
                const f16x8 dv = *(const f16x8 *)(dp + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = (float)dv[j];
#pragma unroll
                    for (int ci = 0; ci < 4; ++ci) acc[ci] += d * s_w[(t * 4 + ci) * C + c + j];
                }
            }
        }
        for (int ci = 0; ci < Cimg; ++ci) dimg[(((size_t)b * Cimg + ci) * H + iy) * W + ix] = acc[ci] * inv_gscale;
    }
}

// generic conv launch for the backward (weights by pointer, optional bias / residual, explicit geometry)
static void vconv_ex(ctx_vae *v, const f16 *x, const f16 *Wt, const f16 *res, int B, int H, int W, int Cin, int Cout, int ups, int poff, int zins,
                     int Ho, int Wo, f16 *out)
{
    GemmArgs a = {};
    a.Ho = Ho; a.Wo = Wo;
    a.X = x; a.Wt = Wt; a.bias = nullptr; a.residual = res; a.out = out;
    a.M = B * Ho * Wo; a.N = Cout; a.K = 9 * Cin; a.ldc = Cout; a.ldr = Cout; a.rows_per_batch = Ho * Wo; a.ldrb = Cout;
    a.H = H; a.W = W; a.Cin = Cin; a.stride = 1; a.ups = ups; a.poff = poff; a.zins = zins;
    v->flops += 2.0 * a.M * a.N * a.K;
    size_t mark = v->top;
    ctx_gemm_plan(a, true);
    if (zins) a.use8 = 0;
    if (a.splitk > 1) a.part = (float *)v->alloc((size_t)a.splitk * a.M * a.N * 4);
    VRUN(ctx_gemm_dispatch(a, true, v->s));
    v->top = mark;
}
static void vgemm_ld(ctx_vae *v, const f16 *X, const f16 *Wt, const f16 *res, int M, int N, int K, f16 *out, int ldc)
{
    GemmArgs a = {};
    a.X = X; a.Wt = Wt; a.bias = nullptr; a.residual = res; a.out = out; a.M = M; a.N = N; a.K = K; a.ldc = ldc; a.ldr = N;
    a.rows_per_batch = 1; a.ldrb = N; a.epi = 0;
    v->flops += 2.0 * M * N * K;
    size_t mark = v->top;
    ctx_gemm_plan(a, false);
    if (a.splitk > 1) a.part = (float *)v->alloc((size_t)a.splitk * M * N * 4);
    VRUN(ctx_gemm_dispatch(a, false, v->s));
    v->top = mark;
}

// resnet backward: dout [M,cout] -> dx [M,cin] (dx may alias nothing the forward still needs)
static int vres_bwd(ctx_vae *v, const VRes &r, const ctx_vae::ResTape &tp, const f16 *dout, int B, int H, int W, f16 *dx, float *gws)
{
    const size_t M = (size_t)B * H * W;
    size_t mark = v->top;
    f16 *dt2 = v->allocH(M * r.cout);
    vconv_ex(v, dout, v->W + r.c2wT, nullptr, B, H, W, r.cout, r.cout, 0, 0, 0, H, W, dt2);
    f16 *dh = v->allocH(M * r.cout);
    if (!v->dry && v->rc == 0) { int e = gn_bwd(v, tp.h, dt2, r.n2g, r.n2b, nullptr, B, H * W, r.cout, 1, dh, gws); if (e) return e; }
    f16 *dt1 = dt2;                                   // dt2 is dead
    if (r.cin != r.cout) dt1 = v->allocH(M * r.cin);
    vconv_ex(v, dh, v->W + r.c1wT, nullptr, B, H, W, r.cout, r.cin, 0, 0, 0, H, W, dt1);
    const f16 *dsc = dout;
    if (r.cin != r.cout) {
        f16 *s2 = v->allocH(M * r.cin);
        vgemm_ld(v, dout, v->W + r.scwT, nullptr, (int)M, r.cin, r.cout, s2, r.cin);
        dsc = s2;
    }
    if (!v->dry && v->rc == 0) { int e = gn_bwd(v, tp.x, dt1, r.n1g, r.n1b, dsc, B, H * W, r.cin, 1, dx, gws); if (e) return e; }
    v->top = mark;
    return 0;
}

// mid-block attention backward: x_out = o + proj(softmax(q k^T / sqrt(top)) v), q|k|v = gn(o) Wqkv^T + b
static int vattn_bwd(ctx_vae *v, const VAttn &at, const f16 *o, const f16 *qkv, const f16 *dx, f16 *d_o, int B, int h, int w, int top, float *gws)
{
    const int S = h * w, M = B * S;
    size_t mark = v->top;
    f16 *datt = v->allocH((size_t)M * top);
    vgemm_ld(v, dx, v->W + at.owT, nullptr, M, top, top, datt, top);
    f16 *dqkv = v->allocH((size_t)M * 3 * top);
    f16 *sc = v->allocH((size_t)S * S), *pr = v->allocH((size_t)S * S), *dp = v->allocH((size_t)S * S), *tr = v->allocH((size_t)S * S);
    f16 *qb = v->allocH((size_t)S * top), *kb = v->allocH((size_t)S * top), *vb = v->allocH((size_t)S * top), *tb = v->allocH((size_t)S * top);
    const float scale = 1.0f / sqrtf((float)top);
    for (int b = 0; b < B; ++b) {
        const f16 *base = qkv ? qkv + (size_t)b * S * 3 * top : nullptr;
        const f16 *da = datt ? datt + (size_t)b * S * top : nullptr;
        f16 *dq = dqkv ? dqkv + (size_t)b * S * 3 * top : nullptr;
        if (!v->dry) {
            (void)hipMemcpy2DAsync(qb, (size_t)top * 2, base, (size_t)3 * top * 2, (size_t)top * 2, S, hipMemcpyDeviceToDevice, v->s);
            (void)hipMemcpy2DAsync(kb, (size_t)top * 2, base + top, (size_t)3 * top * 2, (size_t)top * 2, S, hipMemcpyDeviceToDevice, v->s);
            (void)hipMemcpy2DAsync(vb, (size_t)top * 2, base + 2 * top, (size_t)3 * top * 2, (size_t)top * 2, S, hipMemcpyDeviceToDevice, v->s);
        }
        vgemm_ld(v, qb, kb, nullptr, S, S, top, sc, S);                                             // scores (recomputed)
        if (!v->dry) hipLaunchKernelGGL(k_softmax_rows, dim3(S), dim3(256), 0, v->s, sc, S, 1.4426950408889634f * scale, pr);
        vgemm_ld(v, da, vb, nullptr, S, S, top, dp, S);                                             // dP = dAtt V^T
        // dV = P^T dAtt : X = P^T [S,S], Wt = dAtt^T [top,S]
        VRUN(ctx_transpose_v_f16(pr, 1, S, S, S / 64, S, 0, tr, v->s));
        VRUN(ctx_transpose_v_f16(da, 1, S, top, top / 64, S, 0, tb, v->s));
        vgemm_ld(v, tr, tb, nullptr, S, top, S, dq ? dq + 2 * top : nullptr, 3 * top);
        if (!v->dry) hipLaunchKernelGGL(k_softmax_bwd_rows, dim3(S), dim3(256), 0, v->s, pr, dp, S, scale, sc);   // dS -> sc
        // dQ = dS K : Wt = K^T [top,S]
        VRUN(ctx_transpose_v_f16(kb, 1, S, top, top / 64, S, 0, tb, v->s));
        vgemm_ld(v, sc, tb, nullptr, S, top, S, dq, 3 * top);
        // dK = dS^T Q : X = dS^T, Wt = Q^T
        VRUN(ctx_transpose_v_f16(sc, 1, S, S, S / 64, S, 0, tr, v->s));
        VRUN(ctx_transpose_v_f16(qb, 1, S, top, top / 64, S, 0, tb, v->s));
        vgemm_ld(v, tr, tb, nullptr, S, top, S, dq ? dq + top : nullptr, 3 * top);
    }
    f16 *dg = datt;                                    // datt is dead
    vgemm_ld(v, dqkv, v->W + at.qkvT, nullptr, M, top, 3 * top, dg, top);
    if (!v->dry && v->rc == 0) { int e = gn_bwd(v, o, dg, at.ng, at.nb, dx, B, S, top, 0, d_o, gws); if (e) return e; }
    v->top = mark;
    return 0;
}

static int vae_encode_bwd_run(ctx_vae *v, const float *gmom, float gscale, float *dimg)
{
    const ctx_vae_config_t &c = v->cfg;
    const int n = c.n_levels, top = c.block_out_channels[n - 1], L2 = 2 * c.latent_channels;
    const int B = v->tape.B, H = v->tape.H, W = v->tape.W;
    int h = H >> (n - 1), w = W >> (n - 1);
    v->top = v->tape.top; v->rc = 0;
    float *gws = (float *)v->alloc(gn_bwd_ws_floats(B, c.groups) * sizeof(float));
    const size_t Ml = (size_t)B * h * w;
    f16 *dm = v->allocH(Ml * 64);
    if (!v->dry) hipLaunchKernelGGL(k_quant_bwd, dim3((unsigned)cdiv64((int64_t)Ml, 256)), dim3(256), 0, v->s, gmom, v->W + v->qw, B, L2, (int64_t)h * w, gscale, dm);
    // two rotating gradient buffers sized for the largest activation of the encoder
    size_t big = 0;
    { int hh = H, ww = W; for (int i = 0; i < n; ++i) { big = std::max(big, (size_t)B * hh * ww * c.block_out_channels[i]); if (i != n - 1) { hh /= 2; ww /= 2; } } }
    f16 *ga = v->allocH(big), *gb = v->allocH(big);
    vconv_ex(v, dm, v->W + v->e_cowT, nullptr, B, h, w, 64, top, 0, 0, 0, h, w, ga);                         // conv_out dgrad -> dy [M,top]
    if (!v->dry && v->rc == 0) { int e = gn_bwd(v, v->tape.norm_out_in, ga, v->e_cng, v->e_cnb, nullptr, B, h * w, top, 1, gb, gws); if (e) return e; }
    std::swap(ga, gb);                                                                                         // ga = current gradient
    int ri = (int)v->tape.res.size() - 1;
    auto tape_at = [&](int k) { return v->dry ? ctx_vae::ResTape{nullptr, nullptr} : v->tape.res[k]; };
    { int e = vres_bwd(v, v->e_mid[1], tape_at(ri--), ga, B, h, w, gb, gws); if (e) return e; std::swap(ga, gb); }
    { int e = vattn_bwd(v, v->e_att, v->tape.attn_in, v->tape.qkv, ga, gb, B, h, w, top, gws); if (e) return e; std::swap(ga, gb); }
    { int e = vres_bwd(v, v->e_mid[0], tape_at(ri--), ga, B, h, w, gb, gws); if (e) return e; std::swap(ga, gb); }
    for (int i = n - 1; i >= 0; --i) {
        const int cc = c.block_out_channels[i];
        if (i != n - 1) {
            // downsampler backward: stride-2, pad (0,1,0,1) conv -> zero-inserted 2x grid, flipped weights, offset -2
            vconv_ex(v, ga, v->W + v->dnwT[i], nullptr, B, h, w, cc, cc, 1, -1, 1, 2 * h, 2 * w, gb);
            std::swap(ga, gb); h *= 2; w *= 2;
        }
        for (int j = (int)v->down[i].size() - 1; j >= 0; --j) {
            int e = vres_bwd(v, v->down[i][j], tape_at(ri--), ga, B, h, w, gb, gws); if (e) return e; std::swap(ga, gb);
        }
    }
    if (!v->dry && v->rc == 0) {
        const int C0 = c.block_out_channels[0];
        hipLaunchKernelGGL(k_conv_in_bwd, dim3((unsigned)std::min<int64_t>(cdiv64((int64_t)B * H * W, 256), 8192)), dim3(256), (size_t)9 * 4 * C0 * sizeof(float),
                           v->s, ga, v->W + v->e_ciw, B, H, W, C0, c.out_channels, 1.0f / gscale, dimg);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { ctx_set_error("vae_encode_bwd: launch failed: %s", hipGetErrorString(e)); return CTX_E_LAUNCH; }
    }
    return v->rc;
}

extern "C" int64_t ctx_vae_encode_train_workspace_bytes(const ctx_vae_t *cv, int32_t B, int32_t H, int32_t W)
{
    ctx_vae *v = const_cast<ctx_vae *>(cv);
    if (!v || !vae_encode_dims_ok(v, B, H, W)) return -1;
    v->dry = true; v->train = true;
    vae_encode_run(v, nullptr, B, H, W, nullptr);
    v->tape.B = B; v->tape.H = H; v->tape.W = W;
    size_t fwd_peak = v->peak;
    vae_encode_bwd_run(v, nullptr, 1.0f, nullptr);
    v->dry = false; v->train = false; v->tape.valid = false;
    return (int64_t)std::max(fwd_peak, v->peak) + 4096;
}

extern "C" int32_t ctx_vae_encode_train(ctx_vae_t *v, const float *image, int32_t B, int32_t H, int32_t W, float *moments, ctx_stream_t stream)
{
    CTX_REQUIRE(v && image && moments && v->W && v->ws, "vae_encode_train: null pointer / not bound");
    CTX_REQUIRE(vae_encode_dims_ok(v, B, H, W), "vae_encode_train: need H, W multiples of %d with (H/f)*(W/f) %% 64 == 0 (B=%d H=%d W=%d)",
                1 << (v->cfg.n_levels - 1), B, H, W);
    v->s = (hipStream_t)stream; v->dry = false; v->train = true;
    int rc = vae_encode_run(v, image, B, H, W, moments);
    v->train = false;
    if (rc) v->tape.valid = false;
    return rc;
}

extern "C" int32_t ctx_vae_encode_bwd(ctx_vae_t *v, const float *grad_moments, float gscale, float *grad_image, ctx_stream_t stream)
{
    CTX_REQUIRE(v && grad_moments && grad_image && v->W && v->ws, "vae_encode_bwd: null pointer / not bound");
    CTX_REQUIRE(v->tape.valid, "vae_encode_bwd: no tape (call ctx_vae_encode_train first; any other call on this handle drops the tape)");
    CTX_REQUIRE(gscale > 0.f, "vae_encode_bwd: gscale must be positive");
    v->s = (hipStream_t)stream; v->dry = false;
    int rc = vae_encode_bwd_run(v, grad_moments, gscale, grad_image);
    v->tape.valid = false;
    return rc;
}

extern "C" double ctx_vae_flops(const ctx_vae_t *v) { return v ? v->flops : 0.0; }
