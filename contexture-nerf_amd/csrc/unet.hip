// UNet denoise engine: the SD2-depth `UNet2DConditionModel` graph (diffusers 0.27.2 naming) executed
// as a fixed sequence of hand-written gfx950 kernels on one HIP stream.  Replaces
//   self.unet(latent_model_input_depth, t, encoder_hidden_states=text_embeddings)['sample']
// at src/stable_diffusion_depth.py:422-423.
//
// Memory: one fp16 weight blob (caller-allocated, filled by ctx_unet_set_param from fp32 diffusers-layout
// tensors: conv weights repacked to [Cout][ky][kx][Cin], q/k/v fused, GEGLU rows interleaved so the gate
// lives in the same lane as its value, all time_emb_proj layers concatenated into one GEMV) and one
// workspace arena (bump allocator with mark/release; sized by a dry run of the same code path).
// Activations are NHWC fp16; statistics, softmax and accumulators are fp32.
#include "common.h"
#include "kernels.h"
#include <string>
#include <vector>
#include <string.h>

enum PackKind { PK_COPY = 0, PK_CONV3 = 1, PK_CONVIN = 2, PK_GEGLU_W = 3, PK_GEGLU_B = 4 };

struct Param {
    std::string name;
    int ndim;
    int64_t shape[4];
    int kind;
    size_t dst;      // element offset into the fp16 weight blob
    int a, b;        // kind-specific dims
};

struct ResP {
    int cin, cout;
    size_t n1g, n1b, c1w, c1b, n2g, n2b, c2w, c2b, scw, scb;
    int temb_row;
};
struct TrP {
    int C, heads;
    size_t ng, nb, piw, pib, l1g, l1b, qkv, o1w, o1b, l2g, l2b, q2, kv2, o2w, o2b, l3g, l3b, f1w, f1b, f2w, f2b, pow_, pob;
    int kv_row;     // first row of this block's [k ; v] slice in the fused cross-attention K/V projection
};
struct LevelP {
    std::vector<ResP> res;
    std::vector<TrP> tr;
    bool has_attn = false, has_sampler = false;
    size_t sw = 0, sb = 0;
    int sc = 0;
};

struct ctx_unet {
    ctx_unet_config_t cfg;
    std::vector<Param> params;
    size_t wtop = 0;           // elements
    // parameter offsets
    size_t ciw, cib, t1w, t1b, t2w, t2b, tpw, tpb, cng, cnb, cow, cob;
    int temb_dim = 0, temb_rows = 0;
    // all cross-attention to_k / to_v weights live in one [kv_rows_total, cross_dim] matrix (one GEMM per forward)
    size_t kvw = 0;
    int kv_rows_total = 0, kv_rows = 0;
    std::vector<LevelP> down, up;
    LevelP mid;
    // bound memory
    f16 *W = nullptr;
    char *ws = nullptr;
    size_t ws_cap = 0;
    // arena state
    size_t top = 0, peak = 0;
    bool dry = false;
    hipStream_t s = nullptr;
    int rc = 0;
    // stats
    int64_t launches[3] = {0, 0, 0};
    double flops[3] = {0, 0, 0};
    // reference-only self-attention (Zero123++'s RefOnlyNoisedUNet, spec in src/zero123plus.py:127-237): a 'w' pass parks the
    // attn1 inputs (LayerNorm-1 outputs) of the noised condition latent in a bank, an 'r' pass appends them to the K/V source of
    // the same attn1 layer for the batch rows >= ref_row0 (is_cfg_guidance: the unconditional row 0 attends without them)
    struct RefSlot { size_t off; int tokens, C, rows; };
    int ref_mode = 0;              // 0 off, 1 'w', 2 'r'
    f16 *ref_bank = nullptr;
    size_t ref_cursor = 0;         // elements
    int ref_k = 0, ref_row0 = 0;
    std::vector<RefSlot> ref_slots;
    // ControlNet (diffusers ControlNetModel.from_unet: the UNet's conv_in / time embedding / down blocks / mid block + a
    // conditioning-image embedding + 1x1 "zero" convolutions; Zero123++'s DepthControlUNet, spec in src/zero123plus.py:260-298)
    bool is_controlnet = false;
    int cond_channels = 3;
    struct CondConv { size_t w, b; int cin, cout, stride; };
    std::vector<CondConv> cond_convs;          // conv_in, blocks.0..5, conv_out
    std::vector<size_t> zc_w, zc_b;            // controlnet_down_blocks.i (one per skip tensor)
    std::vector<int> zc_c;
    size_t zm_w = 0, zm_b = 0;                 // controlnet_mid_block
    const float *cn_cond = nullptr;            // this pass's conditioning image f32 NCHW [B, cond_channels, 8H, 8W]
    f16 *cn_cache = nullptr;                   // caller-held [B,H,W,256] output of the embedding's few-channel layers
    bool cn_cache_valid = false;               // true: the image is the one the cache was computed from (it does not change per step)
    f16 *cn_out = nullptr;                     // ControlNet pass: residuals out (skip order, then mid), fp16 NHWC
    // main UNet pass: residuals to add (same layout) and their scale
    const f16 *add_res = nullptr;
    float add_scale = 1.0f;
    // experiment: keep the residual stream — every tensor that is the sum of a block's input and its
    // branch, and the skip copies of those — in fp32; GEMM / conv operands stay fp16 (cast on the way in), accumulators fp32
    bool res32 = false;
    void *allocS(size_t n) { return alloc(n * (res32 ? 4 : 2)); }          // a residual-stream tensor of n elements
    // measurement aid (tools/precision_attribution.py): copy every block's output (fp16 NHWC) into a caller buffer
    struct Tap { size_t off; int rows, C; };
    f16 *tap_buf = nullptr;
    size_t tap_cap = 0, tap_cursor = 0;        // elements
    std::vector<Tap> taps;
    void tap(const void *x, int rows, int C)
    {
        if (!tap_buf || res32) return;
        const size_t n = (size_t)rows * C;
        Tap t = {tap_cursor, rows, C};
        taps.push_back(t);
        if (!dry && rc == 0 && tap_cursor + n <= tap_cap) (void)hipMemcpyAsync(tap_buf + tap_cursor, x, n * 2, hipMemcpyDeviceToDevice, s);
        tap_cursor += n;
    }

    size_t walloc(size_t n) { size_t o = wtop; wtop += (n + 127) / 128 * 128; return o; }
    size_t add(const std::string &name, std::vector<int64_t> shp, int kind, size_t dst, int a = 0, int b = 0)
    {
        Param p; p.name = name; p.ndim = (int)shp.size(); p.kind = kind; p.dst = dst; p.a = a; p.b = b;
        for (int i = 0; i < 4; ++i) p.shape[i] = i < p.ndim ? shp[i] : 1;
        params.push_back(p);
        return dst;
    }
    size_t vec(const std::string &name, int n) { return add(name, {n}, PK_COPY, walloc(n)); }
    size_t lin(const std::string &name, int out, int in) { return add(name, {out, in}, PK_COPY, walloc((size_t)out * in)); }

    void *alloc(size_t bytes)
    {
        size_t o = (top + 255) / 256 * 256;
        top = o + bytes;
        if (top > peak) peak = top;
        if (!dry && top > ws_cap) { rc = CTX_E_STATE; ctx_set_error("unet: workspace too small (%zu > %zu)", top, ws_cap); return ws; }
        return dry ? nullptr : (void *)(ws + o);
    }
    f16 *allocH(size_t n) { return (f16 *)alloc(n * 2); }
};

// ------------------------------------------------------------------------------------------------
static void add_resnet(ctx_unet *u, const std::string &p, int cin, int cout, ResP &r)
{
    r.cin = cin; r.cout = cout;
    r.n1g = u->vec(p + ".norm1.weight", cin); r.n1b = u->vec(p + ".norm1.bias", cin);
    r.c1w = u->add(p + ".conv1.weight", {cout, cin, 3, 3}, PK_CONV3, u->walloc((size_t)cout * cin * 9), cout, cin);
    r.c1b = u->vec(p + ".conv1.bias", cout);
    // time_emb_proj rows live in the concatenated [sum Cout, temb] matrix
    r.temb_row = u->temb_rows;
    u->add(p + ".time_emb_proj.weight", {cout, u->temb_dim}, PK_COPY, u->tpw + (size_t)u->temb_rows * u->temb_dim);
    u->add(p + ".time_emb_proj.bias", {cout}, PK_COPY, u->tpb + u->temb_rows);
    u->temb_rows += cout;
    r.n2g = u->vec(p + ".norm2.weight", cout); r.n2b = u->vec(p + ".norm2.bias", cout);
    r.c2w = u->add(p + ".conv2.weight", {cout, cout, 3, 3}, PK_CONV3, u->walloc((size_t)cout * cout * 9), cout, cout);
    r.c2b = u->vec(p + ".conv2.bias", cout);
    if (cin != cout) {
        r.scw = u->add(p + ".conv_shortcut.weight", {cout, cin, 1, 1}, PK_COPY, u->walloc((size_t)cout * cin));
        r.scb = u->vec(p + ".conv_shortcut.bias", cout);
    } else r.scw = r.scb = 0;
}

static void add_transformer(ctx_unet *u, const std::string &p, int C, int heads, TrP &t)
{
    int cd = u->cfg.cross_attention_dim;
    t.C = C; t.heads = heads;
    t.ng = u->vec(p + ".norm.weight", C); t.nb = u->vec(p + ".norm.bias", C);
    t.piw = u->lin(p + ".proj_in.weight", C, C); t.pib = u->vec(p + ".proj_in.bias", C);
    std::string b = p + ".transformer_blocks.0";
    t.l1g = u->vec(b + ".norm1.weight", C); t.l1b = u->vec(b + ".norm1.bias", C);
    t.qkv = u->walloc((size_t)3 * C * C);
    u->add(b + ".attn1.to_q.weight", {C, C}, PK_COPY, t.qkv);
    u->add(b + ".attn1.to_k.weight", {C, C}, PK_COPY, t.qkv + (size_t)C * C);
    u->add(b + ".attn1.to_v.weight", {C, C}, PK_COPY, t.qkv + (size_t)2 * C * C);
    t.o1w = u->lin(b + ".attn1.to_out.0.weight", C, C); t.o1b = u->vec(b + ".attn1.to_out.0.bias", C);
    t.l2g = u->vec(b + ".norm2.weight", C); t.l2b = u->vec(b + ".norm2.bias", C);
    t.q2 = u->lin(b + ".attn2.to_q.weight", C, C);
    t.kv_row = u->kv_rows;
    t.kv2 = u->kvw + (size_t)u->kv_rows * cd;
    u->kv_rows += 2 * C;
    u->add(b + ".attn2.to_k.weight", {C, cd}, PK_COPY, t.kv2);
    u->add(b + ".attn2.to_v.weight", {C, cd}, PK_COPY, t.kv2 + (size_t)C * cd);
    t.o2w = u->lin(b + ".attn2.to_out.0.weight", C, C); t.o2b = u->vec(b + ".attn2.to_out.0.bias", C);
    t.l3g = u->vec(b + ".norm3.weight", C); t.l3b = u->vec(b + ".norm3.bias", C);
    t.f1w = u->add(b + ".ff.net.0.proj.weight", {8 * C, C}, PK_GEGLU_W, u->walloc((size_t)8 * C * C), 4 * C, C);
    t.f1b = u->add(b + ".ff.net.0.proj.bias", {8 * C}, PK_GEGLU_B, u->walloc((size_t)8 * C), 4 * C, 0);
    t.f2w = u->lin(b + ".ff.net.2.weight", C, 4 * C); t.f2b = u->vec(b + ".ff.net.2.bias", C);
    t.pow_ = u->lin(p + ".proj_out.weight", C, C); t.pob = u->vec(p + ".proj_out.bias", C);
}

static ctx_unet_t *unet_create_impl(const ctx_unet_config_t *cfg, bool controlnet, int cond_channels)
{
    if (!cfg || cfg->n_levels < 1 || cfg->n_levels > 4 || cfg->in_channels > 16 || cfg->out_channels > 4 ||
        cfg->groups > 64 || cfg->layers_per_block < 1 || cfg->layers_per_block > 4 || cfg->cross_attention_dim % 64) {
        ctx_set_error("unet_create: unsupported config");
        return nullptr;
    }
    for (int i = 0; i < cfg->n_levels; ++i)
        if (cfg->block_out_channels[i] % 64 || cfg->block_out_channels[i] % cfg->groups || cfg->heads[i] * 64 != cfg->block_out_channels[i]) {
            ctx_set_error("unet_create: level %d: channels %d must be a multiple of 64 and of groups, with head_dim 64 (heads=%d)",
                          i, cfg->block_out_channels[i], cfg->heads[i]);
            return nullptr;
        }
    ctx_unet *u = new ctx_unet();
    u->cfg = *cfg;
    const int n = cfg->n_levels, lpb = cfg->layers_per_block;
    const int *ch = cfg->block_out_channels;
    u->temb_dim = ch[0] * 4;
    // total time_emb_proj rows
    int rows = 0;
    u->is_controlnet = controlnet; u->cond_channels = cond_channels;
    for (int i = 0; i < n; ++i) rows += lpb * ch[i];
    rows += 2 * ch[n - 1];
    if (!controlnet)
        for (int i = 0; i < n; ++i) rows += (lpb + 1) * ch[n - 1 - i];
    u->tpw = u->walloc((size_t)rows * u->temb_dim);
    u->tpb = u->walloc(rows);
    {
        int kvr = 2 * ch[n - 1];                                        // mid block
        for (int i = 0; i < n; ++i) {
            if (cfg->down_attn[i]) kvr += lpb * 2 * ch[i];
            if (cfg->up_attn[i] && !controlnet) kvr += (lpb + 1) * 2 * ch[n - 1 - i];
        }
        u->kv_rows_total = kvr;
        u->kvw = u->walloc((size_t)kvr * cfg->cross_attention_dim);
    }

    u->ciw = u->add("conv_in.weight", {ch[0], cfg->in_channels, 3, 3}, PK_CONVIN, u->walloc((size_t)ch[0] * 9 * (cfg->in_channels > 8 ? 16 : 8)), ch[0],
                    cfg->in_channels);
    u->cib = u->vec("conv_in.bias", ch[0]);
    u->t1w = u->lin("time_embedding.linear_1.weight", u->temb_dim, ch[0]); u->t1b = u->vec("time_embedding.linear_1.bias", u->temb_dim);
    u->t2w = u->lin("time_embedding.linear_2.weight", u->temb_dim, u->temb_dim); u->t2b = u->vec("time_embedding.linear_2.bias", u->temb_dim);
    u->down.resize(n);
    int out = ch[0];
    for (int i = 0; i < n; ++i) {
        LevelP &L = u->down[i];
        int cin = out; out = ch[i];
        std::string p = "down_blocks." + std::to_string(i);
        L.res.resize(lpb);
        for (int j = 0; j < lpb; ++j) add_resnet(u, p + ".resnets." + std::to_string(j), j == 0 ? cin : out, out, L.res[j]);
        L.has_attn = cfg->down_attn[i] != 0;
        if (L.has_attn) {
            L.tr.resize(lpb);
            for (int j = 0; j < lpb; ++j) add_transformer(u, p + ".attentions." + std::to_string(j), out, cfg->heads[i], L.tr[j]);
        }
        if (i != n - 1) {
            L.has_sampler = true; L.sc = out;
            L.sw = u->add(p + ".downsamplers.0.conv.weight", {out, out, 3, 3}, PK_CONV3, u->walloc((size_t)out * out * 9), out, out);
            L.sb = u->vec(p + ".downsamplers.0.conv.bias", out);
        }
    }
    {
        LevelP &L = u->mid;
        L.res.resize(2); L.tr.resize(1); L.has_attn = true;
        add_resnet(u, "mid_block.resnets.0", ch[n - 1], ch[n - 1], L.res[0]);
        add_transformer(u, "mid_block.attentions.0", ch[n - 1], cfg->heads[n - 1], L.tr[0]);
        add_resnet(u, "mid_block.resnets.1", ch[n - 1], ch[n - 1], L.res[1]);
    }
    if (!controlnet) {
        u->up.resize(n);
        out = ch[n - 1];
        for (int i = 0; i < n; ++i) {
            LevelP &L = u->up[i];
            int prev = out; out = ch[n - 1 - i];
            int inp = ch[n - 1 - (i + 1 < n ? i + 1 : n - 1)];
            std::string p = "up_blocks." + std::to_string(i);
            L.res.resize(lpb + 1);
            for (int j = 0; j <= lpb; ++j) {
                int skip = j == lpb ? inp : out;
                int rin = j == 0 ? prev : out;
                add_resnet(u, p + ".resnets." + std::to_string(j), rin + skip, out, L.res[j]);
            }
            L.has_attn = cfg->up_attn[i] != 0;
            if (L.has_attn) {
                L.tr.resize(lpb + 1);
                for (int j = 0; j <= lpb; ++j) add_transformer(u, p + ".attentions." + std::to_string(j), out, cfg->heads[n - 1 - i], L.tr[j]);
            }
            if (i != n - 1) {
                L.has_sampler = true; L.sc = out;
                L.sw = u->add(p + ".upsamplers.0.conv.weight", {out, out, 3, 3}, PK_CONV3, u->walloc((size_t)out * out * 9), out, out);
                L.sb = u->vec(p + ".upsamplers.0.conv.bias", out);
            }
        }
        u->cng = u->vec("conv_norm_out.weight", ch[0]); u->cnb = u->vec("conv_norm_out.bias", ch[0]);
        u->cow = u->add("conv_out.weight", {cfg->out_channels, ch[0], 3, 3}, PK_CONV3, u->walloc((size_t)cfg->out_channels * ch[0] * 9), cfg->out_channels, ch[0]);
        u->cob = u->vec("conv_out.bias", cfg->out_channels);
    } else {
        // controlnet_cond_embedding: conv_in(cond -> 16) SiLU, [conv(c -> c) SiLU, conv(c -> c', stride 2) SiLU] x 3, conv_out(256 -> ch0)
        const int cc[4] = {16, 32, 96, 256};
        const std::string e = "controlnet_cond_embedding";
        auto addc = [&](const std::string &name, int cin, int cout, int stride) {
            ctx_unet::CondConv c; c.cin = cin; c.cout = cout; c.stride = stride;
            c.w = u->add(name + ".weight", {cout, cin, 3, 3}, PK_CONV3, u->walloc((size_t)cout * cin * 9), cout, cin);
            c.b = u->vec(name + ".bias", cout);
            u->cond_convs.push_back(c);
        };
        addc(e + ".conv_in", cond_channels, cc[0], 1);
        for (int i = 0; i < 3; ++i) {
            addc(e + ".blocks." + std::to_string(2 * i), cc[i], cc[i], 1);
            addc(e + ".blocks." + std::to_string(2 * i + 1), cc[i], cc[i + 1], 2);
        }
        addc(e + ".conv_out", cc[3], ch[0], 1);
        // zero convolutions, one per skip tensor: conv_in output, every resnet(+attention) output, every downsampler output
        std::vector<int> skc; skc.push_back(ch[0]);
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < lpb; ++j) skc.push_back(ch[i]);
            if (i != n - 1) skc.push_back(ch[i]);
        }
        for (size_t k = 0; k < skc.size(); ++k) {
            std::string p = "controlnet_down_blocks." + std::to_string(k);
            u->zc_w.push_back(u->add(p + ".weight", {skc[k], skc[k], 1, 1}, PK_COPY, u->walloc((size_t)skc[k] * skc[k])));
            u->zc_b.push_back(u->vec(p + ".bias", skc[k]));
            u->zc_c.push_back(skc[k]);
        }
        u->zm_w = u->add("controlnet_mid_block.weight", {ch[n - 1], ch[n - 1], 1, 1}, PK_COPY, u->walloc((size_t)ch[n - 1] * ch[n - 1]));
        u->zm_b = u->vec("controlnet_mid_block.bias", ch[n - 1]);
    }
    if (u->kv_rows != u->kv_rows_total) {
        ctx_set_error("unet_create: internal cross-attention K/V row count mismatch %d != %d", u->kv_rows, u->kv_rows_total);
        delete u;
        return nullptr;
    }
    if (u->temb_rows != rows) {
        ctx_set_error("unet_create: internal temb row count mismatch %d != %d", u->temb_rows, rows);
        delete u;
        return nullptr;
    }
    return u;
}

extern "C" ctx_unet_t *ctx_unet_create(const ctx_unet_config_t *cfg) { return unet_create_impl(cfg, false, 0); }
extern "C" ctx_unet_t *ctx_controlnet_create(const ctx_unet_config_t *cfg, int32_t cond_channels)
{
    if (cond_channels < 1 || cond_channels > 8) { ctx_set_error("controlnet_create: cond_channels=%d outside [1,8]", cond_channels); return nullptr; }
    return unet_create_impl(cfg, true, cond_channels);
}
extern "C" void ctx_unet_destroy(ctx_unet_t *u) { delete u; }
extern "C" int32_t ctx_unet_param_count(const ctx_unet_t *u) { return u ? (int32_t)u->params.size() : 0; }
extern "C" const char *ctx_unet_param_name(const ctx_unet_t *u, int32_t i)
{
    return (u && i >= 0 && i < (int)u->params.size()) ? u->params[i].name.c_str() : "";
}
extern "C" int32_t ctx_unet_param_shape(const ctx_unet_t *u, int32_t i, int64_t shape4[4])
{
    if (!u || i < 0 || i >= (int)u->params.size()) return 0;
    for (int k = 0; k < 4; ++k) shape4[k] = u->params[i].shape[k];
    return u->params[i].ndim;
}
extern "C" int64_t ctx_unet_weight_bytes(const ctx_unet_t *u) { return u ? (int64_t)u->wtop * 2 + 256 : 0; }

extern "C" int32_t ctx_unet_bind(ctx_unet_t *u, void *weights, void *workspace, int64_t workspace_bytes)
{
    CTX_REQUIRE(u && weights && workspace && workspace_bytes > 0, "unet_bind: bad args");
    CTX_REQUIRE(((uintptr_t)weights & 255) == 0 && ((uintptr_t)workspace & 255) == 0, "unet_bind: blobs must be 256-byte aligned");
    u->W = (f16 *)weights; u->ws = (char *)workspace; u->ws_cap = (size_t)workspace_bytes;
    return CTX_OK;
}

// ---- parameter repack kernels ----------------------------------------------------------------------
__global__ void k_pack_copy(const float *__restrict__ s, int64_t n, f16 *__restrict__ d)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d[i] = (f16)s[i];
}
// [Cout,Cin,3,3] -> [Cout][3][3][Cinp] (Cinp = Cin, or 8 for conv_in)
__global__ void k_pack_conv3(const float *__restrict__ s, int Cout, int Cin, int Cinp, f16 *__restrict__ d)
{
    int64_t n = (int64_t)Cout * 9 * Cinp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int c = (int)(i % Cinp);
        int tap = (int)((i / Cinp) % 9);
        int o = (int)(i / ((int64_t)Cinp * 9));
        d[i] = c < Cin ? (f16)s[((int64_t)o * Cin + c) * 9 + tap] : (f16)0.f;
    }
}
// GEGLU rows: packed row p (of 2*C4) <- source row  (w<32 ? blk*32+w : C4 + blk*32 + w-32), blk=p/64, w=p%64
__global__ void k_pack_geglu(const float *__restrict__ s, int C4, int K, f16 *__restrict__ d)
{
    int64_t n = (int64_t)2 * C4 * (K > 0 ? K : 1);
    int kk = K > 0 ? K : 1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int k = (int)(i % kk);
        int p = (int)(i / kk);
        int blk = p / 64, w = p % 64;
        int src = w < 32 ? blk * 32 + w : C4 + blk * 32 + (w - 32);
        d[i] = (f16)s[(int64_t)src * kk + k];
    }
}

extern "C" int32_t ctx_unet_set_param(ctx_unet_t *u, int32_t i, const float *src, ctx_stream_t stream)
{
    CTX_REQUIRE(u && u->W && src && i >= 0 && i < (int)u->params.size(), "unet_set_param: bad args / not bound");
    const Param &p = u->params[i];
    hipStream_t s = (hipStream_t)stream;
    int64_t n = 1;
    for (int k = 0; k < p.ndim; ++k) n *= p.shape[k];
    f16 *d = u->W + p.dst;
    int64_t nbk = cdiv64(n, 256);
    unsigned nb = (unsigned)(nbk > 4096 ? 4096 : nbk);
    switch (p.kind) {
    case PK_COPY: hipLaunchKernelGGL(k_pack_copy, dim3(nb), dim3(256), 0, s, src, n, d); break;
    case PK_CONV3: hipLaunchKernelGGL(k_pack_conv3, dim3(nb), dim3(256), 0, s, src, p.a, p.b, p.b, d); break;
    case PK_CONVIN: hipLaunchKernelGGL(k_pack_conv3, dim3(nb), dim3(256), 0, s, src, p.a, p.b, p.b > 8 ? 16 : 8, d); break;
    case PK_GEGLU_W: hipLaunchKernelGGL(k_pack_geglu, dim3(nb), dim3(256), 0, s, src, p.a, p.b, d); break;
    case PK_GEGLU_B: hipLaunchKernelGGL(k_pack_geglu, dim3(nb), dim3(256), 0, s, src, p.a, 0, d); break;
    }
    CTX_CHECK_LAUNCH("unet_set_param");
    return CTX_OK;
}

// 3x3 convolution (pad 1, stride 1 | 2) for the few-channel layers of ControlNet's conditioning embedding (3 -> 16 -> 16 -> 32
// -> 32 -> 96 -> 96 -> 256; they run once per conditioning image, not per denoise step).  One thread = one output pixel x 8 output
// channels; input either the f32 NCHW image (first layer) or f16 NHWC; weights [Cout][3][3][Cin] f16; optional SiLU; f16 NHWC out.
__global__ __launch_bounds__(256) void k_conv_small(const float *__restrict__ x32, const f16 *__restrict__ x16, const f16 *__restrict__ w,
                                                    const f16 *__restrict__ bias, int B, int H, int W, int Cin, int Cout, int stride,
                                                    int silu, f16 *__restrict__ y)
{
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1, o8n = Cout / 8;
    const int64_t total = (int64_t)B * Ho * Wo * o8n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int o8 = (int)(i % o8n);
        const int64_t pix = i / o8n;
        const int b = (int)(pix / ((int64_t)Ho * Wo)), p = (int)(pix % ((int64_t)Ho * Wo));
        const int oy = p / Wo, ox = p % Wo;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = (float)bias[o8 * 8 + j];
        for (int t = 0; t < 9; ++t) {
            const int iy = oy * stride + t / 3 - 1, ix = ox * stride + t % 3 - 1;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            if (x16 && (Cin & 7) == 0) {         // 16-byte chunks of the pixel's channels against 8 weight chunks (L1-resident)
                const f16 *xp = x16 + (((int64_t)b * H + iy) * W + ix) * Cin;
                const f16 *wp = w + ((int64_t)(o8 * 8) * 9 + t) * Cin;
                for (int c = 0; c < Cin; c += 8) {
                    const f16x8 xv = *(const f16x8 *)(xp + c);
                    float xf[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) xf[q] = (float)xv[q];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const f16x8 wv = *(const f16x8 *)(wp + (int64_t)j * 9 * Cin + c);
#pragma unroll
                        for (int q = 0; q < 8; ++q) acc[j] += xf[q] * (float)wv[q];
                    }
                }
                continue;
            }
            for (int c = 0; c < Cin; ++c) {
                const float v = x32 ? (float)(f16)x32[(((int64_t)b * Cin + c) * H + iy) * W + ix]
                                    : (float)x16[(((int64_t)b * H + iy) * W + ix) * Cin + c];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v * (float)w[(((int64_t)(o8 * 8 + j)) * 9 + t) * Cin + c];
            }
        }
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[j];
            if (silu) v = v / (1.0f + __expf(-v));
            o[j] = (f16)v;
        }
        *(f16x8 *)(y + pix * Cout + o8 * 8) = o;
    }
}

// dst += scale * src (fp16, 8 per thread); the residual injection of the ControlNet outputs
__global__ __launch_bounds__(256) void k_add_scaled_f16(f16 *__restrict__ dst, const f16 *__restrict__ src, float scale, int64_t n8)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        f16x8 a = *(const f16x8 *)(dst + i * 8), b = *(const f16x8 *)(src + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = (f16)((float)a[j] + scale * (float)b[j]);
        *(f16x8 *)(dst + i * 8) = a;
    }
}

// ---- op wrappers (skip launches on a dry run, keep accounting identical) ---------------------------------
static void note(ctx_unet *u, int klass, double fl, int n = 1) { u->launches[klass] += n; u->flops[klass] += fl; }
#define RUN(expr) do { if (!u->dry && u->rc == 0) { int r__ = (expr); if (r__ != 0) u->rc = r__; } } while (0)

// res / out are residual-stream tensors (fp32 when u->res32) iff res_s / out_s
static void op_gemm(ctx_unet *u, const f16 *X, size_t w, size_t bias, bool has_bias, const void *res, int M, int N, int K, void *out,
                    int epi = 0, bool res_s = false, bool out_s = false)
{
    GemmArgs a = {};
    a.X = X; a.Wt = u->W + w; a.bias = has_bias ? u->W + bias : nullptr; a.residual = (const f16 *)res; a.out = (f16 *)out;
    a.M = M; a.N = N; a.K = K; a.ldc = epi == 1 ? N / 2 : N; a.ldr = N; a.rows_per_batch = 1; a.ldrb = N; a.epi = epi;
    a.res32 = (u->res32 && res_s && res) ? 1 : 0; a.out32 = (u->res32 && out_s) ? 1 : 0;
    note(u, 0, 2.0 * M * N * K);
    size_t mark = u->top;
    ctx_gemm_plan(a, false);
    if (a.splitk > 1) a.part = (float *)u->alloc((size_t)a.splitk * M * N * 4);
    RUN(ctx_gemm_dispatch(a, false, u->s));
    u->top = mark;
}
static void op_conv(ctx_unet *u, const f16 *x, size_t w, size_t bias, const f16 *rowbias, int ldrb, const void *res, int B, int H,
                    int W, int Cin, int Cout, int stride, int ups, void *out, bool res_s = false, bool out_s = false)
{
    GemmArgs a = {};
    int Hv = H << ups, Wv = W << ups;
    a.Ho = (Hv - 1) / stride + 1; a.Wo = (Wv - 1) / stride + 1;
    a.X = x; a.Wt = u->W + w; a.bias = u->W + bias; a.rowbias = rowbias; a.residual = (const f16 *)res; a.out = (f16 *)out;
    a.M = B * a.Ho * a.Wo; a.N = Cout; a.K = 9 * Cin; a.ldc = Cout; a.ldr = Cout; a.rows_per_batch = a.Ho * a.Wo; a.ldrb = ldrb;
    a.H = H; a.W = W; a.Cin = Cin; a.stride = stride; a.ups = ups;
    a.res32 = (u->res32 && res_s && res) ? 1 : 0; a.out32 = (u->res32 && out_s) ? 1 : 0;
    note(u, 0, 2.0 * a.M * a.N * a.K);
    size_t mark = u->top;
    ctx_gemm_plan(a, true);
    if (a.splitk > 1) a.part = (float *)u->alloc((size_t)a.splitk * a.M * a.N * 4);
    RUN(ctx_gemm_dispatch(a, true, u->s));
    u->top = mark;
}
// x_s: x is a residual-stream tensor
static void op_gn(ctx_unet *u, const void *x, size_t g, size_t b, int B, int HW, int C, float eps, int silu, f16 *y, void *stats, bool x_s = true)
{
    note(u, 2, 0, 2);
    RUN(ctx_groupnorm_any(x, (u->res32 && x_s) ? 1 : 0, u->W + g, u->W + b, B, HW, C, u->cfg.groups, eps, silu, y, stats, u->s));
}
static void op_ln(ctx_unet *u, const void *x, size_t g, size_t b, int64_t rows, int C, f16 *y)
{
    note(u, 2, 0);
    RUN(ctx_layernorm_any(x, u->res32 ? 1 : 0, u->W + g, u->W + b, rows, C, 1e-5f, y, u->s));
}
// fp16 GEMM / conv operand of a residual-stream tensor: the tensor itself, or (res32) a rounded copy above the arena mark
static const f16 *op_as16(ctx_unet *u, const void *x, size_t n)
{
    if (!u->res32) return (const f16 *)x;
    f16 *c = u->allocH(n);
    note(u, 2, 0);
    RUN(ctx_f32_to_f16((const float *)x, (int64_t)n, c, u->s));
    return c;
}
static void op_attn(ctx_unet *u, const f16 *Q, const f16 *K, const f16 *V, int B, int Sq, int Skv, int heads, int qs, int kvs, f16 *O)
{
    note(u, 1, 4.0 * B * heads * (double)Sq * Skv * 64);
    RUN(ctx_attention_core(Q, K, V, B, Sq, Skv, heads, qs, kvs, 0.125f, O, heads * 64, u->s));
}

struct FwdCtx {
    int B, L;
    const f16 *tproj;   // [B, temb_rows]
    const f16 *ctx16;   // [B*L, cd]
    const f16 *kv_all;  // [B*L, kv_rows_total]: every block's cross-attention K | V projection of the context
    void *gn_stats;
};

static void *run_resnet(ctx_unet *u, const FwdCtx &f, const ResP &r, const void *x, int H, int W, void *out)
{
    const int B = f.B, HW = H * W, M = B * HW;
    size_t mark = u->top;
    f16 *t1 = u->allocH((size_t)M * r.cin);
    op_gn(u, x, r.n1g, r.n1b, B, HW, r.cin, u->cfg.norm_eps, 1, t1, f.gn_stats);
    f16 *h = u->allocH((size_t)M * r.cout);
    op_conv(u, t1, r.c1w, r.c1b, f.tproj ? f.tproj + r.temb_row : nullptr, u->temb_rows, nullptr, B, H, W, r.cin, r.cout, 1, 0, h);
    f16 *t2 = u->allocH((size_t)M * r.cout);
    op_gn(u, h, r.n2g, r.n2b, B, HW, r.cout, u->cfg.norm_eps, 1, t2, f.gn_stats, false);
    const void *sc = x;
    if (r.cin != r.cout) {
        void *s2 = u->allocS((size_t)M * r.cout);
        op_gemm(u, op_as16(u, x, (size_t)M * r.cin), r.scw, r.scb, true, nullptr, M, r.cout, r.cin, s2, 0, false, true);
        sc = s2;
    }
    op_conv(u, t2, r.c2w, r.c2b, nullptr, 0, sc, B, H, W, r.cout, r.cout, 1, 0, out, true, true);
    u->top = mark;
    return out;
}

static void *run_transformer(ctx_unet *u, const FwdCtx &f, const TrP &t, const void *x, int H, int W, void *out)
{
    const int B = f.B, S = H * W, M = B * S, C = t.C, cd = u->cfg.cross_attention_dim;
    size_t mark = u->top;
    f16 *g = u->allocH((size_t)M * C);
    op_gn(u, x, t.ng, t.nb, B, S, C, 1e-6f, 0, g, f.gn_stats);
    void *h0 = u->allocS((size_t)M * C);
    op_gemm(u, g, t.piw, t.pib, true, nullptr, M, C, C, h0, 0, false, true);
    // self attention
    f16 *l = g;   // reuse
    op_ln(u, h0, t.l1g, t.l1b, M, C, l);
    f16 *a = nullptr;
    if (u->ref_mode == 2) {
        // 'r': K/V of a batch row come from [its own tokens ; the condition's tokens parked by the 'w' pass]
        if (u->ref_k >= (int)u->ref_slots.size() || u->ref_slots[u->ref_k].C != C ||
            u->ref_slots[u->ref_k].rows != B - u->ref_row0) {
            u->rc = CTX_E_STATE;
            ctx_set_error("unet: the reference bank does not match this pass (run the 'w' pass on the condition latent first; layer %d)", u->ref_k);
            u->top = mark;
            return out;
        }
        const ctx_unet::RefSlot sl = u->ref_slots[u->ref_k++];
        f16 *q = u->allocH((size_t)M * C);
        op_gemm(u, l, t.qkv, 0, false, nullptr, M, C, C, q);
        a = u->allocH((size_t)M * C);
        f16 *kv = u->allocH((size_t)(S + sl.tokens) * 2 * C);
        for (int b = 0; b < B; ++b) {
            const int Sr = b >= u->ref_row0 ? sl.tokens : 0;
            op_gemm(u, l ? l + (size_t)b * S * C : nullptr, t.qkv + (size_t)C * C, 0, false, nullptr, S, 2 * C, C, kv);
            if (Sr)
                op_gemm(u, u->ref_bank ? u->ref_bank + sl.off + (size_t)(b - u->ref_row0) * sl.tokens * C : nullptr,
                        t.qkv + (size_t)C * C, 0, false, nullptr, Sr, 2 * C, C, kv ? kv + (size_t)S * 2 * C : nullptr);
            op_attn(u, q ? q + (size_t)b * S * C : nullptr, kv, kv ? kv + C : nullptr, 1, S, S + Sr, t.heads, C, 2 * C,
                    a ? a + (size_t)b * S * C : nullptr);
        }
    } else {
        if (u->ref_mode == 1) {    // 'w': park this layer's attn1 input
            ctx_unet::RefSlot sl; sl.off = u->ref_cursor; sl.tokens = S; sl.C = C; sl.rows = B;
            u->ref_slots.push_back(sl);
            u->ref_cursor += (size_t)M * C;
            if (!u->dry && u->rc == 0 && u->ref_bank)
                (void)hipMemcpyAsync(u->ref_bank + sl.off, l, (size_t)M * C * 2, hipMemcpyDeviceToDevice, u->s);
        }
        f16 *qkv = u->allocH((size_t)M * 3 * C);
        op_gemm(u, l, t.qkv, 0, false, nullptr, M, 3 * C, C, qkv);
        a = u->allocH((size_t)M * C);
        op_attn(u, qkv, qkv + C, qkv + 2 * C, B, S, S, t.heads, 3 * C, 3 * C, a);
    }
    void *h1 = u->allocS((size_t)M * C);
    op_gemm(u, a, t.o1w, t.o1b, true, h0, M, C, C, h1, 0, true, true);
    // cross attention
    op_ln(u, h1, t.l2g, t.l2b, M, C, l);
    f16 *q = a;   // reuse
    op_gemm(u, l, t.q2, 0, false, nullptr, M, C, C, q);
    const f16 *kv = f.kv_all + t.kv_row;
    f16 *a2 = u->allocH((size_t)M * C);
    op_attn(u, q, kv, kv + C, B, S, f.L, t.heads, C, u->kv_rows_total, a2);
    void *h2 = h0;  // h0 is dead after h1 was produced
    op_gemm(u, a2, t.o2w, t.o2b, true, h1, M, C, C, h2, 0, true, true);
    // feed forward (GEGLU fused into the first GEMM's epilogue)
    op_ln(u, h2, t.l3g, t.l3b, M, C, l);
    f16 *ff = u->allocH((size_t)M * 4 * C);
    op_gemm(u, l, t.f1w, t.f1b, true, nullptr, M, 8 * C, C, ff, 1);
    void *h3 = h1;
    op_gemm(u, ff, t.f2w, t.f2b, true, h2, M, C, 4 * C, h3, 0, true, true);
    op_gemm(u, op_as16(u, h3, (size_t)M * C), t.pow_, t.pob, true, x, M, C, C, out, 0, true, true);
    u->top = mark;
    return out;
}

static int unet_run(ctx_unet *u, const float *sample, const float *timestep, const float *ctx, int B, int H, int W, int L, float *out)
{
    const ctx_unet_config_t &c = u->cfg;
    const int n = c.n_levels, lpb = c.layers_per_block;
    const int *ch = c.block_out_channels;
    u->top = 0; u->peak = 0; u->rc = 0;
    u->ref_k = 0; u->ref_cursor = 0;
    u->taps.clear(); u->tap_cursor = 0;
    if (u->ref_mode == 1) u->ref_slots.clear();
    for (int k = 0; k < 3; ++k) { u->launches[k] = 0; u->flops[k] = 0; }
    FwdCtx f; f.B = B; f.L = L;
    f.gn_stats = u->alloc((size_t)ctx_groupnorm_ws_bytes(B, c.groups));
    // time embedding
    f16 *te0 = u->allocH((size_t)B * ch[0]);
    note(u, 2, 0); RUN(ctx_time_embed_f16(timestep, B, ch[0], te0, u->s));
    f16 *te1 = u->allocH((size_t)B * u->temb_dim);
    note(u, 2, 2.0 * B * ch[0] * u->temb_dim); RUN(ctx_gemv_f16(te0, u->W + u->t1w, u->W + u->t1b, B, u->temb_dim, ch[0], 0, 1, te1, u->s));
    f16 *te2 = u->allocH((size_t)B * u->temb_dim);
    note(u, 2, 2.0 * B * u->temb_dim * u->temb_dim); RUN(ctx_gemv_f16(te1, u->W + u->t2w, u->W + u->t2b, B, u->temb_dim, u->temb_dim, 0, 1, te2, u->s));   // + the resnets' SiLU(temb), once
    f16 *tproj = u->allocH((size_t)B * u->temb_rows);
    note(u, 2, 2.0 * B * u->temb_rows * u->temb_dim); RUN(ctx_gemv_f16(te2, u->W + u->tpw, u->W + u->tpb, B, u->temb_rows, u->temb_dim, 0, 0, tproj, u->s));
    f.tproj = tproj;
    f16 *ctx16 = u->allocH((size_t)B * L * c.cross_attention_dim);
    note(u, 2, 0); RUN(ctx_f32_to_f16(ctx, (int64_t)B * L * c.cross_attention_dim, ctx16, u->s));
    f.ctx16 = ctx16;
    // cross-attention K/V of every transformer block in one GEMM: they depend only on the context
    f16 *kv_all = u->allocH((size_t)B * L * u->kv_rows_total);
    op_gemm(u, ctx16, u->kvw, 0, false, nullptr, B * L, u->kv_rows_total, c.cross_attention_dim, kv_all);
    f.kv_all = kv_all;

    int h = H, w = W;
    if (u->res32 && (u->is_controlnet || u->add_res || u->ref_mode)) {
        ctx_set_error("unet: the fp32 residual-stream mode covers the plain UNet forward only");
        return CTX_E_STATE;
    }
    void *x = u->allocS((size_t)B * h * w * ch[0]);
    {
        size_t m0 = u->top;
        f16 *x16 = u->res32 ? u->allocH((size_t)B * h * w * ch[0]) : (f16 *)x;
        note(u, 2, 2.0 * B * h * w * ch[0] * c.in_channels * 9); RUN(ctx_conv_in_f16(sample, u->W + u->ciw, u->W + u->cib, B, c.in_channels, h, w, ch[0], x16, u->s));
        if (u->res32) { note(u, 2, 0); RUN(ctx_f16_to_f32(x16, (int64_t)B * h * w * ch[0], (float *)x, u->s)); u->top = m0; }
    }

    auto add_scaled = [&](f16 *dst, const f16 *src, float scale, size_t n) {
        note(u, 2, 0);
        if (!u->dry && u->rc == 0)
            hipLaunchKernelGGL(k_add_scaled_f16, dim3((unsigned)std::min<int64_t>(cdiv64((int64_t)n / 8, 256), 4096)), dim3(256), 0, u->s, dst, src,
                               scale, (int64_t)n / 8);
    };
    if (u->is_controlnet) {
        // sample = conv_in(sample) + controlnet_cond_embedding(cond): the conditioning image is 8x the latent grid
        size_t mark = u->top;
        int eh = 8 * H, ew = 8 * W;
        const f16 *cur16 = nullptr;
        const size_t nconv = u->cond_convs.size();
        for (size_t k = 0; k + 1 < nconv; ++k) {
            const ctx_unet::CondConv &cc = u->cond_convs[k];
            int oh = (eh - 1) / cc.stride + 1, ow = (ew - 1) / cc.stride + 1;
            const bool last = k + 2 == nconv;                   // its output is what the cache holds
            f16 *o = (last && u->cn_cache) ? u->cn_cache : u->allocH((size_t)B * oh * ow * cc.cout);
            if (!u->cn_cache_valid) {
                note(u, 2, 2.0 * B * oh * ow * cc.cout * cc.cin * 9);
                if (!u->dry && u->rc == 0) {
                    int64_t items = (int64_t)B * oh * ow * (cc.cout / 8);
                    hipLaunchKernelGGL(k_conv_small, dim3((unsigned)std::min<int64_t>(cdiv64(items, 256), 65535)), dim3(256), 0, u->s,
                                       k == 0 ? u->cn_cond : nullptr, cur16, u->W + cc.w, u->W + cc.b, B, eh, ew, cc.cin, cc.cout, cc.stride, 1, o);
                }
            }
            cur16 = o; eh = oh; ew = ow;
        }
        if (eh != h || ew != w) { ctx_set_error("controlnet: conditioning embedding grid %dx%d != latent grid %dx%d", eh, ew, h, w); return CTX_E_ARG; }
        const ctx_unet::CondConv &co = u->cond_convs.back();
        f16 *emb = u->allocH((size_t)B * h * w * ch[0]);
        op_conv(u, cur16, co.w, co.b, nullptr, 0, x, B, h, w, co.cin, ch[0], 1, 0, emb);       // + conv_in(sample) as the residual operand
        if (!u->dry && u->rc == 0) (void)hipMemcpyAsync(x, emb, (size_t)B * h * w * ch[0] * 2, hipMemcpyDeviceToDevice, u->s);
        u->top = mark;
    }
    struct Skip { void *p; int C, h, w; };
    std::vector<Skip> skips;
    skips.push_back({x, ch[0], h, w});
    u->tap(x, B * h * w, ch[0]);
    int cur = ch[0];
    for (int i = 0; i < n; ++i) {
        LevelP &Lv = u->down[i];
        for (int j = 0; j < lpb; ++j) {
            int cout = Lv.res[j].cout;
            void *o = u->allocS((size_t)B * h * w * cout);
            if (Lv.has_attn) {
                size_t mark = u->top;
                void *t = u->allocS((size_t)B * h * w * cout);
                run_resnet(u, f, Lv.res[j], x, h, w, t);
                u->tap(t, B * h * w, cout);
                run_transformer(u, f, Lv.tr[j], t, h, w, o);
                u->top = mark;
            } else run_resnet(u, f, Lv.res[j], x, h, w, o);
            u->tap(o, B * h * w, cout);
            x = o; cur = cout;
            skips.push_back({x, cur, h, w});
        }
        if (Lv.has_sampler) {
            int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
            void *o = u->allocS((size_t)B * ho * wo * cur);
            size_t m0 = u->top;
            op_conv(u, op_as16(u, x, (size_t)B * h * w * cur), Lv.sw, Lv.sb, nullptr, 0, nullptr, B, h, w, cur, cur, 2, 0, o, false, true);
            u->top = m0;
            x = o; h = ho; w = wo;
            u->tap(x, B * h * w, cur);
            skips.push_back({x, cur, h, w});
        }
    }
    size_t res_off = 0;                        // running element offset into the residual buffers (skip order, then mid)
    if (u->is_controlnet) {
        // zero convolutions of the skip tensors -> residuals (the conditioning scale is applied where they are added)
        if (skips.size() != u->zc_w.size()) { ctx_set_error("controlnet: %zu skip tensors but %zu zero convolutions", skips.size(), u->zc_w.size()); return CTX_E_STATE; }
        for (size_t k = 0; k < skips.size(); ++k) {
            const Skip &sk = skips[k];
            const size_t nel = (size_t)B * sk.h * sk.w * sk.C;
            op_gemm(u, (const f16 *)sk.p, u->zc_w[k], u->zc_b[k], true, nullptr, B * sk.h * sk.w, sk.C, sk.C, u->cn_out ? u->cn_out + res_off : nullptr);
            res_off += nel;
        }
    } else if (u->add_res) {
        // down_block_additional_residuals: added to the skip copies once the down path has consumed the originals
        for (size_t k = 0; k < skips.size(); ++k) {
            Skip &sk = skips[k];
            const size_t nel = (size_t)B * sk.h * sk.w * sk.C;
            if (sk.p == x) {                   // the last skip tensor is also the mid block's input, which stays as it is
                f16 *cp = u->allocH(nel);
                if (!u->dry && u->rc == 0) (void)hipMemcpyAsync(cp, sk.p, nel * 2, hipMemcpyDeviceToDevice, u->s);
                sk.p = cp;
            }
            add_scaled((f16 *)sk.p, u->add_res + res_off, u->add_scale, nel);
            res_off += nel;
        }
    }
    {
        void *o1 = u->allocS((size_t)B * h * w * cur);
        run_resnet(u, f, u->mid.res[0], x, h, w, o1);
        u->tap(o1, B * h * w, cur);
        void *o2 = u->allocS((size_t)B * h * w * cur);
        run_transformer(u, f, u->mid.tr[0], o1, h, w, o2);
        u->tap(o2, B * h * w, cur);
        void *o3 = u->allocS((size_t)B * h * w * cur);
        run_resnet(u, f, u->mid.res[1], o2, h, w, o3);
        u->tap(o3, B * h * w, cur);
        x = o3;
    }
    if (u->is_controlnet) {
        op_gemm(u, (const f16 *)x, u->zm_w, u->zm_b, true, nullptr, B * h * w, cur, cur, u->cn_out ? u->cn_out + res_off : nullptr);
        res_off += (size_t)B * h * w * cur;
        u->ref_cursor = res_off;               // element count of the residual buffer (read by the size query)
        if (!u->dry && u->rc == 0) {
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) { ctx_set_error("controlnet_forward: launch failed: %s", hipGetErrorString(e)); return CTX_E_LAUNCH; }
        }
        return u->rc;
    }
    if (u->add_res) add_scaled((f16 *)x, u->add_res + res_off, u->add_scale, (size_t)B * h * w * cur);
    for (int i = 0; i < n; ++i) {
        LevelP &Lv = u->up[i];
        for (int j = 0; j <= lpb; ++j) {
            Skip sk = skips.back(); skips.pop_back();
            if (sk.h != h || sk.w != w) { ctx_set_error("unet: skip size mismatch (H, W must be multiples of %d)", 1 << (n - 1)); return CTX_E_ARG; }
            int cin = cur + sk.C, cout = Lv.res[j].cout;
            void *o = u->allocS((size_t)B * h * w * cout);
            size_t mark = u->top;
            void *cat = u->allocS((size_t)B * h * w * cin);
            note(u, 2, 0);
            if (u->res32) RUN(ctx_concat_f32((const float *)x, (const float *)sk.p, (int64_t)B * h * w, cur, sk.C, (float *)cat, u->s));
            else RUN(ctx_concat_f16((const f16 *)x, (const f16 *)sk.p, (int64_t)B * h * w, cur, sk.C, (f16 *)cat, u->s));
            if (Lv.has_attn) {
                void *t = u->allocS((size_t)B * h * w * cout);
                run_resnet(u, f, Lv.res[j], cat, h, w, t);
                u->tap(t, B * h * w, cout);
                run_transformer(u, f, Lv.tr[j], t, h, w, o);
            } else run_resnet(u, f, Lv.res[j], cat, h, w, o);
            u->tap(o, B * h * w, cout);
            u->top = mark;
            x = o; cur = cout;
        }
        if (Lv.has_sampler) {
            void *o = u->allocS((size_t)B * (2 * h) * (2 * w) * cur);
            size_t m0 = u->top;
            op_conv(u, op_as16(u, x, (size_t)B * h * w * cur), Lv.sw, Lv.sb, nullptr, 0, nullptr, B, h, w, cur, cur, 1, 1, o, false, true);
            u->top = m0;
            x = o; h *= 2; w *= 2;
            u->tap(x, B * h * w, cur);
        }
    }
    f16 *y = u->allocH((size_t)B * h * w * cur);
    op_gn(u, x, u->cng, u->cnb, B, h * w, cur, c.norm_eps, 1, y, f.gn_stats);
    note(u, 2, 2.0 * B * h * w * cur * c.out_channels * 9);
    RUN(ctx_conv_out_f16(y, u->W + u->cow, u->W + u->cob, B, h, w, cur, c.out_channels, out, u->s));
    if (!u->dry && u->rc == 0) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { ctx_set_error("unet_forward: launch failed: %s", hipGetErrorString(e)); return CTX_E_LAUNCH; }
    }
    return u->rc;
}

static int check_dims(const ctx_unet *u, int B, int H, int W, int L)
{
    int div = 1 << (u->cfg.n_levels - 1);
    if (B < 1 || B > 16 || H < div || W < div || H % div || W % div || L < 1) {
        ctx_set_error("unet: need 1<=B<=16, H,W multiples of %d, ctx_len>=1 (B=%d H=%d W=%d L=%d)", div, B, H, W, L);
        return CTX_E_ARG;
    }
    return 0;
}

extern "C" int64_t ctx_unet_workspace_bytes(const ctx_unet_t *cu, int32_t B, int32_t H, int32_t W, int32_t ctx_len)
{
    ctx_unet *u = const_cast<ctx_unet *>(cu);
    if (!u || check_dims(u, B, H, W, ctx_len)) return -1;
    bool was = u->dry;
    u->dry = true;
    unet_run(u, nullptr, nullptr, nullptr, B, H, W, ctx_len, nullptr);
    u->dry = was;
    return (int64_t)u->peak + 4096;
}

extern "C" int32_t ctx_unet_forward(ctx_unet_t *u, const float *sample, const float *timestep, const float *ctx, int32_t B,
                                    int32_t H, int32_t W, int32_t ctx_len, float *out, ctx_stream_t stream)
{
    CTX_REQUIRE(u && sample && timestep && ctx && out, "unet_forward: null pointer");
    CTX_REQUIRE(u->W && u->ws, "unet_forward: ctx_unet_bind() first");
    if (check_dims(u, B, H, W, ctx_len)) return CTX_E_ARG;
    u->s = (hipStream_t)stream;
    u->dry = false;
    return unet_run(u, sample, timestep, ctx, B, H, W, ctx_len, out);
}

// ---- ControlNet -----------------------------------------------------------------------------------------------------------
extern "C" int64_t ctx_controlnet_residual_bytes(const ctx_unet_t *cu, int32_t B, int32_t H, int32_t W)
{
    ctx_unet *u = const_cast<ctx_unet *>(cu);
    if (!u || !u->is_controlnet || check_dims(u, B, H, W, 1)) return -1;
    bool was = u->dry;
    u->dry = true;
    int rc = unet_run(u, nullptr, nullptr, nullptr, B, H, W, 1, nullptr);
    u->dry = was;
    return rc ? -1 : (int64_t)u->ref_cursor * 2 + 256;
}

extern "C" int64_t ctx_controlnet_cond_cache_bytes(const ctx_unet_t *u, int32_t B, int32_t H, int32_t W)
{
    if (!u || !u->is_controlnet || B < 1 || H < 1 || W < 1 || u->cond_convs.size() < 2) return -1;
    return (int64_t)B * H * W * u->cond_convs[u->cond_convs.size() - 2].cout * 2 + 256;
}

extern "C" int32_t ctx_controlnet_forward(ctx_unet_t *u, const float *sample, const float *timestep, const float *ctx, const float *cond,
                                          void *cond_cache, int32_t cache_valid, int32_t B, int32_t H, int32_t W, int32_t ctx_len,
                                          void *residuals, ctx_stream_t stream)
{
    CTX_REQUIRE(u && sample && timestep && ctx && cond && residuals, "controlnet_forward: null pointer");
    CTX_REQUIRE(!cache_valid || cond_cache, "controlnet_forward: cache_valid without a cond_cache buffer");
    CTX_REQUIRE(u->is_controlnet, "controlnet_forward: the handle is a UNet (create it with ctx_controlnet_create)");
    CTX_REQUIRE(u->W && u->ws, "controlnet_forward: ctx_unet_bind() first");
    if (check_dims(u, B, H, W, ctx_len)) return CTX_E_ARG;
    u->s = (hipStream_t)stream; u->dry = false;
    u->cn_cond = cond; u->cn_out = (f16 *)residuals; u->cn_cache = (f16 *)cond_cache; u->cn_cache_valid = cache_valid != 0;
    int rc = unet_run(u, sample, timestep, ctx, B, H, W, ctx_len, nullptr);
    u->cn_cond = nullptr; u->cn_out = nullptr; u->cn_cache = nullptr; u->cn_cache_valid = false;
    return rc;
}

extern "C" int32_t ctx_unet_set_residuals(ctx_unet_t *u, const void *residuals, float scale)
{
    CTX_REQUIRE(u && !u->is_controlnet, "unet_set_residuals: need a UNet handle");
    u->add_res = (const f16 *)residuals; u->add_scale = scale;
    return CTX_OK;
}

// ---- reference-only attention passes ----------------------------------------------------------------------------------
extern "C" int64_t ctx_unet_ref_bank_bytes(const ctx_unet_t *cu, int32_t B, int32_t H, int32_t W)
{
    ctx_unet *u = const_cast<ctx_unet *>(cu);
    if (!u || check_dims(u, B, H, W, 1)) return -1;
    std::vector<ctx_unet::RefSlot> keep = u->ref_slots;
    bool was = u->dry; int mode = u->ref_mode;
    u->dry = true; u->ref_mode = 1;
    unet_run(u, nullptr, nullptr, nullptr, B, H, W, 1, nullptr);
    int64_t n = (int64_t)u->ref_cursor * 2 + 256;
    u->dry = was; u->ref_mode = mode; u->ref_slots = keep;
    return n;
}

extern "C" int64_t ctx_unet_workspace_bytes_ref(const ctx_unet_t *cu, int32_t B, int32_t H, int32_t W, int32_t ctx_len, int32_t mode,
                                                int32_t ref_row0)
{
    ctx_unet *u = const_cast<ctx_unet *>(cu);
    if (!u || check_dims(u, B, H, W, ctx_len) || mode < 1 || mode > 2 || ref_row0 < 0 || ref_row0 >= B) return -1;
    std::vector<ctx_unet::RefSlot> keep = u->ref_slots;
    bool was = u->dry; int m0 = u->ref_mode, r0 = u->ref_row0;
    u->dry = true; u->ref_mode = mode; u->ref_row0 = ref_row0;
    int rc = unet_run(u, nullptr, nullptr, nullptr, B, H, W, ctx_len, nullptr);
    int64_t n = rc ? -1 : (int64_t)u->peak + 4096;
    u->dry = was; u->ref_mode = m0; u->ref_row0 = r0;
    if (mode == 1) u->ref_slots = keep;
    return n;
}

extern "C" int32_t ctx_unet_forward_ref(ctx_unet_t *u, const float *sample, const float *timestep, const float *ctx, int32_t B,
                                        int32_t H, int32_t W, int32_t ctx_len, int32_t mode, void *bank, int32_t ref_row0, float *out,
                                        ctx_stream_t stream)
{
    CTX_REQUIRE(u && sample && timestep && ctx && out && bank, "unet_forward_ref: null pointer");
    CTX_REQUIRE(u->W && u->ws, "unet_forward_ref: ctx_unet_bind() first");
    CTX_REQUIRE(mode == 1 || mode == 2, "unet_forward_ref: mode %d (1 = 'w' park the attn1 inputs, 2 = 'r' attend to them)", mode);
    CTX_REQUIRE(ref_row0 >= 0 && ref_row0 < B, "unet_forward_ref: ref_row0=%d outside [0, B)", ref_row0);
    if (check_dims(u, B, H, W, ctx_len)) return CTX_E_ARG;
    u->s = (hipStream_t)stream;
    u->dry = false;
    u->ref_mode = mode; u->ref_bank = (f16 *)bank; u->ref_row0 = mode == 2 ? ref_row0 : 0;
    int rc = unet_run(u, sample, timestep, ctx, B, H, W, ctx_len, out);
    u->ref_mode = 0; u->ref_bank = nullptr; u->ref_row0 = 0;
    return rc;
}

/* Experiment switch: 1 = keep the residual stream (block outputs, skip tensors, the transformer's running
   sums) in fp32 instead of fp16; GEMM / conv operands and weights stay fp16.  Plain UNet forward only. */
extern "C" int32_t ctx_unet_set_residual_fp32(ctx_unet_t *u, int32_t on)
{
    CTX_REQUIRE(u, "unet_set_residual_fp32: null handle");
    u->res32 = on != 0;
    return CTX_OK;
}

/* Measurement aid: buf (fp16 device memory of `capacity` elements, or NULL to switch off) receives a copy of every block's output
   of the following forwards, in execution order (conv_in; per level resnet, transformer, sampler; mid; up path).  After a forward,
   ctx_unet_tap_count / ctx_unet_tap_info describe what was (or would have been: offsets past the capacity are not written) copied. */
extern "C" int32_t ctx_unet_set_taps(ctx_unet_t *u, void *buf, int64_t capacity)
{
    CTX_REQUIRE(u && capacity >= 0, "unet_set_taps: bad args");
    u->tap_buf = (f16 *)buf; u->tap_cap = buf ? (size_t)capacity : 0;
    return CTX_OK;
}
extern "C" int32_t ctx_unet_tap_count(const ctx_unet_t *u) { return u ? (int32_t)u->taps.size() : -1; }
extern "C" int32_t ctx_unet_tap_info(const ctx_unet_t *u, int32_t i, int64_t *offset, int32_t *rows, int32_t *channels)
{
    CTX_REQUIRE(u && offset && rows && channels && i >= 0 && i < (int32_t)u->taps.size(), "unet_tap_info: bad args");
    *offset = (int64_t)u->taps[i].off; *rows = u->taps[i].rows; *channels = u->taps[i].C;
    return CTX_OK;
}

extern "C" int32_t ctx_unet_stats(const ctx_unet_t *u, int32_t klass, int64_t *launches, double *flops)
{
    CTX_REQUIRE(u && klass >= 0 && klass < 3 && launches && flops, "unet_stats: bad args");
    *launches = u->launches[klass];
    *flops = u->flops[klass];
    return CTX_OK;
}
