/*
 * ctx_nerf.h — C-ABI of libctxnerf.so: the MI355X (gfx950) hot path of ConTEXTure's per-view
 * texture-painting loop.
 *
 * The reference (zaiisao/ConTEXTure-NeRF) has no FFI of its own: its hot ops live in third-party
 * Python packages (kaolin, torch-scatter, diffusers).  Each entry point below replaces the
 * third-party call the reference makes at the cited file:line; the Python shims in
 * contexture-nerf_amd/ keep those call signatures (INTEGRATION.md shows the ctypes binding).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous row-major memory owned by the caller
 *     (PyTorch allocates; the library never frees or retains it beyond the call, except the
 *     weight/workspace blobs explicitly bound to a ctx_unet_t);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing syncs;
 *   - return 0 = OK, negative = error (CTX_E_*); ctx_last_error() gives a thread-local message;
 *   - int64 face indices follow the reference's dtype (kaolin returns int64, -1 = background).
 */
#ifndef CTX_NERF_H
#define CTX_NERF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CTX_OK 0
#define CTX_E_ARG (-1)      /* bad shape / null pointer / unsupported size */
#define CTX_E_LAUNCH (-2)   /* hipGetLastError() after launch */
#define CTX_E_STATE (-3)    /* handle not bound / wrong phase */

typedef void *ctx_stream_t;

int32_t ctx_version(void);
const char *ctx_last_error(void);
/* Device check used by the Python loader: 0 when a gfx950 device is current. */
int32_t ctx_device_check(void);

/* ---- raster path ------------------------------------------------------------------------- */
/* kal.render.mesh.prepare_vertices  (src/models/render.py:112-113; textured_mesh.py:167-168)
   verts[B,V,3] faces[F,3] cam[B,4,3] proj3[3] -> fv_cam[B,F,3,3] fv_img[B,F,3,2] fnorm[B,F,3] */
int32_t ctx_prepare_vertices(const float *verts, const int64_t *faces, const float *cam,
                             const float *proj3, int32_t B, int32_t V, int32_t F,
                             float *fv_cam, float *fv_img, float *fnorm,
                             void *ws /* B*V*5 floats */, ctx_stream_t stream);

/* Scratch for the per-coarse-tile face lists of the two rasterise entry points. */
int64_t ctx_rasterize_ws_bytes(int32_t H, int32_t W, int32_t B, int32_t F);
/* kal.render.mesh.rasterize  (src/models/render.py:115-120; textured_mesh.py:170-175)
   face_z[B,F,3] face_xy[B,F,3,2] feat[B,F,3,C] -> out[B,H,W,C] face_idx[B,H,W]  (C <= 64) */
int32_t ctx_rasterize_fwd(int32_t H, int32_t W, const float *face_z, const float *face_xy,
                          const float *feat, int32_t B, int32_t F, int32_t C,
                          float multiplier, float eps, float *out, int64_t *face_idx,
                          void *ws, int64_t ws_bytes, ctx_stream_t stream);

/* The reference rasterises twice per render (depth pass, then UV pass: render.py:115 and :119).
   Fused single pass: depth[B,H,W] (feature = z), uv[B,H,W,2], face_idx[B,H,W], and optionally the
   gathered face normals normals[B,H,W,3] (render.py:150-157; background reads the LAST face).
   fv_cam[B,F,3,3] supplies z (= fv_cam[...,2]); uv_attr[Bu,F,3,2] with Bu in {1,B}. */
int32_t ctx_rasterize_fused(int32_t H, int32_t W, const float *fv_cam, const float *face_xy,
                            const float *uv_attr, int32_t Bu, const float *fnorm /*nullable*/,
                            int32_t B, int32_t F, float multiplier, float eps,
                            float *depth, float *uv, int64_t *face_idx, float *normals /*nullable*/,
                            void *ws, int64_t ws_bytes, ctx_stream_t stream);

/* Renderer.normalize_multiple_depth (src/models/render.py:48-74).  ws: ctx_normalize_depth_ws_bytes(B).
   status[0] (device int32, nullable): 0 ok, 1 positive depth present, 2 all-zero (the reference's
   two asserts, render.py:49-50). */
int64_t ctx_normalize_depth_ws_bytes(int32_t B);
int32_t ctx_normalize_depth(const float *depth, int32_t B, int32_t HW, float *out,
                            void *ws, int32_t *status, ctx_stream_t stream);

/* kal.render.mesh.texture_mapping (src/models/render.py:135) == grid_sample(align_corners=False,
   padding 'border'); tex[Bt,C,T,T] with Bt in {1,B}; mode 0 bilinear, 1 nearest; out[B,HW,C].
   mask_idx (nullable, int64[B,HW]): when given, out *= (face_idx > -1)   (render.py:133,141). */
int32_t ctx_texture_mapping_fwd(const float *uv, const float *tex, int32_t B, int32_t HW,
                                int32_t C, int32_t T, int32_t Bt, int32_t mode,
                                const int64_t *mask_idx, float *out, ctx_stream_t stream);
/* autograd of the above w.r.t. the (expanded) atlas: grad_tex[C,T,T] += ...  (caller zeroes). */
int32_t ctx_texture_mapping_bwd(const float *grad_out, const float *uv, int32_t B, int32_t HW,
                                int32_t C, int32_t T, const int64_t *mask_idx, float *grad_tex,
                                ctx_stream_t stream);
/* The same scatter without global float atomics (uvscatter.hip): pixels are binned by 32 x 32-texel atlas tile once per raster
   (`plan`: depends on uv / mask_idx only, reusable by every backward of the SDS loop), then one workgroup per tile accumulates
   its pixel list in LDS as fixed-point int64 sums and writes each texel once.  Bit-reproducible (integer sums do not depend
   on arrival order); C <= 4, B*HW < 2^32, T <= ctx_texmap_plan_max_res().  grad_tex [C,T,T] is added to (as above).
   The fixed-point unit is chosen per call from max|grad_out| (2^-E of the largest tap, E = 62 - ceil(log2(B*HW))), so gradients of
   any magnitude keep the same relative resolution; a non-finite grad_out, or a (uv, mask_idx) that no longer matches the plan's
   sampled checksum, turns the whole of grad_tex into NaN (and ctx_texmap_plan_stale() reports the latter).
   ws: ctx_texture_mapping_bwd_binned_ws_bytes. */
int64_t ctx_texmap_bwd_plan_bytes(int32_t B, int32_t HW, int32_t T);
int32_t ctx_texmap_plan_max_res(void);
int32_t ctx_texmap_bwd_plan(const float *uv, const int64_t *mask_idx, int32_t B, int32_t HW, int32_t T, void *plan, ctx_stream_t stream);
int32_t ctx_texmap_plan_stale(const void *plan, ctx_stream_t stream);
int64_t ctx_texture_mapping_bwd_binned_ws_bytes(int32_t C, int32_t T);
int32_t ctx_texture_mapping_bwd_binned(const float *grad_out, const float *uv, const int64_t *mask_idx, int32_t B, int32_t HW, int32_t C, int32_t T,
                                       const void *plan, void *ws, float *grad_tex, ctx_stream_t stream);

/* UV back-projection of painted views (north_star "torch-scatter UV back-projection"; call contract src/training/trainer.py:1076-1090)
   as INTEGER sums: acc [C,T,T] int64 += round(values * bilinear weight * 2^frac_bits) at the 4 texels of every unmasked pixel.
   The caller owns acc across calls and ranks: view shards are all-reduced (SUM, int64) and converted once by ctx_fixed_to_float,
   so the N-rank atlas equals the 1-rank atlas bit for bit.  Requires |values| * B*HW * 2^frac_bits < 2^63.
   plan: as above, or NULL (any T, any C: one global int64 atomic per tap). */
int32_t ctx_uv_scatter_fixed(const float *values, const float *uv, const int64_t *mask_idx, int32_t B, int32_t HW, int32_t C, int32_t T,
                             const void *plan, int32_t frac_bits, int64_t *acc, ctx_stream_t stream);
int32_t ctx_fixed_to_float(const int64_t *acc, int64_t n, int32_t frac_bits, int32_t accumulate, float *out, ctx_stream_t stream);

/* Texel-interleaved forward for C <= 4 and one texture shared by the batch (the reference's texture_img.expand(B, ...),
   render.py:133-135): ctx_texture_pack4 repacks [C,T,T] into [T,T,4] once, ctx_texture_mapping_packed_fwd then gathers one
   16-byte texel per bilinear tap.  Results are bit-identical to ctx_texture_mapping_fwd. */
int32_t ctx_texture_pack4(const float *tex, int32_t C, int32_t T, float *packed, ctx_stream_t stream);
int32_t ctx_texture_mapping_packed_fwd(const float *uv, const float *packed, int32_t B, int32_t HW, int32_t C, int32_t T,
                                       int32_t mode, const int64_t *mask_idx, float *out, ctx_stream_t stream);

/* ---- view weights: torch_scatter.scatter_max seam (src/training/trainer.py:213-249) ------- */
/* phase 0: max_z[f] = max(max_z[f], fnz[b,f]) over pixels of the B local views showing f.
   Caller pre-fills max_z with -inf; between the phases a multi-GPU caller all-reduces(MAX). */
int32_t ctx_view_weights_max(const int64_t *face_idx, const float *fnz, int32_t B, int32_t HW,
                             int32_t F, float *max_z, void *vis_ws /* B*F bytes */, ctx_stream_t stream);
/* phase 1: mask[b,p] = !(fnz[b,f] < max_z[f]) for f>=0, 1 for background. */
int32_t ctx_view_weights_mask(const int64_t *face_idx, const float *fnz, const float *max_z,
                              int32_t B, int32_t HW, int32_t F, uint8_t *mask, ctx_stream_t stream);
/* ConTEXTure.create_face_view_map (trainer.py:155-211): rows (face,view,i,j) of valid pixels in
   (view, row, col) order.  ws: ctx_face_view_map_ws_bytes.  n_rows: device int64[1]. */
int64_t ctx_face_view_map_ws_bytes(int32_t B, int32_t H, int32_t W);
int32_t ctx_face_view_map(const int64_t *face_idx, int32_t B, int32_t H, int32_t W,
                          int64_t *rows /*[B*H*W,4] capacity*/, int64_t *n_rows, void *ws,
                          ctx_stream_t stream);

/* ---- texture field: get_embedder / NeRF2D (src/run_nerf_helpers.py:15-135) ---------------- */
/* embed(x[N,d]) -> [N, d*(1+2L)], order [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...]. */
int32_t ctx_embed_fwd(const float *x, int64_t N, int32_t d, int32_t L, float *out, ctx_stream_t stream);
/* Bytes of the packed-weight blob for NeRF2D(D,W,input_ch,output_ch,skip). */
int64_t ctx_uvmlp_packed_bytes(int32_t D, int32_t W, int32_t input_ch, int32_t output_ch, int32_t skip);
/* Pack nn.Linear weights: ws[i] -> device float [out_i,in_i], bs[i] -> device float [out_i];
   index D is output_linear. */
int32_t ctx_uvmlp_pack(const float *const *ws, const float *const *bs, int32_t D, int32_t W,
                       int32_t input_ch, int32_t output_ch, int32_t skip, void *packed,
                       ctx_stream_t stream);
/* Fused  embed(uv) -> NeRF2D -> raw[N,3]  (+ optional tex_chw[3,N] = (tanh(raw)+1)/2 laid out as the
   [1,3,res,res] atlas of textured_mesh.py:298-301).  uv nullable: then uv = the res x res 'xy'
   meshgrid of linspace(0,1,res) (textured_mesh.py:269-273), N = res*res. */
/* emb (nullable): precomputed embedding [N, 2*(1+2L)] as returned by embed(); when given it is
   loaded instead of being recomputed from uv (keeps the NeRF2D.forward(embedded) seam). */
int32_t ctx_uvmlp_fwd(const float *uv /*nullable*/, const float *emb /*nullable*/, int64_t N, int32_t res, const void *packed,
                      int32_t D, int32_t W, int32_t L, int32_t output_ch, int32_t skip,
                      float *raw, float *tex_chw /*nullable*/, ctx_stream_t stream);

/* General / training forward.  dims = 2 (uv, as ctx_uvmlp_fwd) or 3 (xyz sample points of the ray path: the reference's
   NeRF2D defaults, input_ch = 3*(1+2L) <= 64, output_ch = 4 = rgb + sigma); `uv` is then the [N,dims] point list.
   saved (nullable: plain forward) keeps what the backward needs (ctx_uvmlp_saved_bytes(N,D,W,input_ch) bytes: the padded
   embedding [N,48|64], the post-ReLU activations [D,N,W] fp32 and the ReLU pattern as bit masks). */
int64_t ctx_uvmlp_saved_bytes(int64_t N, int32_t D, int32_t W, int32_t input_ch);
int32_t ctx_uvmlp_fwd_save(const float *uv /*nullable*/, const float *emb /*nullable*/, int64_t N, int32_t res, const void *packed,
                           int32_t D, int32_t W, int32_t dims, int32_t L, int32_t output_ch, int32_t skip,
                           float *raw, float *tex_chw /*nullable*/, void *saved /*nullable: plain forward*/, ctx_stream_t stream);
/* Backward of NeRF2D.forward (autograd of src/run_nerf_helpers.py:106-135; the SDS loop src/training/trainer.py:644-907
   drives it with the atlas gradient).  Upstream gradient: grad_raw [N,output_ch] (d loss / d mlp_output) and / or
   grad_tex [output_ch,N] (d loss / d texture atlas of textured_mesh.py:298-301; the (tanh+1)/2 is differentiated
   here from `raw`).  Writes (not accumulates) d loss / d weight into gws[i] ([out_i,in_i], nn.Linear layout) and
   d loss / d bias into gbs[i], i = 0..D-1 hidden, D = output_linear (host arrays of device pointers).
   ws: ctx_uvmlp_bwd_ws_bytes(N,D,W) bytes of scratch.  Deterministic (fixed-order partial sums). */
int64_t ctx_uvmlp_bwd_ws_bytes(int64_t N, int32_t D, int32_t W);
int32_t ctx_uvmlp_bwd(const float *grad_raw /*nullable*/, const float *grad_tex /*nullable*/, const float *raw /*nullable w/o grad_tex*/,
                      int64_t N, const void *packed, int32_t D, int32_t W, int32_t dims, int32_t L, int32_t output_ch, int32_t skip,
                      const void *saved, void *ws, float *const *gws, float *const *gbs, ctx_stream_t stream);

/* ---- ray path (north_star; dead/absent in the reference) ------------------------ */
/* get_rays (run_nerf_helpers.py:139-148): K row-major [3,3] host floats passed by value fields. */
int32_t ctx_get_rays(int32_t H, int32_t W, float fx, float fy, float cx, float cy,
                     const float *c2w /*[3,4] device*/, float *rays_o, float *rays_d, ctx_stream_t stream);
/* nerf-pytorch raw2outputs: raw[R,S,4] z[R,S] rays_d[R,3] -> rgb[R,3] disp[R] acc[R]
   weights[R,S] (nullable) depth[R]. */
int32_t ctx_raymarch_composite_fwd(const float *raw, const float *z_vals, const float *rays_d,
                                   int64_t R, int32_t S, int32_t white_bkgd, float *rgb, float *disp,
                                   float *acc, float *weights, float *depth, ctx_stream_t stream);

/* ---- UNet denoise engine (src/stable_diffusion_depth.py:422-430,514) ----------------------- */
typedef struct ctx_unet ctx_unet_t;
typedef struct {
    int32_t in_channels, out_channels;
    int32_t n_levels;            /* <= 4 */
    int32_t block_out_channels[4];
    int32_t heads[4];
    int32_t down_attn[4], up_attn[4];
    int32_t layers_per_block;
    int32_t cross_attention_dim;
    int32_t groups;
    float norm_eps;
} ctx_unet_config_t;

ctx_unet_t *ctx_unet_create(const ctx_unet_config_t *cfg);
void ctx_unet_destroy(ctx_unet_t *u);
/* Parameter table in diffusers state_dict naming ("down_blocks.0.resnets.0.conv1.weight", ...). */
int32_t ctx_unet_param_count(const ctx_unet_t *u);
const char *ctx_unet_param_name(const ctx_unet_t *u, int32_t i);
int32_t ctx_unet_param_shape(const ctx_unet_t *u, int32_t i, int64_t shape4[4]); /* returns ndim */
int64_t ctx_unet_weight_bytes(const ctx_unet_t *u);
int64_t ctx_unet_workspace_bytes(const ctx_unet_t *u, int32_t B, int32_t H, int32_t W, int32_t ctx_len);
/* Bind caller-allocated blobs (256-B aligned). */
int32_t ctx_unet_bind(ctx_unet_t *u, void *weights, void *workspace, int64_t workspace_bytes);
/* Convert + repack parameter i from fp32 [diffusers layout] into the fp16 weight blob. */
int32_t ctx_unet_set_param(ctx_unet_t *u, int32_t i, const float *src, ctx_stream_t stream);
/* sample[B,Cin,H,W] f32 NCHW, timestep: device float[1] (graph-replay friendly), ctx[B,L,D] f32
   -> out[B,Cout,H,W] f32 NCHW.  Computes in fp16 with fp32 accumulation/statistics. */
int32_t ctx_unet_forward(ctx_unet_t *u, const float *sample, const float *timestep, const float *ctx,
                         int32_t B, int32_t H, int32_t W, int32_t ctx_len, float *out, ctx_stream_t stream);
/* Reference-only self-attention (Zero123++'s RefOnlyNoisedUNet / ReferenceOnlyAttnProc; the reference keeps the spec in
   src/zero123plus.py:127-237 and drives it from src/training/trainer.py:644-907).
   mode 1 ('w'): an ordinary forward over the noised condition latent that also parks every attn1 input (the LayerNorm-1 output,
                 [B, tokens, C] fp16 per layer) in `bank` (ctx_unet_ref_bank_bytes(u, B, H, W) bytes).
   mode 2 ('r'): forward whose attn1 layers use [own tokens ; parked tokens] as the K/V source for the batch rows >= ref_row0
                 (is_cfg_guidance => ref_row0 = 1: the unconditional row attends without the reference); row b reads the parked
                 row b - ref_row0, so the 'w' pass must have had B - ref_row0 rows.  The bank layout is private to the engine
                 that wrote it.  Workspace: ctx_unet_workspace_bytes_ref (for mode 2 call it after the 'w' pass). */
int64_t ctx_unet_ref_bank_bytes(const ctx_unet_t *u, int32_t B, int32_t H, int32_t W);
int64_t ctx_unet_workspace_bytes_ref(const ctx_unet_t *u, int32_t B, int32_t H, int32_t W, int32_t ctx_len, int32_t mode,
                                     int32_t ref_row0);
int32_t ctx_unet_forward_ref(ctx_unet_t *u, const float *sample, const float *timestep, const float *ctx,
                             int32_t B, int32_t H, int32_t W, int32_t ctx_len, int32_t mode, void *bank, int32_t ref_row0,
                             float *out, ctx_stream_t stream);
/* ControlNet (diffusers ControlNetModel.from_unet; Zero123++'s DepthControlUNet, spec in src/zero123plus.py:260-298, loaded at
   src/training/trainer.py:302-304): an engine with the UNet's conv_in / time embedding / down blocks / mid block, the
   conditioning-image embedding (channels 16-32-96-256, three stride-2 steps: the image is 8x the latent grid) and the 1x1 zero
   convolutions.  ctx_controlnet_forward writes the residuals (one per skip tensor, then the mid block; fp16, engine layout,
   ctx_controlnet_residual_bytes) — UNSCALED; ctx_unet_set_residuals hands them to a UNet engine of the same configuration, which
   adds `scale` x residual to its skip tensors and mid-block output in the following forwards (NULL switches it off).
   Parameter names are diffusers' ControlNetModel state_dict keys; the size / bind / set_param / destroy calls are ctx_unet_*. */
ctx_unet_t *ctx_controlnet_create(const ctx_unet_config_t *cfg, int32_t cond_channels);
int64_t ctx_controlnet_residual_bytes(const ctx_unet_t *cn, int32_t B, int32_t H, int32_t W);
/* cond_cache (nullable; ctx_controlnet_cond_cache_bytes): holds the output of the embedding's few-channel layers, which depends
   only on the conditioning image; pass cache_valid = 1 while the image is unchanged (every denoise step after the first). */
int64_t ctx_controlnet_cond_cache_bytes(const ctx_unet_t *cn, int32_t B, int32_t H, int32_t W);
int32_t ctx_controlnet_forward(ctx_unet_t *cn, const float *sample, const float *timestep, const float *ctx,
                               const float *cond /*[B,cond_channels,8H,8W] f32 NCHW*/, void *cond_cache, int32_t cache_valid,
                               int32_t B, int32_t H, int32_t W, int32_t ctx_len, void *residuals, ctx_stream_t stream);
int32_t ctx_unet_set_residuals(ctx_unet_t *u, const void *residuals /*nullable*/, float scale);
/* Per-kernel accounting of the last forward: number of launches and algorithmic FLOPs by class
   (0 gemm/conv MFMA, 1 attention MFMA, 2 other). */
/* Precision experiment: on != 0 keeps the UNet's residual stream (block outputs, skip tensors, the transformer blocks' running
   sums) in fp32; operands, weights and everything else stay as they are.  Plain forward only (not the ControlNet / reference-only
   passes).  The workspace grows: query ctx_unet_workspace_bytes again after switching. */
int32_t ctx_unet_set_residual_fp32(ctx_unet_t *u, int32_t on);
int32_t ctx_unet_stats(const ctx_unet_t *u, int32_t klass, int64_t *launches, double *flops);
/* Measurement aid (tools/precision_attribution.py; no reference counterpart): buf (fp16 device memory of `capacity` elements, NULL =
   off) receives a copy of every block's output (fp16 NHWC [rows, channels]) of the following plain forwards, in execution order:
   conv_in; per down level resnet (, transformer) x layers, downsampler; mid resnet, transformer, resnet; per up level resnet
   (, transformer) x (layers + 1), upsampler.  ctx_unet_tap_count / ctx_unet_tap_info describe the last forward's taps. */
int32_t ctx_unet_set_taps(ctx_unet_t *u, void *buf, int64_t capacity);
int32_t ctx_unet_tap_count(const ctx_unet_t *u);
int32_t ctx_unet_tap_info(const ctx_unet_t *u, int32_t i, int64_t *offset, int32_t *rows, int32_t *channels);

/* ---- VAE decoder (src/stable_diffusion_depth.py:976-990 decode_latents -> diffusers AutoencoderKL.decode) ---------- */
typedef struct ctx_vae ctx_vae_t;
typedef struct {
    int32_t latent_channels, out_channels;
    int32_t n_levels;                 /* <= 4 */
    int32_t block_out_channels[4];    /* encoder order, e.g. 128,256,512,512 */
    int32_t layers_per_block;
    int32_t groups;
} ctx_vae_config_t;
ctx_vae_t *ctx_vae_create(const ctx_vae_config_t *cfg);
void ctx_vae_destroy(ctx_vae_t *v);
int32_t ctx_vae_param_count(const ctx_vae_t *v);
const char *ctx_vae_param_name(const ctx_vae_t *v, int32_t i);      /* diffusers AutoencoderKL state_dict keys */
int32_t ctx_vae_param_shape(const ctx_vae_t *v, int32_t i, int64_t shape4[4]);
int64_t ctx_vae_weight_bytes(const ctx_vae_t *v);
int64_t ctx_vae_workspace_bytes(const ctx_vae_t *v, int32_t B, int32_t H, int32_t W);
int32_t ctx_vae_bind(ctx_vae_t *v, void *weights, void *workspace, int64_t workspace_bytes);
int32_t ctx_vae_set_param(ctx_vae_t *v, int32_t i, const float *src, ctx_stream_t stream);
/* latents [B,L,H,W] f32 (already divided by the 0.18215 scaling factor) -> image [B,3,8H,8W] f32 */
int32_t ctx_vae_decode(ctx_vae_t *v, const float *latents, int32_t B, int32_t H, int32_t W, float *image, ctx_stream_t stream);
/* AutoencoderKL.encode behind StableDiffusion.encode_imgs (src/stable_diffusion_depth.py:971-975): image f32 NCHW
   [B,3,H,W] (the caller applies 2x-1) -> moments f32 NCHW [B, 2*latent_channels, H/8, W/8] = quant_conv(Encoder(x)):
   mean | logvar of DiagonalGaussianDistribution; `.sample()` (mean + exp(0.5 clamp(logvar,-30,20)) * randn) and the
   0.18215 factor stay with the caller so that the noise comes from the caller's RNG stream.
   The parameter table lists post_quant_conv + decoder first (ctx_vae_decoder_param_count entries), then encoder + quant_conv. */
int32_t ctx_vae_decoder_param_count(const ctx_vae_t *v);
int64_t ctx_vae_encode_workspace_bytes(const ctx_vae_t *v, int32_t B, int32_t H, int32_t W);
int32_t ctx_vae_encode(ctx_vae_t *v, const float *image, int32_t B, int32_t H, int32_t W, float *moments, ctx_stream_t stream);
double ctx_vae_flops(const ctx_vae_t *v);     /* algorithmic FLOPs of the last decode / encode / dry run */
/* Autograd of `vae.encode` in the reference's SDS loop (`loss.backward()` reaches the texture through
   `vae.encode(rendered_grid_clean)`, src/training/trainer.py:732, 866; torch autograd over diffusers' Encoder there).
   ctx_vae_encode_train = ctx_vae_encode that keeps what the backward needs (every GroupNorm input and the attention's q|k|v)
   in the bound workspace; ctx_vae_encode_bwd consumes that tape once: grad_moments f32 NCHW [B, 2L, H/8, W/8] ->
   grad_image f32 NCHW [B,3,H,W] (input gradients only: the VAE's parameters are frozen on this path).  Internally the
   gradients are fp16 times `gscale` (choose it so that gscale * max|grad_moments| is O(1..100)); the result is exact in gscale.
   Any other call on the handle between the two drops the tape (ctx_vae_encode_bwd then fails with CTX_E_ARG). */
int64_t ctx_vae_encode_train_workspace_bytes(const ctx_vae_t *v, int32_t B, int32_t H, int32_t W);
int32_t ctx_vae_encode_train(ctx_vae_t *v, const float *image, int32_t B, int32_t H, int32_t W, float *moments, ctx_stream_t stream);
int32_t ctx_vae_encode_bwd(ctx_vae_t *v, const float *grad_moments, float gscale, float *grad_image, ctx_stream_t stream);

/* Building blocks, exported for unit parity tests (fp16 tensors passed as uint16 bit patterns). */
/* C[M,N] = A[M,K] @ Wt[N,K]^T (+bias[N]) (+residual[M,N]); K%64==0, N%8==0. */
int32_t ctx_gemm_f16(const void *A, const void *Wt, const void *bias, const void *residual,
                     int32_t M, int32_t N, int32_t K, void *C, ctx_stream_t stream);
/* x[B,H,W,Cin] (NHWC f16) * w[Cout,3,3,Cin] stride s pad 1 (+bias) (+rowbias[B,Cout]) (+res) -> [B,Ho,Wo,Cout];
   upsample=1: nearest x2 of x first (Upsample2D). */
int32_t ctx_conv3x3_f16(const void *x, const void *w, const void *bias, const void *rowbias,
                        const void *residual, int32_t B, int32_t H, int32_t W, int32_t Cin,
                        int32_t Cout, int32_t stride, int32_t upsample, void *y, ctx_stream_t stream);
/* GroupNorm(+SiLU) over NHWC f16; stats_ws: ctx_groupnorm_ws_bytes(B, groups). */
int64_t ctx_groupnorm_ws_bytes(int32_t B, int32_t groups);
int32_t ctx_groupnorm_f16(const void *x, const void *gamma, const void *beta, int32_t B, int32_t HW,
                          int32_t C, int32_t groups, float eps, int32_t silu, void *y, void *stats_ws,
                          ctx_stream_t stream);
int32_t ctx_layernorm_f16(const void *x, const void *gamma, const void *beta, int64_t rows, int32_t C,
                          float eps, void *y, ctx_stream_t stream);
/* softmax(Q K^T * scale) V; Q[B,Sq,heads*64], K[B,Skv,heads*64], V likewise (f16) -> O[B,Sq,heads*64];
   row strides in elements.  vt_ws: unused since V is consumed untransposed (ds_read_b64_tr_b16); may be null, ctx_attention_ws_bytes() returns a token size. */
int64_t ctx_attention_ws_bytes(int32_t B, int32_t Skv, int32_t heads);
int32_t ctx_attention_f16(const void *Q, const void *K, const void *V, int32_t B, int32_t Sq, int32_t Skv,
                          int32_t heads, int32_t q_stride, int32_t kv_stride, float scale, void *O,
                          int32_t o_stride, void *vt_ws, ctx_stream_t stream);
/* GEGLU: y[M,C4] = h[:, :C4] * gelu(h[:, C4:])  for h[M,2*C4] f16. */
int32_t ctx_geglu_f16(const void *h, int64_t M, int32_t C4, void *y, ctx_stream_t stream);

/* CFG combine + PNDM/PLMS update fused (stable_diffusion_depth.py:428-430,514; diffusers PNDMScheduler
   step_plms with skip_prk_steps).  eps_pair[2,n] (uncond, text); ets[4,n] history ring (newest at
   slot `head`); coef4 = HOST float[4] linear-multistep weights for (e_t, e_t-1, e_t-2, e_t-3) after insertion;
   sample_coeff, eps_coeff from _get_prev_sample; x[n] updated in place; also writes the blended
   epsilon into ets[head]. mode 0: normal; 1: second call of the first step (average with ets[head],
   use cur_sample_ws as x). */
int32_t ctx_cfg_plms_step(const float *eps_pair, int64_t n, float guidance, float *ets, int32_t head,
                          const float *coef4, float sample_coeff, float eps_coeff, int32_t mode,
                          float *cur_sample_ws, float *x, ctx_stream_t stream);

/* Live per-kernel timing for bench.py's roofline: between begin and end every MFMA kernel launch (class 0 =
   GEMM / implicit-GEMM conv, class 1 = attention) is bracketed by dispatch-tight HIP events on ITS stream;
   end() synchronises them and returns the summed kernel time and the launch count of one class. */
int32_t ctx_profile_begin(void);
int32_t ctx_profile_end(int32_t klass, double *total_ms /*host*/, int64_t *count /*host*/);

/* Benchmark support (tools/bench_gemm.py): `iters` back-to-back launches timed on the device; conv_B > 0 selects the
   implicit-GEMM conv (N = Cout, input [conv_B, conv_H, conv_W, conv_Cin]; conv_flags bit 0: stride 2, bit 1: fused nearest
   x2 upsample).  epi 1: GEGLU epilogue.  splitk < 0: the UNet executor's own choice (needs `part`).  Returns average ms
   per launch (< 0: error). */
float ctx_bench_gemm(const void *A, const void *Wt, const void *bias, const void *residual, int32_t M, int32_t N, int32_t K,
                     void *C, int32_t conv_B, int32_t conv_H, int32_t conv_W, int32_t conv_Cin, int32_t conv_flags, int32_t epi,
                     void *part, int32_t splitk, int32_t iters, ctx_stream_t stream);

/* Tuning support (tools/tune_gemm.py): force the tile id of gemm.hip (-1 = planner's choice) and the 256x256 kernel of
   gemm8.hip (-1 planner, 0 never, 1 whenever applicable) for every following GEMM / conv launch of this process. */
void ctx_gemm_tune(int32_t tile, int32_t gemm8);

/* Unit-test support: one 32x32 tile through the MFMA fragment maps the kernels assume.
   which 0: f16 32x32x16 (A[32][16], Bt[32][16]); 1: f32 32x32x2 (A[32][2], Bt[32][2]); C[32][32] f32.
   which 2: the transposing LDS read ds_read_b64_tr_b16 on an f16 tile A[8][32]; C[64 lanes][4] f32 (Bt unused but non-null). */
int32_t ctx_probe_mfma(int32_t which, const void *A, const void *Bt, float *C, ctx_stream_t stream);

/* Measurement support (tools/probe_stage.py): bytes per second one workgroup per CU stages out of L2 by LDS-DMA (mode 0), by
   16-byte register loads (mode 1) or by both (mode 2); `waves` waves per workgroup, `u` (4 | 8) KiB in flight per wave.
   src >= 2 MiB of device memory, sink >= 4 bytes.  Returns the milliseconds of `iters` turns, negative on error. */
float ctx_probe_stage(int32_t mode, int32_t waves, int32_t u, int32_t iters, int32_t shared_region, const void *src, void *sink,
                      ctx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
