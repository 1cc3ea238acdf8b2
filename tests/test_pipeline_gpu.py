"""GPU end-to-end: the per-view paint path assembled from the HIP kernels —
img2img_step (denoise loop) against the oracle's fp32 UNet + PNDM restatement, and the trainer's
define_view_weights / paint_viewpoint / paint on a bundled mesh with a small random-init UNet."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _tiny_sd(dev, seed=0):
    from contexture_nerf_amd.unet import UNet2DConditionModel
    from contexture_nerf_amd.stable_diffusion_depth import StableDiffusion
    from oracle import unet_ref
    cfg = unet_ref.tiny_config()
    torch.manual_seed(seed)
    ref = unet_ref.randomize_affine(unet_ref.UNet2DConditionModelRef(cfg)).eval()
    net = UNet2DConditionModel(cfg, device=dev, init=False)
    net.load_state_dict(ref.state_dict())
    return StableDiffusion(dev, unet=net), ref, cfg


@pytest.mark.parametrize("steps", [2, 6, 50])
def test_img2img_denoised_latents_vs_oracle(dev, steps):
    """Denoised latents after the full PLMS loop (steps+1 UNet evals, CFG 10) vs the fp32 oracle loop; 50 steps = the chain
    BASELINE quotes (51 evaluations, the error fed back through the 4-step PLMS history at guidance 10).
    Gate: measured x 1.15 (fp16 UNet error 1.4e-3 per evaluation, amplified by guidance 10 and chained steps)."""
    from oracle.scheduler import PNDMRef, cfg as cfg_combine
    sd, ref, cfg = _tiny_sd(dev)
    g = torch.Generator().manual_seed(5)
    text_z = torch.randn(2, 9, cfg['cross_attention_dim'], generator=g)
    depth = torch.rand(1, 1, 80, 80, generator=g)
    mask = torch.ones(1, 1, 80, 80)
    size = 128                                              # latent 16x16
    # product
    torch.manual_seed(123)
    rgb, lat = sd.img2img_step(text_z.to(dev), torch.rand(1, 3, 80, 80).to(dev), depth.to(dev), guidance_scale=10.0, strength=1.0,
                               num_inference_steps=steps, update_mask=mask.to(dev), latent_mode=False, fixed_seed=7, image_size=size), None
    rgb = rgb[0]
    # oracle loop with the same initial noise: replay the product's RNG sequence
    from contexture_nerf_amd.utils import seed_everything
    seed_everything(7)
    _ = torch.randn(1, 4, size // 8, size // 8, device=dev)          # randn_like(latents) -> `noise`, drawn first
    lat0 = torch.randn(1, 4, size // 8, size // 8, device=dev).cpu().numpy()
    d = F.interpolate(depth, size=(size // 8, size // 8), mode='bicubic', align_corners=False)
    d = 2.0 * (d - d.min()) / (d.max() - d.min()) - 1.0
    sch = PNDMRef()
    ts = sch.set_timesteps(steps)
    x = lat0
    with torch.no_grad():
        for t in ts:
            xin = torch.cat([torch.tensor(x)] * 2)
            xin = torch.cat([xin, torch.cat([d] * 2)], 1)
            eps = ref(xin, torch.tensor(float(t)), text_z)['sample'].numpy()
            x = sch.step(cfg_combine(eps, 10.0), t, x)
    # product latents: rerun the loop exposing latents (latent_mode returns them)
    seed_everything(7)
    rgb2, lat_p = sd.img2img_step(text_z.to(dev), torch.zeros(1, 4, size // 8, size // 8, device=dev), depth.to(dev), guidance_scale=10.0,
                                  strength=1.0, num_inference_steps=steps, update_mask=mask.to(dev), latent_mode=True, fixed_seed=7,
                                  image_size=size)
    rel = np.linalg.norm(lat_p.cpu().numpy() - x) / np.linalg.norm(x)
    print(f"img2img {steps} steps: denoised-latent rel L2 vs fp32 oracle = {rel:.3e}")
    # measured x 1.15 — 2 steps: 1.6-2.35e-3 by box / plan table, 6 steps: 1.6-2.2e-3, 50 steps (51 evaluations, the chain BASELINE
    # quotes): 9.2e-4 — the errors of successive evaluations are independent and the PLMS history averages them
    gate = {2: 2.8e-3, 6: 2.6e-3, 50: 1.2e-3}[steps]
    assert rel < gate, rel
    assert rgb.shape == (1, 3, size, size) and torch.isfinite(rgb).all()
    assert torch.equal(rgb, rgb2)                                   # same seed => same result (deterministic kernels)


def test_trainer_view_weights_paint_and_atlas(dev, meshes):
    from contexture_nerf_amd import config as CFG
    from contexture_nerf_amd.trainer import ConTEXTure
    from oracle import geometry as og
    cfg = CFG.TrainConfig()
    cfg.guide.text = "a test mesh"
    cfg.guide.shape_path = "shapes/spot_triangulated.obj"
    cfg.guide.texture_resolution = 128
    cfg.guide.guidance_scale = 10.0
    cfg.guide.sd_image_size = 128
    cfg.guide.num_inference_steps = 2
    cfg.render.train_grid_size = 160
    sd, _, _ = _tiny_sd(dev)
    tr = ConTEXTure(cfg, device=dev, diffusion=sd)
    assert len(tr.train_views) == 7 and tr.mesh_model.face_attributes.shape == (1, 5856, 3, 2)
    # view weights for all 7 views == oracle on the same raster
    masks = tr.define_view_weights()
    c = tr._vw_cache
    mz, om = og.view_weights(c['face_idx'][:, 0].cpu().numpy(), c['face_normals'][:, 2, :].cpu().numpy())
    assert np.array_equal(masks[:, 0].cpu().numpy(), om)
    # render contract (textured_mesh.py:476-580)
    out = tr.mesh_model.render(theta=[1.0471976, 1.0471976], phi=[0.0, 0.5235988], radius=[1.5, 1.5],
                               background=torch.tensor([0.5, 0.5, 0.5], device=dev))
    assert set(out) == {'image', 'mask', 'background', 'foreground', 'depth', 'normals', 'render_cache', 'texture_map', 'mlp_output'}
    assert out['image'].shape == (2, 3, 160, 160) and out['texture_map'].shape == (1, 3, 128, 128)
    assert float(out['image'].detach().min()) >= 0 and float(out['image'].detach().max()) <= 1
    m = out['mask'][0, 0] > 0
    assert 0.1 < m.float().mean() < 0.9
    d = out['depth'][0, 0]
    assert float(d[m].min()) >= 0 and float(d[m].max()) == 1.0 and float(d[~m].abs().max()) == 0
    out2 = tr.mesh_model.render(background=torch.tensor([0.5, 0.5, 0.5], device=dev), render_cache=out['render_cache'])
    assert torch.equal(out2['image'], out['image'])                 # cached raster path
    # one painted view and the full sharded loop (world 1)
    rgb, obj = tr.paint_viewpoint(tr.train_views[0])
    assert rgb.shape == (1, 3, 160, 160) and obj.shape == (1, 1, 160, 160) and torch.isfinite(rgb).all()
    # two views in flight (two streams, two engines over one weight blob) == the same two views one after the other
    rgb1, obj1 = tr.paint_viewpoint(tr.train_views[1])
    pair = tr.paint_viewpoints_pair(tr.train_views[0], tr.train_views[1])
    assert torch.equal(pair[0][0], rgb) and torch.equal(pair[1][0], rgb1) and torch.equal(pair[1][1], obj1)
    atlas, cov = tr.paint()
    assert atlas.shape == (3, 128, 128) and torch.isfinite(atlas).all()
    tr.cfg.optim.views_in_flight = 1                                    # serial loop gives the same atlas
    atlas1, _ = tr.paint()
    assert torch.equal(atlas1, atlas)                                   # integer (2^-32 fixed-point) scatter: order-free
    assert 0.05 < float((cov > 0).float().mean()) <= 1.0
    assert float(atlas.min()) >= 0 and float(atlas.max()) <= 1.0 + 1e-5
    # outputs on disk (export_mesh layout): the OBJ reads back, the albedo map has the atlas resolution
    import struct, tempfile, os
    from contexture_nerf_amd import kal
    with tempfile.TemporaryDirectory() as td:
        p = tr.export(os.path.join(td, 'mesh'))
        mm = kal.io.obj.import_mesh(os.path.join(p, 'mesh.obj'))
        assert mm.faces.shape == tr.mesh_model.mesh.faces.shape and mm.uvs.shape[1] == 2
        raw = open(os.path.join(p, 'albedo.png'), 'rb').read()
        assert struct.unpack('>II', raw[16:24]) == (128, 128) and os.path.exists(os.path.join(p, 'mesh.mtl'))


def test_img2img_step_batched_lockstep_views(dev):
    """StableDiffusion.img2img_step_batched: V views denoised in lockstep as ONE UNet evaluation of batch 2V per step.
    (a) Inside full groups a view's result does not depend on which other views share its batch, nor on its position in it: the
        executor's plan depends on the row count only and rows are arithmetically independent  -> torch.equal.
    (b) Views left over after the full groups go through the batch-2 streams and equal img2img_step bit for bit.
    (c) Against the batch-2 loop (other tile / split-K plans -> another summation order) the batched latents agree to the fp16
        tolerance of the parity tests, not bit for bit."""
    sd, _, cfg = _tiny_sd(dev)
    g = torch.Generator().manual_seed(9)
    calls = []
    for v in range(5):
        calls.append(dict(text_embeddings=torch.randn(2, 9, cfg['cross_attention_dim'], generator=g).to(dev),
                          inputs=torch.rand(1, 3, 72, 72, generator=g).to(dev), original_depth_mask=torch.rand(1, 1, 72, 72, generator=g).to(dev),
                          guidance_scale=10.0, strength=1.0, num_inference_steps=4, update_mask=torch.ones(1, 1, 72, 72, device=dev),
                          latent_mode=False, fixed_seed=11 + v, image_size=128))
    lat = lambda kw: dict(kw, latent_mode=True, inputs=torch.zeros(1, 4, 16, 16, device=dev))
    serial = []
    for k in range(5):
        kw = lat(calls[k])
        serial.append(sd.img2img_step(kw['text_embeddings'], kw['inputs'], kw['original_depth_mask'],
                                      **{kk: vv for kk, vv in kw.items() if kk not in ('text_embeddings', 'inputs', 'original_depth_mask')}))
    a = sd.img2img_step_batched([lat(c) for c in calls[:3]], views_per_eval=3)
    b = sd.img2img_step_batched([lat(calls[2]), lat(calls[4]), lat(calls[0])], views_per_eval=3)      # other mates, other positions
    d = sd.img2img_step_batched([lat(c_) for c_ in calls], views_per_eval=3)                        # one full group + two left over
    assert torch.equal(a[0][1], b[2][1]) and torch.equal(a[2][1], b[0][1]) and torch.equal(a[0][0], b[2][0])
    assert all(torch.equal(a[k][1], d[k][1]) for k in range(3))
    assert torch.equal(d[3][1], serial[3][1]) and torch.equal(d[4][1], serial[4][1])                # the remainder: batch-2 streams
    for k in range(3):
        rel = float((d[k][1] - serial[k][1]).norm() / serial[k][1].norm())
        assert rel < 4e-3 and torch.isfinite(d[k][0]).all(), (k, rel)         # 0 on this tiny net (the same plans at batch 6); ~1e-3 at full size
    assert not torch.equal(d[0][1], d[1][1])
    # fewer views than a group: everything takes the stream path; image-mode calls return (rgb, []) like img2img_step
    e = sd.img2img_step_batched([lat(calls[0]), lat(calls[1])], views_per_eval=3)
    assert torch.equal(e[0][1], serial[0][1]) and torch.equal(e[1][1], serial[1][1])
    f = sd.img2img_step_batched(calls[:2], views_per_eval=2)
    assert f[0][0].shape == (1, 3, 128, 128) and f[0][1] == []


def test_mesh_batch_painter_configs3_on_the_hip_path(dev):
    """BASELINE configs[3] on the HIP path: the 8-mesh batch (the 6 bundled shapes + 2 repeats), 6 views each, through
    MeshBatchPainter.paint_all (3 denoise loops in flight, groups spanning mesh boundaries) with a tiny random-init UNet on a small
    grid.  Every mesh's atlas and coverage must be BIT-IDENTICAL to that mesh's own ConTEXTure.paint over the same six views
    (serial loop): the denoise loops in flight are bit-reproducible per view and the UV scatter sums integers."""
    from contexture_nerf_amd import config as CFG
    from contexture_nerf_amd.trainer import ConTEXTure
    from contexture_nerf_amd.batch import MeshBatchPainter, schedule
    names = ["nascar", "spot_triangulated", "bunny", "blub_no_texture", "sphere", "env_sphere", "nascar", "spot_triangulated"]
    sd, _, _ = _tiny_sd(dev)

    def make(nm, in_flight):
        cfg = CFG.TrainConfig()
        cfg.guide.text = f"a photo of a {nm}"
        cfg.guide.shape_path = f"shapes/{nm}.obj"
        cfg.guide.texture_resolution = 128
        cfg.guide.guidance_scale = 10.0
        cfg.guide.sd_image_size = 128
        cfg.guide.num_inference_steps = 2
        cfg.render.train_grid_size = 160
        cfg.optim.views_in_flight = in_flight
        tr = ConTEXTure(cfg, device=dev, diffusion=sd)
        tr.text_z = sd.get_text_embeds([cfg.guide.text])
        return tr
    trainers = [make(nm, 3) for nm in names]
    bp = MeshBatchPainter(trainers)
    assert bp.view_ids == [1, 2, 3, 4, 5, 6] and bp.plan == schedule(8, 6, 1) and len(bp.plan[0]) == 48
    res = bp.paint_all()
    assert len(res) == 8
    for m, nm in enumerate(names):
        atlas, cov = res[m]
        assert atlas.shape == (3, 128, 128) and torch.isfinite(atlas).all() and float((cov > 0).float().mean()) > 0.02
        solo = make(nm, 1)
        solo.train_views = [solo.train_views[i] for i in bp.view_ids]           # the same six Zero123++ views, painted one by one
        a1, c1 = solo.paint()
        assert torch.equal(c1, cov), f"{nm}: coverage differs from the mesh's own paint"
        assert torch.equal(a1, atlas), f"{nm}: atlas differs from the mesh's own paint ({(a1 != atlas).sum().item()} texels)"
    assert torch.equal(res[0][0], res[6][0]) and torch.equal(res[1][0], res[7][0])       # the two repeats reproduce their originals


def test_volume_render_and_refine(dev):
    """BASELINE configs[4] in small: ray-marched 3-D field -> depth map -> SD2-depth refine (tiny random-init UNet).
    Row-tile sharding reproduces the unsharded render bit for bit (rays are independent), and the hierarchical pass runs."""
    from contexture_nerf_amd import volume_render as vr, run_nerf_helpers as rnh
    sd, _, cfg = _tiny_sd(dev)
    torch.manual_seed(2)
    field = rnh.NeRF2D(D=8, W=64, input_ch=63, output_ch=4, skips=[4]).to(dev)
    with torch.no_grad():
        field.output_linear.bias[3] = 3.0           # some density everywhere, so that the accumulated alpha is not ~0
    H = W = 40
    c2w = torch.tensor([[1, 0, 0, 0.0], [0, 1, 0, 0.0], [0, 0, 1, 1.5]], dtype=torch.float32, device=dev)
    K = vr.pinhole(H, W)
    full = vr.render_image(field, H, W, K, c2w, 0.5, 2.5, 24)
    tiles = [vr.render_image(field, H, W, K, c2w, 0.5, 2.5, 24, rows=vr.shard_rows(H, r, 3)) for r in range(3)]
    assert [t['rgb'].shape[0] for t in tiles] == [14, 13, 13]
    for k in ('rgb', 'depth', 'acc'):
        assert torch.equal(torch.cat([t[k] for t in tiles], 0), full[k])
    fine = vr.render_image(field, H, W, K, c2w, 0.5, 2.5, 24, N_importance=16)
    assert torch.isfinite(fine['rgb']).all() and fine['rgb'].shape == (H, W, 3)
    g = torch.Generator().manual_seed(5)
    text_z = torch.randn(2, 9, cfg['cross_attention_dim'], generator=g).to(dev)
    out, r = vr.render_and_refine(field, sd, text_z, H, W, c2w, N_samples=24, guidance_scale=5.0, num_inference_steps=3,
                                  fixed_seed=3, image_size=128)
    assert out.shape == (1, 3, 128, 128) and torch.isfinite(out).all()
    assert float(r['acc'].min()) > 0.5
    d = vr.depth_for_diffusion(r['depth'], r['acc'])
    assert float(d.min()) >= 0.5 and float(d.max()) <= 1.0


def test_zero123plus_pipeline_one_step_and_loop(dev):
    """Tensor-level Zero123++ pipeline on tiny random-init engines: the SDS loop's call shape (one explicit timestep, given noisy
    latents, noise_pred through callback_on_step_end, src/training/trainer.py:785-795) equals the manual composition of the UNet
    stack + CFG; a short sampling loop decodes to an image."""
    from contexture_nerf_amd.unet import UNet2DConditionModel, ControlNetModel
    from contexture_nerf_amd.vae import AutoencoderKL
    from contexture_nerf_amd.scheduler import EulerAncestralDiscreteScheduler, DDPMScheduler
    from contexture_nerf_amd.zero123plus import RefOnlyNoisedUNet, DepthControlUNet, Zero123PlusPipeline, scale_latents
    from oracle import unet_ref
    cfg = unet_ref.tiny_config(in_channels=4)
    net = UNet2DConditionModel(cfg, device=dev, seed=1)
    cnet = ControlNetModel(cfg, device=dev, seed=2)
    vae = AutoencoderKL(dict(latent_channels=4, out_channels=3, block_out_channels=(64, 128, 128, 128), layers_per_block=1, groups=32), device=dev, seed=3)
    sch = EulerAncestralDiscreteScheduler()
    stack = DepthControlUNet(RefOnlyNoisedUNet(net, DDPMScheduler(), sch).eval(), cnet, conditioning_scale=2.0).eval()
    pipe = Zero123PlusPipeline(vae, stack, sch)
    g = torch.Generator().manual_seed(4)
    image = (torch.rand(1, 3, 64, 64, generator=g) * 2 - 1).to(dev)           # condition image -> 8x8 latent (64 reference tokens)
    depth = torch.rand(1, 3, 192, 128, generator=g).to(dev)                   # 3x2 depth grid, 8x the 24x16 latent
    pe = torch.randn(1, 9, cfg['cross_attention_dim'], generator=g).to(dev)
    z_t = torch.randn(1, 4, 24, 16, generator=g).to(dev)
    seen = {}

    def cb(p, i, t, kw):
        seen['noise_pred'] = kw['noise_pred'].clone(); seen['t'] = float(t)
        return kw
    torch.manual_seed(11)
    out = pipe(image, prompt_embeds=pe, depth_image=depth, guidance_scale=10.0, num_inference_steps=1, timesteps=[515.0], latents=z_t,
               width=128, height=192, output_type='latent', callback_on_step_end=cb, callback_on_step_end_tensor_inputs=["latents", "noise_pred"])
    assert seen['t'] == 515.0 and seen['noise_pred'].shape == z_t.shape and torch.isfinite(seen['noise_pred']).all()
    assert out.images.shape == z_t.shape
    # manual composition with the same RNG stream (condition-latent samples, then the reference pass's noise draw)
    torch.manual_seed(11)
    pos = vae.encode(image).latent_dist.sample()                              # the pipeline encodes the image first, then the zeros
    cl = torch.cat([vae.encode(torch.zeros_like(image)).latent_dist.sample(), pos])
    sch.set_timesteps(1, timesteps=[515.0])
    # caller-supplied latents are multiplied by init_noise_sigma too (diffusers prepare_latents), then scaled for the model
    assert float(sch.init_noise_sigma) > 1.0
    x = sch.scale_model_input(torch.cat([z_t * sch.init_noise_sigma] * 2), sch.timesteps[0])
    ctx = torch.cat([torch.zeros_like(pe), pe])
    v = stack(x, sch.timesteps[0].reshape(1), ctx, cross_attention_kwargs=dict(cond_lat=cl, control_depth=torch.cat([depth] * 2)))['sample']
    vu, vt = v.chunk(2)
    assert torch.equal(vu + 10.0 * (vt - vu), seen['noise_pred'])
    # a short sampling loop to an image
    img = pipe(image, prompt_embeds=pe, depth_image=depth, guidance_scale=4.0, num_inference_steps=3, width=128, height=192,
               generator=torch.Generator(device=dev).manual_seed(2)).images
    assert img.shape == (1, 3, 192, 128) and torch.isfinite(img).all() and float(img.min()) >= 0 and float(img.max()) <= 1


def test_sds_iteration_targets(dev):
    """The denoise side of one SDS iteration (trainer.py:700-850) on tiny engines: shapes, the identities targets = z0 - grad and
    grad = 0.2 (1 - abar) sqrt(abar) (v_pred - v), loss = 0.5 * |grad tile|^2, the DDPM noising identity, determinism under a seed."""
    from contexture_nerf_amd.unet import UNet2DConditionModel, ControlNetModel
    from contexture_nerf_amd.vae import AutoencoderKL
    from contexture_nerf_amd.scheduler import EulerAncestralDiscreteScheduler, DDPMScheduler
    from contexture_nerf_amd.zero123plus import RefOnlyNoisedUNet, DepthControlUNet, Zero123PlusPipeline
    from contexture_nerf_amd.utils import DreamTimeScheduler
    from contexture_nerf_amd import sds
    from oracle import unet_ref
    cfg = unet_ref.tiny_config(in_channels=4)
    net = UNet2DConditionModel(cfg, device=dev, seed=1); cnet = ControlNetModel(cfg, device=dev, seed=2)
    vae = AutoencoderKL(dict(latent_channels=4, out_channels=3, block_out_channels=(64, 128, 128, 128), layers_per_block=1, groups=32), device=dev, seed=3)
    train_sched, val_sched = DDPMScheduler(), EulerAncestralDiscreteScheduler()
    pipe = Zero123PlusPipeline(vae, DepthControlUNet(RefOnlyNoisedUNet(net, train_sched, val_sched).eval(), cnet, conditioning_scale=2.0).eval(), val_sched)
    g = torch.Generator().manual_seed(7)
    six = torch.rand(6, 3, 64, 64, generator=g).to(dev)                       # six rendered views -> 192 x 128 grid -> 24 x 16 latent
    cond = (torch.rand(1, 3, 64, 64, generator=g) * 2 - 1).to(dev)
    depth = torch.rand(1, 3, 192, 128, generator=g).to(dev)
    pe = torch.randn(1, 9, cfg['cross_attention_dim'], generator=g).to(dev)
    t = int(DreamTimeScheduler(train_sched.alphas_cumprod, 100).get_t(40))

    def run():
        torch.manual_seed(5)
        return sds.sds_iteration_targets(pipe, six, cond, depth, pe, t, train_sched.alphas_cumprod, train_sched.add_noise, index_to_train=4)
    r = run()
    assert r['z0'].shape == (1, 4, 24, 16) and r['v_pred'].shape == r['z0'].shape and torch.isfinite(r['targets']).all()
    ac = train_sched.alphas_cumprod[t].item()
    want_grad = 0.2 * (1 - ac) * ac ** 0.5 * (r['v_pred'] - r['v'])
    assert torch.allclose(r['grad'], want_grad, rtol=1e-5, atol=1e-7)
    assert torch.allclose(r['targets'], r['z0'] - r['grad'], rtol=0, atol=1e-6)
    noise = (r['v'] + (1 - ac) ** 0.5 * r['z0']) / ac ** 0.5                 # invert v = sqrt(abar) eps - sqrt(1-abar) z0
    assert torch.allclose(r['latents_noisy'], ac ** 0.5 * r['z0'] + (1 - ac) ** 0.5 * noise, rtol=1e-4, atol=1e-5)
    from contexture_nerf_amd.utils import split_3x2_grid_to_tensor_with_6_elements as split
    assert abs(r['loss'].item() - 0.5 * split(r['grad'].float(), 8)[4].pow(2).sum().item()) <= 1e-4 * (1 + r['loss'].item())
    r2 = run()
    assert torch.equal(r2['v_pred'], r['v_pred']) and torch.equal(r2['targets'], r['targets'])


def _tiny_zero123(dev, tr):
    """Tiny Zero123++ stack with fp32 twins of its VAE for the oracle side."""
    from contexture_nerf_amd.unet import UNet2DConditionModel, ControlNetModel
    from contexture_nerf_amd.vae import AutoencoderKL
    from oracle import unet_ref, vae_ref
    ucfg = unet_ref.tiny_config(in_channels=4)
    vcfg = dict(latent_channels=4, out_channels=3, block_out_channels=(64, 128, 128, 128), layers_per_block=1, groups=32)
    torch.manual_seed(31)
    vref = unet_ref.randomize_affine(vae_ref.AutoencoderKLRef(vcfg)).eval()
    vae = AutoencoderKL(vcfg, device=dev, init=False); vae.load_state_dict(vref.state_dict())
    tr.init_zero123plus(unet=UNet2DConditionModel(ucfg, device=dev, seed=1), controlnet=ControlNetModel(ucfg, device=dev, seed=2), vae=vae,
                        unet_config=ucfg)
    return vref


def test_sds_iteration_backward_vs_autograd_oracle(dev):
    """ONE full iteration of the reference's SDS loop (src/training/trainer.py:700-866) on tiny engines: texture field -> cached
    raster render of the 7 views -> six crops resized to tile^2 -> 3x2 grid -> VAE encode -> loss on one latent tile ->
    backward.  The product's chain (HIP: VAE-encoder backward, texture_mapping backward, texture-field backward; torch only for
    the crop/resize) against a plain torch fp32 autograd restatement of the SAME chain (torch MLP, F.grid_sample, the oracle's
    Encoder) with the same noise sample and the same (detached) targets: every parameter gradient of the UV-MLP must agree."""
    from contexture_nerf_amd import config as CFG, sds, utils
    from contexture_nerf_amd.trainer import ConTEXTure
    from contexture_nerf_amd.scheduler import DDPMScheduler
    import types
    cfg = CFG.TrainConfig()
    cfg.guide.text = "a test mesh"; cfg.guide.shape_path = "shapes/spot_triangulated.obj"
    cfg.guide.texture_resolution = 128; cfg.guide.sd_image_size = 128; cfg.guide.num_inference_steps = 2
    cfg.render.train_grid_size = 192
    sd, _, _ = _tiny_sd(dev)
    tr = ConTEXTure(cfg, device=dev, diffusion=sd)
    vref = _tiny_zero123(dev, tr)
    pipe, tile = tr.zero123plus, 64
    gray = torch.tensor([0.5, 0.5, 0.5], device=dev)
    with torch.no_grad():
        views = tr.train_views
        allv = tr.mesh_model.render(theta=[v['theta'] for v in views], phi=[tr._offset_phi(v['phi']) for v in views],
                                    radius=[float(v['radius']) for v in views], background=gray)
        rc, masks = allv['render_cache'], allv['mask']
        boxes = [utils.get_nonzero_region_tuple(masks[j, 0]) for j in range(1, 7)]
        depth_grid = sds.build_depth_grid(1.0 - allv['depth'], masks, size=tile)
    g = torch.Generator().manual_seed(3)
    cond = (torch.rand(1, 3, tile, tile, generator=g) * 2 - 1).to(dev)
    noise = torch.randn(1, 4, 3 * tile // 8, 2 * tile // 8, generator=g)
    # a VAE shim that samples with a KNOWN noise tensor, so that the oracle can take the same sample
    real_vae = pipe.vae

    class FixedNoise:
        def encode(self, x):
            d = real_vae.encode(x).latent_dist
            if d.mean.shape == noise.shape:             # the view grid; the condition image's own encode keeps its random sample
                d.sample = lambda generator=None: d.mean + d.std * noise.to(dev)
            return types.SimpleNamespace(latent_dist=d)

        def __getattr__(self, k):
            return getattr(real_vae, k)
    pipe.vae = FixedNoise()
    train_sched = DDPMScheduler(prediction_type="v_prediction")
    params = list(tr.texture_mlp.parameters())
    for p in params:
        p.grad = None

    def six_views(image):
        return torch.cat([F.interpolate(image[1:][j:j + 1, :, b[0]:b[2], b[1]:b[3]], (tile, tile), mode='bilinear', align_corners=False)
                          for j, b in enumerate(boxes)], 0)
    out = tr.mesh_model.render(render_cache=rc, background=gray)
    torch.manual_seed(17)
    r = sds.sds_iteration(pipe, six_views(out['image']), cond, depth_grid, tr.zero123plus_prompt_embeds, 400, train_sched.alphas_cumprod,
                          train_sched.add_noise, index_to_train=2)
    r['loss'].backward()
    got = [p.grad.detach().cpu().clone() for p in params]
    assert all(torch.isfinite(gp).all() for gp in got) and float(r['loss']) > 0
    # ---- torch fp32 autograd restatement -----------------------------------------------------------------------------
    T = cfg.guide.texture_resolution
    net = tr.texture_mlp
    lin = [(l.weight.detach().cpu().clone().requires_grad_(True), l.bias.detach().cpu().clone().requires_grad_(True))
           for l in list(net.pts_linears) + [net.output_linear]]
    u = torch.stack(torch.meshgrid(torch.linspace(0, 1, T), torch.linspace(0, 1, T), indexing='xy'), dim=-1).reshape(-1, 2)
    freqs = 2.0 ** torch.linspace(0, 9, 10)
    emb = torch.cat([u] + [f(u * fr) for fr in freqs for f in (torch.sin, torch.cos)], -1)
    h = emb
    for i in range(8):
        h = torch.relu(h @ lin[i][0].t() + lin[i][1])
        if i == 4:
            h = torch.cat([emb, h], -1)
    mlp_out = h @ lin[8][0].t() + lin[8][1]
    tex = ((mlp_out.tanh() + 1) / 2).reshape(1, T, T, 3).permute(0, 3, 1, 2)
    with torch.no_grad():
        assert torch.allclose(tex, out['texture_map'].detach().cpu(), atol=2e-4)             # same atlas layout and values
    uv = rc['uv_features'].cpu(); fidx = rc['face_idx'].cpu()
    grid = torch.stack([uv[..., 0], 1 - uv[..., 1]], -1) * 2 - 1
    feat = F.grid_sample(tex.expand(uv.shape[0], -1, -1, -1), grid, mode='bilinear', padding_mode='border', align_corners=False)
    m = (fidx > -1).float()[:, None]
    image = (gray.cpu().view(1, 3, 1, 1) * (1 - m) + feat * m * m).clamp(0, 1)
    with torch.no_grad():
        assert torch.allclose(image, out['image'].detach().cpu(), atol=3e-4)
    six = six_views(image)
    gridimg = utils.scale_image(sds.views_to_grid(six, tile) * 2 - 1)
    mom = vref.encode_moments(gridimg)
    mean, logvar = mom.chunk(2, 1)
    z0 = utils.scale_latents((mean + torch.exp(0.5 * logvar.clamp(-30, 20)) * noise) * sds.VAE_SCALING)
    with torch.no_grad():
        rz = float((z0 - r['z0'].detach().cpu()).norm() / z0.norm())
        assert rz < 5e-3, rz
    loss, _ = sds.tile_loss(z0, r['targets'].detach().cpu(), 2)
    loss.backward()
    assert abs(float(loss) - float(r['loss'])) <= 2e-2 * abs(float(loss)) + 1e-6
    num = sum(float(((a - w_.grad) ** 2).sum()) for a, (w_, b_) in zip(got[0::2], lin)) + \
        sum(float(((a - b_.grad) ** 2).sum()) for a, (w_, b_) in zip(got[1::2], lin))
    den = sum(float((w_.grad ** 2).sum()) + float((b_.grad ** 2).sum()) for (w_, b_) in lin)
    rel = (num / den) ** 0.5
    print(f"SDS iteration: loss {float(r['loss']):.4f} (oracle {float(loss):.4f}), rel L2 of the UV-MLP parameter gradient vs torch autograd = {rel:.3e}")
    assert den > 0 and rel < 2e-2, rel
    pipe.vae = real_vae


def test_paint_zero123plus_loop_and_eval(dev, tmp_path):
    """The reference's live paint(): front view painted with SD2-depth, then SDS iterations with Adam on the UV-MLP
    (paint_zero123plus, trainer.py:545-911) on tiny engines — runs, every record is finite, the parameters move, the DreamTime
    timesteps descend from the noisy end; then the eval orbit (full_eval, :913-968) writes its frames, atlas and mesh."""
    import os
    from contexture_nerf_amd import config as CFG
    from contexture_nerf_amd.trainer import ConTEXTure
    cfg = CFG.TrainConfig()
    cfg.guide.text = "a test mesh"; cfg.guide.shape_path = "shapes/spot_triangulated.obj"
    cfg.guide.texture_resolution = 128; cfg.guide.sd_image_size = 128; cfg.guide.num_inference_steps = 2
    cfg.render.train_grid_size = 192; cfg.render.eval_grid_size = 96
    cfg.log.exp_root = tmp_path; cfg.log.exp_name = "sds"; cfg.log.full_eval_size = 5
    sd, _, _ = _tiny_sd(dev)
    tr = ConTEXTure(cfg, device=dev, diffusion=sd)
    _tiny_zero123(dev, tr)
    before = [p.detach().clone() for p in tr.texture_mlp.parameters()]
    log = tr.paint_zero123plus(iterations=4, tile=64)
    assert [r['i'] for r in log] == [0, 1, 2, 3] and all(np.isfinite(r['loss']) and np.isfinite(r['grad_norm']) and r['grad_norm'] > 0 for r in log)
    assert log[0]['t'] >= log[-1]['t'] and log[0]['t'] > 900
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, tr.texture_mlp.parameters()))
    n = tr.full_eval()
    out = cfg.log.exp_dir / 'results'
    assert n == 5 and len([f for f in os.listdir(out) if f.endswith('_rgb.jpg')]) == 5
    assert os.path.exists(out / "eval:texture_atlas:texture.png") and os.path.exists(cfg.log.exp_dir / 'mesh' / 'mesh.obj')
    from contexture_nerf_amd.video import read_mjpeg_avi
    fps, frames = read_mjpeg_avi(out / f"eval:constructed_video:all_rendered_rgb_{cfg.optim.seed}.avi")      # the reference's mp4, as Motion-JPEG
    assert fps == 25 and len(frames) == 5 and frames[0].shape[2] == 3
    # should_project_back (the reference's default): the painted view lands in the running atlas and the fitted render returns
    rgb, obj = tr.paint_viewpoint(tr.train_views[1])
    assert tr.fitted_pred_rgb.shape == rgb.shape and float(tr.atlas_contrib[3].sum()) > 0
    m = (obj > 0).expand_as(rgb)
    assert float((tr.fitted_pred_rgb - rgb)[m].abs().mean()) < 0.08       # the scattered atlas reproduces the view it was painted from


def test_zero123plus_inpaint_and_blend_extension(dev):
    """ConTEXTure's extension of the Zero123++ denoising loop (spec: src/zero123plus.py:436-440, 650-708) on tiny engines:
    with use_blending the result outside the mask is the clean render latents (final blend) and every non-inpaint step
    re-anchors the latents; with use_inpaint steps 11..19 are predicted by the 9-channel inpaint UNet on
    cat([latents, mask, masked_input_latents]); the plain call is unchanged."""
    from contexture_nerf_amd.unet import UNet2DConditionModel
    from contexture_nerf_amd.vae import AutoencoderKL
    from contexture_nerf_amd.scheduler import EulerAncestralDiscreteScheduler, DDPMScheduler
    from contexture_nerf_amd.zero123plus import RefOnlyNoisedUNet, Zero123PlusPipeline, unscale_latents
    from contexture_nerf_amd._lib import CtxError
    from oracle import unet_ref
    cfg = unet_ref.tiny_config(in_channels=4)
    net = UNet2DConditionModel(cfg, device=dev, seed=1)
    vae = AutoencoderKL(dict(latent_channels=4, out_channels=3, block_out_channels=(64, 128, 128, 128), layers_per_block=1, groups=32), device=dev, seed=3)
    sch = EulerAncestralDiscreteScheduler()
    pipe = Zero123PlusPipeline(vae, RefOnlyNoisedUNet(net, DDPMScheduler(), sch).eval(), sch)
    g = torch.Generator().manual_seed(4)
    image = (torch.rand(1, 3, 64, 64, generator=g) * 2 - 1).to(dev)
    pe = torch.randn(1, 9, cfg['cross_attention_dim'], generator=g).to(dev)
    mask = (torch.rand(1, 1, 24, 16, generator=g) > 0.5).float().to(dev)                       # one channel: 4 + 1 + 4 = the inpaint UNet's 9
    renders = torch.randn(1, 4, 24, 16, generator=g).to(dev)
    masked_in = torch.randn(1, 4, 24, 16, generator=g).to(dev)
    kw = dict(prompt_embeds=pe, guidance_scale=4.0, num_inference_steps=22, width=128, height=192, output_type='latent')
    with pytest.raises(CtxError, match="latent_mask_grid"):
        pipe(image, use_blending=True, **kw)
    out = pipe(image, use_blending=True, latent_mask_grid=mask, latent_renders_grid=renders, generator=torch.Generator(device=dev).manual_seed(1), **kw).images
    assert torch.isfinite(out).all()
    outside = (mask == 0).expand(1, 4, 24, 16)
    assert torch.allclose(out[outside], unscale_latents(renders)[outside], atol=1e-6)           # final blend keeps the clean renders
    assert not torch.allclose(out[~outside], unscale_latents(renders)[~outside], atol=1e-3)
    with pytest.raises(CtxError, match="inpaint_unet"):
        pipe(image, use_inpaint=True, latent_mask_grid=mask, masked_input_latents=masked_in, **kw)
    inp = UNet2DConditionModel(dict(cfg, in_channels=9), device=dev, seed=5)
    calls = []

    class Spy:
        def __call__(self, x, t, encoder_hidden_states=None):
            calls.append((tuple(x.shape), float(t)))
            assert torch.equal(x[:, 4:5] > 0, torch.cat([mask] * 2) > 0)       # the mask rides in channel 4 (scaled with the rest, as the spec does)
            return inp(x, t, encoder_hidden_states=encoder_hidden_states)
    pipe.inpaint_unet = Spy()
    out2 = pipe(image, use_inpaint=True, use_blending=True, latent_mask_grid=mask, latent_renders_grid=renders, masked_input_latents=masked_in,
                generator=torch.Generator(device=dev).manual_seed(1), **kw).images
    assert len(calls) == 9 and all(s == (2, 9, 24, 16) for s, _ in calls)                        # steps 11..19, CFG pair, 4 + 1x4 + 4 channels
    assert torch.isfinite(out2).all() and torch.allclose(out2[outside], unscale_latents(renders)[outside], atol=1e-6)
    plain = pipe(image, generator=torch.Generator(device=dev).manual_seed(1), **kw).images
    assert not torch.equal(plain, out)
