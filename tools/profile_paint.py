#!/usr/bin/env python3
"""Where one painted view's wall time goes outside the denoise loop (synchronised stage timers around ConTEXTure's own calls).
Usage: python tools/profile_paint.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contexture_nerf_amd import config as CFG
from contexture_nerf_amd.trainer import ConTEXTure
from contexture_nerf_amd.stable_diffusion_depth import StableDiffusion

dev = torch.device("cuda:0")
cfg = CFG.TrainConfig(); cfg.guide.text = "a photo of a car"; cfg.guide.shape_path = "shapes/nascar.obj"; cfg.guide.guidance_scale = 10.0
cfg.guide.sd_image_size = 768; cfg.guide.num_inference_steps = 50
sd = StableDiffusion(dev); tr = ConTEXTure(cfg, device=dev, diffusion=sd)
tr.train_views = tr.train_views[1:7]; tr.text_z = sd.get_text_embeds([cfg.guide.text])
tr.paint(); torch.cuda.synchronize()


def T(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return r, (time.perf_counter() - t) * 1e3


out = {}
_, out["define_view_weights_6_views_ms"] = T(lambda: tr.define_view_weights(list(range(6))))
(kw, ctx), out["paint_prepare_ms"] = T(lambda: tr._paint_prepare(tr.train_views[0]))
te, inp, dm = kw.pop('text_embeddings'), kw.pop('inputs'), kw.pop('original_depth_mask')
(lat, depth_mask, um), out["img2img_prepare_incl_vae_encode_ms"] = T(lambda: sd._prepare(inp, dm, kw.get('update_mask'), False, 768))
(rgb, _), out["img2img_step_total_ms"] = T(lambda: sd.img2img_step(te, inp, dm, **kw))
_, out["vae_decode_ms"] = T(lambda: sd.decode_latents(torch.randn(1, 4, 96, 96, device=dev)))
(rgb_out, mask), out["paint_finish_ms"] = T(lambda: tr._paint_finish(ctx, rgb))
_, out["project_back_scatter_ms"] = T(lambda: tr.project_back_scatter(ctx['render_cache'], rgb_out, tr.view_weights[0:1] & (mask > 0)))
_, out["paint_6_views_3_in_flight_ms"] = T(lambda: tr.paint())
print(json.dumps({k: round(v, 2) for k, v in out.items()}))
