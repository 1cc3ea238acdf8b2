#!/usr/bin/env python3
"""Measured sec/mesh of the full per-view paint loop on ONE GPU (BASELINE.json metric, first half): bundled mesh, 6 views,
render grid 1200^2, SD2-depth UNet (random-init, fp16 MFMA engine) at 768^2 with 50 PLMS steps (51 evaluations) per view,
VAE decode, view weights, UV back-projection scatter into the 1024^2 atlas, atlas merge.  Prints one JSON object.
Usage: python tools/bench_mesh.py [--mesh shapes/nascar.obj] [--views 6] [--image 768] [--steps 50] [--in-flight 2]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contexture_nerf_amd import config as CFG
from contexture_nerf_amd.trainer import ConTEXTure
from contexture_nerf_amd.stable_diffusion_depth import StableDiffusion

ap = argparse.ArgumentParser()
ap.add_argument("--mesh", default="shapes/nascar.obj")
ap.add_argument("--views", type=int, default=6)
ap.add_argument("--image", type=int, default=768)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--in-flight", type=int, default=3)
ap.add_argument("--per-eval", type=int, default=6, help="also time the lockstep batched evaluation of this many views (0: skip)")
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = CFG.TrainConfig()
cfg.guide.text = "a photo of a car"
cfg.guide.shape_path = a.mesh
cfg.guide.guidance_scale = 10.0
cfg.guide.sd_image_size = a.image
cfg.guide.num_inference_steps = a.steps
sd = StableDiffusion(dev)                                       # SD2-depth architecture + VAE decoder, seeded random init
tr = ConTEXTure(cfg, device=dev, diffusion=sd)
tr.train_views = tr.train_views[1:1 + a.views]                  # Zero123PlusDataset views 1..6 (SURVEY §8d cfg 3)
tr.text_z = sd.get_text_embeds([cfg.guide.text])
res = {}
for infl in sorted({1, 2, a.in_flight}):
    cfg.optim.views_in_flight = infl
    tr.paint(); torch.cuda.synchronize()                        # warm-up (workspaces, first-touch)
    t = time.perf_counter()
    atlas, cov = tr.paint()
    torch.cuda.synchronize()
    res[infl] = time.perf_counter() - t
    assert torch.isfinite(atlas).all()
batched = None
if a.per_eval > 1:
    cfg.optim.views_per_eval = a.per_eval
    tr.paint(); torch.cuda.synchronize()
    t = time.perf_counter()
    atlas_b, cov_b = tr.paint()
    torch.cuda.synchronize()
    batched = time.perf_counter() - t
    assert torch.isfinite(atlas_b).all()
    cfg.optim.views_per_eval = 0
out = {"metric": "sec/mesh full texture", "mesh": a.mesh, "faces": int(tr.mesh_model.mesh.faces.shape[0]), "views": len(tr.train_views),
       "render_grid": cfg.render.train_grid_size, "image": a.image, "plms_steps": a.steps, "unet_evals_per_view": a.steps + 1,
       "sec_per_mesh_serial": round(res[1], 3), "data": "synthetic (random-init weights, seeded text embedding)",
       "atlas_coverage": round(float((cov > 0).float().mean()), 4)}
for k in sorted(res):
    if k != 1:
        out[f"sec_per_mesh_{k}_views_in_flight"] = round(res[k], 3)
if batched is not None:
    out[f"sec_per_mesh_{a.per_eval}_views_per_evaluation"] = round(batched, 3)
print(json.dumps(out))
