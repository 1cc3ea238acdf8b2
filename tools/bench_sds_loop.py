#!/usr/bin/env python3
"""ms per iteration of the reference's SDS loop (`paint_zero123plus`, src/training/trainer.py:644-907) at the REFERENCE's sizes:
1024^2 atlas from the UV-MLP (training forward), 7 views @1200^2 from the cached raster, six 320^2 crops -> 960x640 grid,
VAE encode with autograd (latent 120x80), one Zero123++ evaluation (reference-only attention over 1 600 tokens + depth ControlNet,
CFG 10), tile loss, backward through the VAE encoder / resize / texture_mapping / texture field, Adam.
Random-init engines (no checkpoints offline).  Usage: python tools/bench_sds_loop.py [iterations]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contexture_nerf_amd import config as CFG
from contexture_nerf_amd.trainer import ConTEXTure
from contexture_nerf_amd.stable_diffusion_depth import StableDiffusion

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
cfg = CFG.TrainConfig()
cfg.guide.text = "a photo of a car"
cfg.guide.shape_path = "shapes/nascar.obj"
cfg.guide.guidance_scale = 10.0
cfg.guide.sd_image_size = 512                        # the reference hard-wires 512 for the front view
sd = StableDiffusion(dev)
tr = ConTEXTure(cfg, device=dev, diffusion=sd)
tr.text_z = sd.get_text_embeds([cfg.guide.text])
tr.init_zero123plus()
stamps = []


def on_it(rec):
    torch.cuda.synchronize()
    stamps.append(time.perf_counter())


t0 = time.perf_counter()
log = tr.paint_zero123plus(iterations=iters + 3, on_iteration=on_it)
per = [(b - a) * 1e3 for a, b in zip(stamps[2:-1], stamps[3:])]        # the first iterations size the workspaces
out = {"metric": "ms per SDS iteration (paint_zero123plus at the reference's sizes)", "iterations_timed": len(per),
       "ms_per_iteration": round(sum(per) / len(per), 2), "min_ms": round(min(per), 2), "setup_s": round(stamps[0] - t0, 2),
       "est_min_per_5000_iterations": round(sum(per) / len(per) * 5000 / 6e4, 2),
       "loss_first_last": [round(log[0]['loss'], 4), round(log[-1]['loss'], 4)], "t_first_last": [log[0]['t'], log[-1]['t']],
       "grad_norm_last": log[-1]['grad_norm'], "finite": all(r['loss'] == r['loss'] for r in log),
       "sizes": {"atlas": cfg.guide.texture_resolution, "render": cfg.render.train_grid_size, "views": len(tr.train_views), "tile": 320,
                 "latent": [120, 80]}, "data": "synthetic (random-init engines)"}
print(json.dumps(out))
