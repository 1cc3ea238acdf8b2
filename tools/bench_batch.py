#!/usr/bin/env python3
"""BASELINE configs[3]: the 8-mesh batch (the 6 bundled shapes/*.obj + 2 repeats: only 6 exist upstream), 6 views
each, through MeshBatchPainter (contexture_nerf_amd/batch.py): items (mesh, view) dealt round-robin over the ranks, one
all-reduce(MAX) + one all-reduce(SUM) per mesh.  Runs on 1 GPU as is, or under torchrun on N.  Prints one JSON object on rank 0.
Usage: python tools/bench_batch.py [--image 768] [--steps 50] [--meshes 8] [--in-flight 3]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contexture_nerf_amd import config as CFG, dist as D
from contexture_nerf_amd.trainer import ConTEXTure
from contexture_nerf_amd.batch import MeshBatchPainter
from contexture_nerf_amd.stable_diffusion_depth import StableDiffusion

ap = argparse.ArgumentParser()
ap.add_argument("--image", type=int, default=768)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--meshes", type=int, default=8)
ap.add_argument("--in-flight", type=int, default=3)
ap.add_argument("--per-eval", type=int, default=6, help="views denoised in lockstep as one UNet evaluation (0: --in-flight streams)")
a = ap.parse_args()
rank, world, dev = D.init()
names = ["nascar", "spot_triangulated", "bunny", "blub_no_texture", "sphere", "env_sphere"]
names = (names + names)[:a.meshes]
sd = StableDiffusion(dev)
trainers = []
for nm in names:
    cfg = CFG.TrainConfig()
    cfg.guide.text = f"a photo of a {nm}"
    cfg.guide.shape_path = f"shapes/{nm}.obj"
    cfg.guide.guidance_scale = 10.0
    cfg.guide.sd_image_size = a.image
    cfg.guide.num_inference_steps = a.steps
    cfg.optim.views_in_flight = a.in_flight
    cfg.optim.views_per_eval = a.per_eval if a.per_eval > 1 else 0
    tr = ConTEXTure(cfg, device=dev, diffusion=sd)
    tr.text_z = sd.get_text_embeds([cfg.guide.text])
    trainers.append(tr)
bp = MeshBatchPainter(trainers)
if world > 1:
    D.dist.barrier()
torch.cuda.synchronize()
t = time.perf_counter()
res = bp.paint_all()
torch.cuda.synchronize()
if world > 1:
    D.dist.barrier()
dt = time.perf_counter() - t
if rank == 0:
    print(json.dumps({"metric": "sec per mesh batch (BASELINE configs[3])", "meshes": names, "views_per_mesh": 6, "n_gpus": world,
                      "items_per_rank": [len(p) for p in bp.plan], "image": a.image, "plms_steps": a.steps,
                      "views_in_flight": a.in_flight, "views_per_eval": a.per_eval, "sec_total": round(dt, 3), "sec_per_mesh": round(dt / len(names), 3),
                      "coverage": [round(float((c > 0).float().mean()), 4) for _, c in res],
                      "finite": bool(all(torch.isfinite(at).all() for at, _ in res)),
                      "data": "synthetic (random-init weights, seeded text embeddings; cold: includes first-touch of the workspaces)"}))
