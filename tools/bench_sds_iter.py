#!/usr/bin/env python3
"""Denoise side of one Zero123++ SDS iteration at the reference's sizes (src/training/trainer.py:700-850): six 320x320 renders ->
960x640 grid -> VAE encode -> noise at a DreamTime t -> pipeline one step (condition 'w' pass, depth ControlNet, main 'r' pass,
CFG 10) -> v target, gradient, targets.  SD2-architecture UNet / ControlNet (4 latent channels), SD VAE, random init, synthetic
inputs; the texture render before it and the backward after it are not part of this figure."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, importlib
U = importlib.import_module('contexture_nerf_amd.unet')
V = importlib.import_module('contexture_nerf_amd.vae')
S = importlib.import_module('contexture_nerf_amd.scheduler')
Z = importlib.import_module('contexture_nerf_amd.zero123plus')
sds = importlib.import_module('contexture_nerf_amd.sds')
ut = importlib.import_module('contexture_nerf_amd.utils')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device('cuda:0')
cfg = dict(U.SD2_DEPTH); cfg['in_channels'] = 4
net = U.UNet2DConditionModel(cfg, device=dev, seed=0); cnet = U.ControlNetModel(cfg, device=dev, seed=1); vae = V.AutoencoderKL(device=dev, seed=2)
train_sched, val_sched = S.DDPMScheduler(), S.EulerAncestralDiscreteScheduler()
pipe = Z.Zero123PlusPipeline(vae, Z.DepthControlUNet(Z.RefOnlyNoisedUNet(net, train_sched, val_sched).eval(), cnet, conditioning_scale=2.0).eval(), val_sched)
g = torch.Generator(device=dev).manual_seed(0)
six = torch.rand(6, 3, 320, 320, generator=g, device=dev)
cond = torch.rand(1, 3, 320, 320, generator=g, device=dev) * 2 - 1
depth = torch.rand(1, 3, 960, 640, generator=g, device=dev)
pe = torch.randn(1, 77, 1024, generator=g, device=dev)
dt_sched = ut.DreamTimeScheduler(train_sched.alphas_cumprod, 5000)
avg = None


def it(i):
    global avg
    r = sds.sds_iteration_targets(pipe, six, cond, depth, pe, dt_sched.get_t(i), train_sched.alphas_cumprod, train_sched.add_noise,
                                  ikl_running_avg=avg)
    avg = r['ikl_running_avg']
    return r


for i in range(3):
    r = it(i * 100)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(iters):
    r = it(300 + i * 100)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(json.dumps({"sds_iteration_denoise_side_ms": round(dt * 1e3, 2), "iterations_per_s": round(1 / dt, 2),
                  "finite": bool(torch.isfinite(r['targets']).all()), "latent": list(r['z0'].shape),
                  "note": "VAE encode (960x640) + condition-image encodes + 'w' pass + ControlNet + 'r' pass (CFG 10) + SDS arithmetic; "
                          "incl. two 320x320 condition-image VAE encodes per call as the pipeline does"}))
