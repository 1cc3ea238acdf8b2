#!/usr/bin/env python3
"""Zero123++-shaped denoise evaluation (SURVEY §8d cfg 3): the SD2-architecture UNet (4 input channels) on the 3x2 view-grid
latent [2,4,120,80] (CFG pair) with reference-only attention — a 'w' pass over the noised 40x40 condition latent (1 600 tokens
at level 0), the depth ControlNet over the 960x640 depth grid image (conditioning scale 2, as src/training/trainer.py:302-304), and
an 'r' pass whose self-attention K/V carry the parked tokens for the conditional row and whose skip tensors / mid output take the
ControlNet residuals: the three passes of one SDS iteration's denoise.  Random-init weights, synthetic inputs."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, importlib
U = importlib.import_module('contexture_nerf_amd.unet')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device('cuda:0')
cfg = dict(U.SD2_DEPTH); cfg['in_channels'] = 4
net = U.UNet2DConditionModel(cfg, device=dev, seed=0)
cnet = U.ControlNetModel(cfg, device=dev, seed=1)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(2, 4, 120, 80, generator=g, device=dev)
cond = torch.randn(1, 4, 40, 40, generator=g, device=dev)
ctx = torch.randn(2, 77, 1024, generator=g, device=dev)
depth = torch.rand(2, 3, 960, 640, generator=g, device=dev)
bank = None


def step():
    global bank
    _, bank = net.forward_ref(cond, 500.0, ctx[1:], 'w', bank=bank)
    res, _ = cnet(x, 500.0, encoder_hidden_states=ctx, controlnet_cond=depth, conditioning_scale=2.0)
    with net.residuals(res):
        return net.forward_ref(x, 500.0, ctx, 'r', bank=bank, ref_row0=1)[0]['sample']


for _ in range(3):
    y = step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(iters):
    y = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / iters
e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
e[0].record(); _, bank = net.forward_ref(cond, 500.0, ctx[1:], 'w', bank=bank); e[1].record()
res, _ = cnet(x, 500.0, encoder_hidden_states=ctx, controlnet_cond=depth, conditioning_scale=2.0); e[2].record()
with net.residuals(res):
    y = net.forward_ref(x, 500.0, ctx, 'r', bank=bank, ref_row0=1)[0]['sample']
e[3].record()
torch.cuda.synchronize()
depth2 = depth.clone()                                                     # a new conditioning image: the embedding is recomputed
e2 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
e2[0].record(); cnet(x, 500.0, encoder_hidden_states=ctx, controlnet_cond=depth2, conditioning_scale=2.0); e2[1].record()
torch.cuda.synchronize()
fl_main = sum(v[1] for v in net.flops(2, 120, 80, 77).values())          # plain forward; the extra K/V tokens add attention work
print(json.dumps({"ms_per_iteration": round(dt * 1e3, 3), "iterations_per_s": round(1 / dt, 2),
                  "cond_write_pass_ms": round(e[0].elapsed_time(e[1]), 3), "controlnet_ms": round(e[1].elapsed_time(e[2]), 3),
                  "controlnet_with_new_depth_image_ms": round(e2[0].elapsed_time(e2[1]), 3),
                  "main_read_pass_ms": round(e[2].elapsed_time(e[3]), 3),
                  "plain_forward_tflop_batch2_120x80": round(fl_main / 1e12, 3), "finite": bool(torch.isfinite(y).all()),
                  "note": "CFG batch 2 on the main and ControlNet passes, batch 1 condition pass"}))
