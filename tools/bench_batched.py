#!/usr/bin/env python3
"""One UNet evaluation over V views batched (batch 2V, CFG pairs) against V evaluations of batch 2: milliseconds per view-step.
Usage: python tools/bench_batched.py [latent] [iters] [views,views,...]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contexture_nerf_amd.unet import UNet2DConditionModel

S = int(sys.argv[1]) if len(sys.argv) > 1 else 96
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device('cuda:0')
net = UNet2DConditionModel(device=dev, seed=0)
g = torch.Generator(device=dev).manual_seed(0)
views = tuple(int(v) for v in sys.argv[3].split(',')) if len(sys.argv) > 3 else (1, 2, 3, 6)
for V in views:
    x = torch.randn(2 * V, 5, S, S, generator=g, device=dev)
    ctx = torch.randn(2 * V, 77, 1024, generator=g, device=dev)
    for _ in range(2):
        net(x, 500.0, encoder_hidden_states=ctx)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        net(x, 500.0, encoder_hidden_states=ctx)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / iters
    print(json.dumps({"views_batched": V, "batch": 2 * V, "latent": S, "ms_per_evaluation": round(dt * 1e3, 3),
                      "ms_per_view_step": round(dt * 1e3 / V, 3), "view_steps_per_s": round(V / dt, 1)}), flush=True)
