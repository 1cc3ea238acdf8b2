#!/usr/bin/env python3
"""Two independent UNet evaluations (two views) on one GPU: back to back on one stream vs concurrently on two HIP
streams.  Tells whether painting two views per GPU at once pays (kernels of the deep UNet levels do not fill the chip).
Usage: python tools/bench_concurrent.py [latent] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contexture_nerf_amd.unet import UNet2DConditionModel

S = int(sys.argv[1]) if len(sys.argv) > 1 else 96
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device('cuda:0')
nets = [UNet2DConditionModel(device=dev, seed=0), UNet2DConditionModel(device=dev, seed=0)]
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.randn(2, 5, S, S, generator=g, device=dev) for _ in range(2)]
ctx = [torch.randn(2, 77, 1024, generator=g, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for k in range(2):
    nets[k](xs[k], 500.0, encoder_hidden_states=ctx[k])
torch.cuda.synchronize()

def seq():
    for _ in range(iters):
        for k in range(2):
            nets[k](xs[k], 500.0, encoder_hidden_states=ctx[k])
    torch.cuda.synchronize()

def conc():
    for _ in range(iters):
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                nets[k](xs[k], 500.0, encoder_hidden_states=ctx[k])
    torch.cuda.synchronize()

for name, fn in (("sequential (one stream)", seq), ("concurrent (two streams)", conc), ("sequential (one stream)", seq), ("concurrent (two streams)", conc)):
    fn()
    t = time.perf_counter(); fn(); dt = time.perf_counter() - t
    print(f"{name}: {dt / (2 * iters) * 1e3:.3f} ms per UNet evaluation ({2 * iters / dt:.1f} evals/s)")
