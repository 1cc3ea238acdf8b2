#!/usr/bin/env python3
"""N independent UNet evaluations (N views) on one GPU: back to back on one stream vs concurrently on N HIP streams
(engines share one weight blob).  Tells how many views per GPU to keep in flight (kernels of the deep UNet levels do not
fill the chip).  Usage: python tools/bench_concurrent.py [latent] [iters] [batch]   (batch 12: lockstep groups of six views in flight)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contexture_nerf_amd.unet import UNet2DConditionModel

S = int(sys.argv[1]) if len(sys.argv) > 1 else 96
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
BT = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device('cuda:0')
base = UNet2DConditionModel(device=dev, seed=0)
nets = [base] + [base.clone_shared() for _ in range(3)]
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.randn(BT, 5, S, S, generator=g, device=dev) for _ in range(4)]
ctx = [torch.randn(BT, 77, 1024, generator=g, device=dev) for _ in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]
for k in range(4):
    nets[k](xs[k], 500.0, encoder_hidden_states=ctx[k])
torch.cuda.synchronize()

def run(n, concurrent):
    for _ in range(iters):
        for k in range(n):
            if concurrent:
                with torch.cuda.stream(streams[k]):
                    nets[k](xs[k], 500.0, encoder_hidden_states=ctx[k])
            else:
                nets[k](xs[k], 500.0, encoder_hidden_states=ctx[k])
    torch.cuda.synchronize()

for n, conc in ((1, False), (2, True), (3, True), (4, True), (1, False), (2, True)):
    run(n, conc)
    t = time.perf_counter(); run(n, conc); dt = time.perf_counter() - t
    print(f"{n} view(s) {'concurrent' if conc else 'serial'}: {dt / (n * iters) * 1e3:.3f} ms per UNet evaluation ({n * iters / dt:.1f} evals/s)" + (f" [batch {BT}: {n * iters * BT / 2 / dt:.1f} view-steps/s]" if BT != 2 else ""))
