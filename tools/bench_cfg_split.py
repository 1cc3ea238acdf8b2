#!/usr/bin/env python3
"""One CFG-batched UNet evaluation as (a) one B=2 call, (b) two B=1 calls on two HIP streams (engines share the weights)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contexture_nerf_amd.unet import UNet2DConditionModel
S = int(sys.argv[1]) if len(sys.argv) > 1 else 96
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda:0')
base = UNet2DConditionModel(device=dev, seed=0)
nets = [base, base.clone_shared()]
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(2, 5, S, S, generator=g, device=dev); ctx = torch.randn(2, 77, 1024, generator=g, device=dev)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def b2():
    for _ in range(iters): nets[0](x, 500.0, encoder_hidden_states=ctx)
    torch.cuda.synchronize()
def b1x2():
    for _ in range(iters):
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                nets[k](x[k:k + 1], 500.0, encoder_hidden_states=ctx[k:k + 1])
    torch.cuda.synchronize()
for name, fn in (("one B=2 call", b2), ("two B=1 calls, two streams", b1x2), ("one B=2 call", b2), ("two B=1 calls, two streams", b1x2)):
    fn(); t = time.perf_counter(); fn(); dt = time.perf_counter() - t
    print(f"{name}: {dt / iters * 1e3:.3f} ms per CFG-batched evaluation")
