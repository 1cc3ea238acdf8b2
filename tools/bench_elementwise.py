#!/usr/bin/env python3
"""Device time of the UNet's HBM-bound kernels (GroupNorm, LayerNorm) at their in-situ sizes (latent 96^2, CFG batch 2),
with a torch copy of the same byte count as the bandwidth reference.  Usage: python tools/bench_elementwise.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contexture_nerf_amd import _lib as L
lib = L.load(); dev = torch.device('cuda:0')

def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

g = torch.Generator(device=dev).manual_seed(0)
ws = torch.empty(lib.ctx_groupnorm_ws_bytes(2, 32), dtype=torch.uint8, device=dev)
tot = {}
# (B, HW, C, count per UNet eval)
for B, HW, C, cnt in [(2, 9216, 320, 19), (2, 9216, 640, 2), (2, 9216, 960, 1), (2, 2304, 640, 17), (2, 2304, 320, 1), (2, 2304, 960, 1),
                      (2, 2304, 1280, 1), (2, 2304, 1920, 1), (2, 576, 1280, 17), (2, 576, 640, 1), (2, 576, 1920, 1), (2, 576, 2560, 2),
                      (2, 144, 1280, 13), (2, 144, 2560, 3)]:
    x = torch.randn(B, HW, C, generator=g, device=dev).half(); y = torch.empty_like(x)
    ga = torch.ones(C, device=dev).half(); be = torch.zeros(C, device=dev).half()
    t = timeit(lambda: lib.ctx_groupnorm_f16(L.ptr(x), L.ptr(ga), L.ptr(be), B, HW, C, 32, 1e-5, 1, L.ptr(y), L.ptr(ws), L.stream()))
    tc = timeit(lambda: y.copy_(x))
    mb = x.numel() * 2 / 1e6
    print(f"groupnorm B={B} HW={HW} C={C}: {t:7.1f} us (2 kernels; {3 * mb:.1f} MB moved -> {3 * mb / t * 1e-3:.2f} TB/s)   torch copy {tc:6.1f} us ({2 * mb / tc * 1e-3:.2f} TB/s)  x{cnt}")
    tot['gn'] = tot.get('gn', 0) + t * cnt
for rows, C, cnt in [(18432, 320, 15), (4608, 640, 15), (1152, 1280, 15), (288, 1280, 3)]:
    x = torch.randn(rows, C, generator=g, device=dev).half(); y = torch.empty_like(x)
    ga = torch.ones(C, device=dev).half(); be = torch.zeros(C, device=dev).half()
    t = timeit(lambda: lib.ctx_layernorm_f16(L.ptr(x), L.ptr(ga), L.ptr(be), rows, C, 1e-5, L.ptr(y), L.stream()))
    tc = timeit(lambda: y.copy_(x))
    mb = x.numel() * 2 / 1e6
    print(f"layernorm rows={rows} C={C}: {t:7.1f} us ({2 * mb / t * 1e-3:.2f} TB/s)   torch copy {tc:6.1f} us ({2 * mb / tc * 1e-3:.2f} TB/s)  x{cnt}")
    tot['ln'] = tot.get('ln', 0) + t * cnt
print({k: round(v) for k, v in tot.items()}, "us per UNet eval (wall, includes launch gaps)")
