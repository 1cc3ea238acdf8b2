#!/usr/bin/env python3
"""Diagnostic: the GEMM kernels on large square problems (where tile quantisation, prologue and epilogue vanish), to separate the
steady-state rate of each kernel's K loop from the UNet's shape effects.  Usage: python tools/bench_gemm_square.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contexture_nerf_amd import _lib as L
lib = L.load(); dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
part = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
for (M, N, K) in ((4096, 4096, 4096), (8192, 8192, 8192), (18432, 2560, 1280), (18432, 1280, 5760)):
    x = torch.randn(M, K, generator=g, device=dev).half()
    w = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).half()
    y = torch.empty(M, N, dtype=torch.float16, device=dev)
    for name, tile, u8 in (("gemm8 256x256", -1, 1), ("pipe 256x128 16w", 10, 0), ("pipe 256x128 8w", 12, 0), ("pipe 128x128 8w r3", 14, 0),
                           ("pipe 128x128 8w r2", 20, 0), ("pipe 64x64 4w", 15, 0)):
        lib.ctx_gemm_tune(tile, u8)
        ts = [lib.ctx_bench_gemm(L.ptr(x), L.ptr(w), None, None, M, N, K, L.ptr(y), 0, 0, 0, 0, 0, 0, L.ptr(part), 1, 5, L.stream()) for _ in range(3)]
        t = min(ts)
        print(f"{M}x{N}x{K} {name:20s} {t * 1e3:8.1f} us  {2.0 * M * N * K / t / 1e9:7.1f} TF/s", flush=True)
lib.ctx_gemm_tune(-1, -1)
