#!/usr/bin/env python3
"""One GEMM / conv shape through a forced kernel, N launches (for rocprofv3 --pmc runs on a single kernel).
Usage: one_gemm.py USE8 TILE B H W Cin Cout FLAGS SPLITK ITERS   (B = 0: linear layer with M=H, K=Cin, N=Cout)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contexture_nerf_amd import _lib as L

use8, tile, B, H, W, Cin, Cout, flags, S, iters = [int(v) for v in sys.argv[1:11]]
lib = L.load(); dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
if B:
    st, up = (2 if flags & 1 else 1), (1 if flags & 2 else 0)
    Ho, Wo = ((H << up) - 1) // st + 1, ((W << up) - 1) // st + 1
    M, K, N = B * Ho * Wo, 9 * Cin, Cout
    x = torch.randn(B, H, W, Cin, generator=g, device=dev).half(); cb = (B, H, W, Cin, flags)
else:
    M, K, N = H, Cin, Cout
    x = torch.randn(M, K, generator=g, device=dev).half(); cb = (0, 0, 0, 0, 0)
w = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).half()
y = torch.empty(M, N, dtype=torch.float16, device=dev)
res = torch.randn(M, N, generator=g, device=dev).half()
bias = torch.randn(N, generator=g, device=dev).half()
part = torch.empty(max(S, 1) * M * N, dtype=torch.float32, device=dev)
lib.ctx_gemm_tune(tile, use8)
ms = lib.ctx_bench_gemm(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(res), M, N, K, L.ptr(y), *cb, 0, L.ptr(part), S, iters, L.stream())
print(f"M={M} N={N} K={K} use8={use8} tile={tile} S={S}: {ms * 1e3:.1f} us  {2.0 * M * N * K / ms / 1e9:.1f} TF/s")
