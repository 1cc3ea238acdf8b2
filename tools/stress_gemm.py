#!/usr/bin/env python3
"""Race screen for the GEMM / conv kernels: every tile form of gemm.hip, gemm8.hip, every form of gemm144.hip and the halo conv,
with split-K 1 / 2 / 4, five launches each on UNet-sized problems, each result checked against torch.  An LDS-DMA ring that is
read or refilled one barrier too early fails here intermittently (that is how the missing lgkmcnt(0) of the steady K loop showed:
929 wrong elements in one launch out of a few)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F
from contexture_nerf_amd import _lib as L
lib = L.load(); dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
def check(y, want):
    err = (y.float() - want).abs(); tol = 4e-3 + 3e-3 * want.abs()
    return int((err > tol).sum())
# linear layers: every forced kernel, split-K 1 / 2 / 4, five repetitions
for (M, N, K) in [(4608, 640, 640), (18432, 320, 320), (1152, 1280, 1280), (4608, 640, 2560)]:
    x = torch.randn(M, K, generator=g, device=dev).half(); w = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).half()
    b = torch.randn(N, generator=g, device=dev).half(); r = torch.randn(M, N, generator=g, device=dev).half()
    want = x.float() @ w.float().T + b.float() + r.float()
    part = torch.empty(4 * M * N, dtype=torch.float32, device=dev)
    bad = []
    for (tile, u8) in [(t, 0) for t in range(28)] + [(-1, 1), (-1, 4), (-1, 5), (-1, 6), (-1, 7), (-1, 8)]:
        for S in (1, 2, 4):
            lib.ctx_gemm_tune(tile, u8)
            worst = 0
            for rep in range(5):
                y = torch.zeros(M, N, dtype=torch.float16, device=dev)
                ms = lib.ctx_bench_gemm(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(r), M, N, K, L.ptr(y), 0, 0, 0, 0, 0, 0, L.ptr(part), S, 1, L.stream())
                assert ms > 0
                worst = max(worst, check(y, want))
            if worst: bad.append((tile, u8, S, worst))
    print("lin", M, N, K, "bad:", bad, flush=True)
# convolutions
for (B, H, W, Cin, Cout, flags) in [(2, 48, 48, 640, 640, 0), (2, 96, 96, 320, 320, 0), (2, 24, 24, 1280, 1280, 2), (2, 48, 48, 640, 640, 1)]:
    st, up = (2 if flags & 1 else 1), (1 if flags & 2 else 0)
    Ho, Wo = ((H << up) - 1) // st + 1, ((W << up) - 1) // st + 1
    M, K, N = B * Ho * Wo, 9 * Cin, Cout
    x = torch.randn(B, H, W, Cin, generator=g, device=dev).half(); w = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).half()
    b = torch.randn(N, generator=g, device=dev).half(); r = torch.randn(M, N, generator=g, device=dev).half()
    xin = x.float().permute(0, 3, 1, 2)
    if up: xin = F.interpolate(xin, scale_factor=2.0, mode='nearest')
    wt = w.float().view(N, 3, 3, Cin).permute(0, 3, 1, 2)
    want = F.conv2d(xin, wt, b.float(), stride=st, padding=1).permute(0, 2, 3, 1).reshape(M, N) + r.float()
    part = torch.empty(4 * M * N, dtype=torch.float32, device=dev)
    bad = []
    for (tile, u8) in [(t, 0) for t in (1, 5, 10, 12, 14, 15, 19, 20, 22, 25)] + [(-1, 1), (-1, 5), (-1, 6), (-1, 7), (-1, 8)] + ([(-1, 2), (-1, 3)] if flags == 0 else []):
        for S in (1, 2, 4):
            lib.ctx_gemm_tune(tile, u8)
            worst = 0
            for rep in range(5):
                y = torch.zeros(M, N, dtype=torch.float16, device=dev)
                ms = lib.ctx_bench_gemm(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(r), M, N, K, L.ptr(y), B, H, W, Cin, flags, 0, L.ptr(part), S, 1, L.stream())
                assert ms > 0
                worst = max(worst, check(y, want))
            if worst: bad.append((tile, u8, S, worst))
    print("conv", (B, H, W, Cin, Cout, flags), "bad:", bad, flush=True)
lib.ctx_gemm_tune(-1, -1)
