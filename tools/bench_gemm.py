#!/usr/bin/env python3
"""Micro-benchmark of the fp16 MFMA GEMM / implicit-GEMM conv kernel on the SD2-depth UNet's own shapes
(latent 96^2, CFG batch 2).  Interleaved rounds in one process (guide rule 24), random operands (rule 25).
Usage: python tools/bench_gemm.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contexture_nerf_amd import _lib as L

lib = L.load()
dev = torch.device('cuda:0')
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
only = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else None
# (kind, M or (B,H,W), N/Cout, K/Cin)
shapes = [("conv", (2, 96, 96), 320, 320), ("conv", (2, 48, 48), 640, 640), ("conv", (2, 24, 24), 1280, 1280),
          ("conv", (2, 12, 12), 1280, 1280), ("conv", (2, 96, 96), 320, 640), ("conv", (2, 48, 48), 640, 1280),
          ("conv", (2, 24, 24), 1280, 2560), ("conv", (2, 96, 96), 320, 960),
          ("gemm", 18432, 320, 320), ("gemm", 18432, 960, 320), ("gemm", 18432, 2560, 320), ("gemm", 18432, 320, 1280),
          ("gemm", 4608, 640, 640), ("gemm", 4608, 5120, 640), ("gemm", 4608, 640, 2560),
          ("gemm", 1152, 1280, 1280), ("gemm", 1152, 10240, 1280), ("gemm", 1152, 1280, 5120), ("gemm", 154, 640, 1024),
          ("gemm", 18432, 320, 32), ("gemm", 18432, 320, 64), ("gemm", 18432, 320, 128), ("gemm", 18432, 320, 640),
          ("gemm", 18432, 128, 320), ("gemm", 18432, 640, 320), ("gemm", 4608, 320, 320), ("gemm", 36864, 320, 320)]
if only is not None:
    shapes = [shapes[i] for i in only]
g = torch.Generator(device=dev).manual_seed(0)
with_res = os.environ.get("BENCH_RES", "0") == "1"
tot_fl = tot_ms = 0
for s_ in shapes:
    if s_[0] == "conv":
        (B, H, W), N, Cin = s_[1], s_[2], s_[3]
        M, K = B * H * W, 9 * Cin
        x = torch.randn(B, H, W, Cin, generator=g, device=dev).half()
        cb = (B, H, W, Cin)
    else:
        M, N, K = s_[1], s_[2], s_[3]
        x = torch.randn(M, K, generator=g, device=dev).half()
        cb = (0, 0, 0, 0)
    w = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).half()
    y = torch.empty(M, N, dtype=torch.float16, device=dev)
    res = torch.randn(M, N, generator=g, device=dev).half() if with_res else None
    bias = torch.randn(N, generator=g, device=dev).half() if with_res else None
    fl = 2.0 * M * N * K
    times = []
    for r_ in range(rounds):
        ms = lib.ctx_bench_gemm(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(res), M, N, K, L.ptr(y), *cb, None, 1, 20, L.stream())
        assert ms > 0, lib.ctx_last_error()
        times.append(ms)
    ms = sorted(times)[len(times) // 2]
    tot_fl += fl; tot_ms += ms
    print(f"{str(s_):45s} {fl / 1e9:9.2f} GFLOP  median {ms * 1e3:8.1f} us  {fl / ms / 1e9:8.1f} TFLOP/s  (min {min(times) * 1e3:.1f} us)")
print(f"tile={os.environ.get('CTX_GEMM_TILE', 'auto')} res={with_res}  sum: {tot_fl / tot_ms / 1e9:.1f} TFLOP/s")
sys.exit(0)
g = torch.Generator(device=dev).manual_seed(0)
items = []
for s in shapes:
    if s[0] == "conv":
        (B, H, W), Cout, Cin = s[1], s[2], s[3]
        x = torch.randn(B, H, W, Cin, generator=g, device=dev).half()
        w = (torch.randn(Cout, 3, 3, Cin, generator=g, device=dev) / (9 * Cin) ** 0.5).half()
        y = torch.empty(B, H, W, Cout, dtype=torch.float16, device=dev)
        fl = 2.0 * B * H * W * Cout * 9 * Cin
        call = lambda x=x, w=w, y=y, B=B, H=H, W=W, Cin=Cin, Cout=Cout: lib.ctx_conv3x3_f16(
            L.ptr(x), L.ptr(w), None, None, None, B, H, W, Cin, Cout, 1, 0, L.ptr(y), L.stream())
    else:
        M, N, K = s[1], s[2], s[3]
        x = torch.randn(M, K, generator=g, device=dev).half()
        w = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).half()
        y = torch.empty(M, N, dtype=torch.float16, device=dev)
        fl = 2.0 * M * N * K
        call = lambda x=x, w=w, y=y, M=M, N=N, K=K: lib.ctx_gemm_f16(L.ptr(x), L.ptr(w), None, None, M, N, K, L.ptr(y), L.stream())
    items.append((s, fl, call))
best = {}
for r in range(rounds + 1):
    for s, fl, call in items:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        call(); torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            L.check(call())
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        if r > 0:
            best.setdefault(s, []).append(ms)
tot_fl = tot_ms = 0
for s, fl, _ in items:
    ms = sorted(best[s])[len(best[s]) // 2]
    tot_fl += fl; tot_ms += ms
    print(f"{str(s):45s} {fl / 1e9:9.2f} GFLOP  median {ms * 1e3:8.1f} us  {fl / ms / 1e9:8.1f} TFLOP/s  (min {min(best[s]) * 1e3:.1f} us)")
print(f"impl={os.environ.get('CTX_GEMM_IMPL', '1')}  sum: {tot_fl / tot_ms / 1e9:.1f} TFLOP/s")
