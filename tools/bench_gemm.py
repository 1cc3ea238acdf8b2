#!/usr/bin/env python3
"""Micro-benchmark of the fp16 MFMA GEMM / implicit-GEMM conv kernels on the SD2-depth UNet's own layer list
(latent 96^2, CFG batch 2): every distinct (kind, shape) with the number of times one UNet evaluation launches it, the
split-K the executor would choose, bias + residual epilogues on.  The weighted sum estimates the GEMM time of one step.
Random operands, device-timed back-to-back launches, interleaved rounds in one process.

  python tools/bench_gemm.py [rounds] [subset]       subset: all | conv | lin | small | comma-separated indices
Variants are selected by the CTX_GEMM_* environment switches (one process per variant)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contexture_nerf_amd import _lib as L

lib = L.load()
dev = torch.device('cuda:0')
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
subset = sys.argv[2] if len(sys.argv) > 2 else "all"

# ("conv", (B,H,W) of the INPUT, Cout, Cin, flags, count)   flags: 1 = stride 2, 2 = fused x2 upsample
# ("lin", M, N, K, epi, count)                               epi 1 = GEGLU
S = [
    ("conv", (2, 96, 96), 320, 320, 0, 7), ("conv", (2, 96, 96), 320, 640, 0, 2), ("conv", (2, 96, 96), 320, 960, 0, 1),
    ("conv", (2, 48, 48), 640, 320, 0, 1), ("conv", (2, 48, 48), 640, 640, 0, 6), ("conv", (2, 48, 48), 640, 960, 0, 1),
    ("conv", (2, 48, 48), 640, 1280, 0, 1), ("conv", (2, 48, 48), 640, 1920, 0, 1),
    ("conv", (2, 24, 24), 1280, 640, 0, 1), ("conv", (2, 24, 24), 1280, 1280, 0, 6), ("conv", (2, 24, 24), 1280, 1920, 0, 1),
    ("conv", (2, 24, 24), 1280, 2560, 0, 2),
    ("conv", (2, 12, 12), 1280, 1280, 0, 11), ("conv", (2, 12, 12), 1280, 2560, 0, 3),
    ("conv", (2, 96, 96), 320, 320, 1, 1), ("conv", (2, 48, 48), 640, 640, 1, 1), ("conv", (2, 24, 24), 1280, 1280, 1, 1),
    ("conv", (2, 12, 12), 1280, 1280, 2, 1), ("conv", (2, 24, 24), 1280, 1280, 2, 1), ("conv", (2, 48, 48), 640, 640, 2, 1),
    # linear layers
    ("lin", 18432, 320, 320, 0, 25), ("lin", 18432, 960, 320, 0, 5), ("lin", 18432, 2560, 320, 1, 5), ("lin", 18432, 320, 1280, 0, 5),
    ("lin", 18432, 320, 960, 0, 1), ("lin", 18432, 320, 640, 0, 2),
    ("lin", 4608, 640, 640, 0, 25), ("lin", 4608, 1920, 640, 0, 5), ("lin", 4608, 5120, 640, 1, 5), ("lin", 4608, 640, 2560, 0, 5),
    ("lin", 4608, 640, 320, 0, 1), ("lin", 4608, 640, 1920, 0, 1), ("lin", 4608, 640, 1280, 0, 1), ("lin", 4608, 640, 960, 0, 1),
    ("lin", 1152, 1280, 1280, 0, 25), ("lin", 1152, 3840, 1280, 0, 5), ("lin", 1152, 10240, 1280, 1, 5), ("lin", 1152, 1280, 5120, 0, 5),
    ("lin", 1152, 1280, 640, 0, 1), ("lin", 1152, 1280, 2560, 0, 2), ("lin", 1152, 1280, 1920, 0, 1),
    ("lin", 288, 1280, 1280, 0, 5), ("lin", 288, 3840, 1280, 0, 1), ("lin", 288, 10240, 1280, 1, 1), ("lin", 288, 1280, 5120, 0, 1),
    ("lin", 288, 1280, 2560, 0, 3),
    ("lin", 154, 24960, 1024, 0, 1),
]
if subset == "conv":
    S = [s for s in S if s[0] == "conv"]
elif subset == "lin":
    S = [s for s in S if s[0] == "lin"]
elif subset == "small":
    S = [s for s in S if s[0] == "lin" and s[5] == 25]
elif subset != "all":
    S = [S[int(i)] for i in subset.split(",")]

g = torch.Generator(device=dev).manual_seed(0)
part = torch.empty(256 << 20, dtype=torch.uint8, device=dev)          # split-K slabs
items = []
for s_ in S:
    if s_[0] == "conv":
        (B, H, W), N, Cin, flags, cnt = s_[1:]
        st, up = (2 if flags & 1 else 1), (1 if flags & 2 else 0)
        Ho, Wo = ((H << up) - 1) // st + 1, ((W << up) - 1) // st + 1
        M, K = B * Ho * Wo, 9 * Cin
        x = torch.randn(B, H, W, Cin, generator=g, device=dev).half()
        cb = (B, H, W, Cin, flags)
        epi = 0
    else:
        M, N, K, epi, cnt = s_[1:]
        x = torch.randn(M, K, generator=g, device=dev).half()
        cb = (0, 0, 0, 0, 0)
    w = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).half()
    No = N // 2 if epi else N
    y = torch.empty(M, No, dtype=torch.float16, device=dev)
    res = torch.randn(M, No, generator=g, device=dev).half() if not epi else None
    bias = torch.randn(N, generator=g, device=dev).half()
    items.append((s_, M, N, K, epi, cnt, x, w, y, res, bias, cb))

times = {i: [] for i in range(len(items))}
for r_ in range(rounds):
    for i, (s_, M, N, K, epi, cnt, x, w, y, res, bias, cb) in enumerate(items):
        ms = lib.ctx_bench_gemm(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(res), M, N, K, L.ptr(y), *cb, epi, L.ptr(part), -1, 10, L.stream())
        assert ms > 0, lib.ctx_last_error()
        times[i].append(ms)
tot_fl = tot_ms = 0.0
by = {"conv": [0.0, 0.0], "lin": [0.0, 0.0]}
for i, (s_, M, N, K, epi, cnt, *_rest) in enumerate(items):
    ms = sorted(times[i])[len(times[i]) // 2]
    fl = 2.0 * M * N * K
    tot_fl += fl * cnt; tot_ms += ms * cnt
    by[s_[0]][0] += fl * cnt; by[s_[0]][1] += ms * cnt
    print(f"{i:2d} {str(s_[:-1]):44s} x{cnt:2d} {fl / 1e9:8.2f} GF {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF/s  ({ms * cnt * 1e3:7.0f} us/step)")
env = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("CTX_"))
for k, (fl, ms) in by.items():
    if ms > 0:
        print(f"[{env}] {k}: {ms:.3f} ms/step, {fl / ms / 1e9:.1f} TFLOP/s")
print(f"[{env}] total: {tot_ms:.3f} ms/step  {tot_fl / 1e12:.3f} TFLOP  {tot_fl / tot_ms / 1e9:.1f} TFLOP/s")
