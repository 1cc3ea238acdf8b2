#!/usr/bin/env python3
"""How much of a small projection's in-step time is cold operands: one GEMM timed hot (back to back), with everything evicted
(512 MB written in between), and with only the activations / only the weights re-touched after the eviction."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contexture_nerf_amd import _lib as L
lib = L.load(); dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for (M, N, K) in [(1152, 1280, 1280), (4608, 640, 640), (18432, 320, 320), (1152, 1280, 5120), (288, 1280, 1280)]:
    x = torch.randn(M, K, generator=g, device=dev).half()
    w = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).half()
    b = torch.randn(N, generator=g, device=dev).half(); r = torch.randn(M, N, generator=g, device=dev).half()
    y = torch.empty(M, N, dtype=torch.float16, device=dev)
    part = torch.empty(16 * M * N, dtype=torch.float32, device=dev)
    def run():
        lib.ctx_bench_gemm(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(r), M, N, K, L.ptr(y), 0, 0, 0, 0, 0, 0, L.ptr(part), -1, 1, L.stream())
    def timed(prep):
        ts = []
        for _ in range(12):
            prep()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.check(lib.ctx_gemm_f16(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(r), M, N, K, L.ptr(y), L.stream()))
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]
    hot = timed(lambda: None)
    cold = timed(lambda: big.fill_(1))
    xhot = timed(lambda: (big.fill_(1), x.add_(0), r.add_(0)))
    whot = timed(lambda: (big.fill_(1), w.add_(0)))
    print(f"M={M} N={N} K={K}: hot {hot:6.1f} us | all cold {cold:6.1f} | activations re-touched {xhot:6.1f} | weights re-touched {whot:6.1f}", flush=True)
