#!/usr/bin/env python3
"""Attention micro-benchmark at the UNet's shapes. Usage: python tools/bench_attn.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contexture_nerf_amd import _lib as L
lib = L.load(); dev = torch.device('cuda:0')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
shapes = [(2, 9216, 9216, 5), (2, 2304, 2304, 10), (2, 576, 576, 20), (2, 9216, 77, 5), (2, 8192, 8192, 12), (12, 9216, 9216, 5), (4, 9216, 9216, 5), (6, 9216, 9216, 5), (1, 9216, 9216, 5), (2, 9216, 9216, 10), (2, 4608, 9216, 5)]   # the last: 6144 waves = whole rounds at 2 and at 3 waves per SIMD
if len(sys.argv) > 2:
    shapes = [shapes[int(i)] for i in sys.argv[2].split(',')]
for (B, S, Skv, heads) in shapes:
    C = heads * 64
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(B, S, 3 * C, generator=g, device=dev).half()
    kv = torch.randn(B, Skv, 3 * C, generator=g, device=dev).half() if Skv != S else qkv
    o = torch.empty(B, S, C, dtype=torch.float16, device=dev)
    ws = torch.empty(lib.ctx_attention_ws_bytes(B, Skv, heads), dtype=torch.uint8, device=dev)
    k = kv[:, :, C:]; v = kv[:, :, 2 * C:]
    call = lambda: L.check(lib.ctx_attention_f16(C_ptr(qkv), C_ptr(k), C_ptr(v), B, S, Skv, heads, 3 * C, 3 * C, 0.125, L.ptr(o), C, L.ptr(ws), L.stream()))
    import ctypes
    C_ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 4.0 * B * heads * S * Skv * 64
    print(f"B{B} Sq{S} Skv{Skv} heads{heads}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s")
