import os, sys
sys.path.insert(0, '/root/repo')
import torch
from contexture_nerf_amd import _lib as L
lib = L.load(); dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
part = torch.empty(384 << 20, dtype=torch.uint8, device=dev)
for (H, N, Cin) in [(96, 320, 320), (96, 320, 640), (96, 320, 960), (48, 640, 640), (48, 640, 1280), (48, 640, 1920)]:
    M, K = 2 * H * H, 9 * Cin
    x = torch.randn(2, H, H, Cin, generator=g, device=dev).half()
    w = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).half()
    y = torch.empty(M, N, dtype=torch.float16, device=dev); res = torch.randn(M, N, generator=g, device=dev).half(); bias = torch.randn(N, generator=g, device=dev).half()
    out = []
    for u8 in (2, 3):
        for S in (1, 2, 3, 5):
            if S > Cin // 64: continue
            lib.ctx_gemm_tune(-1, u8)
            ms = min(lib.ctx_bench_gemm(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(res), M, N, K, L.ptr(y), 2, H, H, Cin, 0, 0, L.ptr(part), S, 10, L.stream()) for _ in range(2))
            out.append(f"u8={u8} S={S}: {ms*1e3:6.1f} us {2.0*M*N*K/ms/1e9:6.0f} TF")
    lib.ctx_gemm_tune(-1, -1)
    ms = lib.ctx_bench_gemm(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(res), M, N, K, L.ptr(y), 2, H, H, Cin, 0, 0, L.ptr(part), -1, 10, L.stream())
    print(f"conv {H}^2 {Cin}->{N}: plan {ms*1e3:6.1f} us | " + " | ".join(out))
