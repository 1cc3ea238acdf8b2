import os, sys
sys.path.insert(0, '/root/repo')
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ws = torch.empty(lib.ctx_groupnorm_ws_bytes(B, 32), dtype=torch.uint8, device=dev)
tot = 0
shapes = [(9216, 320, 19), (9216, 640, 2), (9216, 960, 1), (2304, 640, 17), (2304, 1280, 1), (2304, 1920, 1), (576, 1280, 17), (576, 2560, 2)]
if len(sys.argv) > 2 and sys.argv[2] == 'vae':      # the VAE decoder's GroupNorms at 768^2 (batch 1)
    shapes = [(9216, 512, 10), (36864, 512, 7), (147456, 512, 1), (147456, 256, 6), (589824, 256, 1), (589824, 128, 6)]
for HW, C, cnt in shapes:
    x = torch.randn(B, HW, C, generator=g, device=dev).half(); y = torch.empty_like(x)
    ga = torch.ones(C, device=dev).half(); be = torch.zeros(C, device=dev).half()
    t = timeit(lambda: lib.ctx_groupnorm_f16(L.ptr(x), L.ptr(ga), L.ptr(be), B, HW, C, 32, 1e-5, int(os.environ.get("GN_SILU", "1")), L.ptr(y), L.ptr(ws), L.stream()))
    tc = timeit(lambda: y.copy_(x))
    mb = x.numel() * 2 / 1e6
    print(f"groupnorm B={B} HW={HW} C={C}: {t:7.1f} us ({3 * mb / t * 1e-3:.2f} TB/s)   torch copy {tc:6.1f} us ({2 * mb / tc * 1e-3:.2f} TB/s)  x{cnt}")
    tot += t * cnt
print("total", round(tot), "us per evaluation")
