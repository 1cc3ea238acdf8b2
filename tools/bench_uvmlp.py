#!/usr/bin/env python3
"""Texture field (NeRF2D) forward / training forward / backward at the reference's atlas size (1024^2 texels).
FLOP counts: SURVEY.md §8d (962 048 FLOP / texel forward); backward = dgrad chain (7 hidden 256x256 layers + output layer)
+ weight gradients (all layers): 2*(7*256*256 + 3*256) + 2*(42*256 + 6*256*256 + 298*256 + 3*256) FLOP / texel."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import importlib
rnh = importlib.import_module('contexture_nerf_amd.run_nerf_helpers')

res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = rnh.NeRF2D(D=8, W=256, input_ch=42, output_ch=3, skips=[4]).to(dev)
N = res * res
fwd_flop = 962048 * N
dgrad_flop = 2 * (7 * 256 * 256 + 3 * 256) * N
wgrad_flop = 2 * (42 * 256 + 6 * 256 * 256 + 298 * 256 + 3 * 256) * N


def timed(fn, n):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


with torch.no_grad():
    def infer():
        net._tex_cache = None            # measure the kernel, not the no-grad atlas cache
        return net.texture_map(res)
    t_inf = timed(infer, iters)
gt = torch.randn(1, 3, res, res, device=dev)
state = {}


def fwd():
    state.clear()                                   # one set of saved activations alive at a time
    state['tex'], state['raw'] = net.texture_map(res)


t_fwd = timed(fwd, iters)
t_bwd = 0.0
for it in range(iters + 1):
    fwd()
    net.zero_grad(set_to_none=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    state['tex'].backward(gt)
    e1.record()
    torch.cuda.synchronize()
    if it > 0:
        t_bwd += e0.elapsed_time(e1) / 1e3 / iters
out = {"texels": N,
       "fwd_infer_ms": round(t_inf * 1e3, 3), "fwd_infer_tflops": round(fwd_flop / t_inf / 1e12, 1),
       "fwd_train_ms": round(t_fwd * 1e3, 3), "fwd_train_tflops": round(fwd_flop / t_fwd / 1e12, 1),
       "saved_GB": round(N * (48 + 8 * 256) * 4 / 1e9, 2),
       "bwd_ms": round(t_bwd * 1e3, 3), "bwd_tflops": round((dgrad_flop + wgrad_flop) / t_bwd / 1e12, 1),
       "bwd_flop": dgrad_flop + wgrad_flop, "peak_f32_mfma_tflops": 157.3}
out["bwd_frac_of_peak"] = round(out["bwd_tflops"] / 157.3, 3)
print(json.dumps(out))
