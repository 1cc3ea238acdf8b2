#!/usr/bin/env python3
"""The 1e-3 question on the device: SD2-depth UNet (866 M parameters, seeded random init), CFG batch 2, latent
96 x 96, 77 context tokens — relative L2 error against the fp32 oracle (host cores) of (a) the default engine (fp16 residual
stream) and (b) the fp32-residual-stream variant (ctx_unet_set_residual_fp32), and what (b) costs per step.
Usage: python tools/bench_precision.py [latent]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contexture_nerf_amd.unet import UNet2DConditionModel
from oracle import unet_ref          # tools/ may use the checker: this script is a measurement, not the product path

S = int(sys.argv[1]) if len(sys.argv) > 1 else 96
dev = torch.device("cuda:0")
torch.set_num_threads(min(os.cpu_count() or 1, 16))
torch.manual_seed(0)
ref = unet_ref.randomize_affine(unet_ref.UNet2DConditionModelRef(unet_ref.SD2_DEPTH)).eval()
net = UNet2DConditionModel(device=dev, init=False)
net.load_state_dict(ref.state_dict())
out = {"latent": S}
for seed, t in ((1, 501.0), (2, 981.0), (3, 21.0)):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(2, 5, S, S, generator=g); ctx = torch.randn(2, 77, 1024, generator=g)
    with torch.no_grad():
        want = ref(x, torch.tensor(t), ctx)['sample']
    rel = lambda y: float((y.float().cpu() - want).norm() / want.norm())
    net.set_residual_fp32(False); r16 = rel(net(x.to(dev), t, ctx.to(dev))['sample'])
    net.set_residual_fp32(True); r32 = rel(net(x.to(dev), t, ctx.to(dev))['sample'])
    out[f"t={int(t)}"] = {"fp16_stream": round(r16, 6), "fp32_stream": round(r32, 6)}
xd, cd = x.to(dev), ctx.to(dev)
for name, on in (("fp16_stream_ms", False), ("fp32_stream_ms", True)):
    net.set_residual_fp32(on)
    for _ in range(3): net(xd, 501.0, cd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): net(xd, 501.0, cd)
    torch.cuda.synchronize(); out[name] = round((time.perf_counter() - t0) * 100, 3)
net.set_residual_fp32(False)
print(json.dumps(out))
