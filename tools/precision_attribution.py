#!/usr/bin/env python3
"""Per-block attribution of the engine's distance from the precision contract (contract = torch.autocast fp16 as the
reference runs its UNet, src/stable_diffusion_depth.py:330-514).  SD2-depth UNet, seeded random init, CFG batch 2.

For every block output (45 taps: conv_in, each ResBlock / transformer, samplers) it prints the relative L2 distance of
  engine      vs fused      the HIP engine against oracle.unet_ref.forward_fp16_storage (one rounding per fused op: same rounding points)
  perm        vs fused      that restatement against ITSELF with every matmul / conv summing K in another order (identical real-number
                            result, fp32-order noise only): how far two correct implementations of one contract end up
  engine      vs fp32, fused vs fp32
and, for the final output, the other contract variants (P in fp16, fp16 time-embedding tensors, literal op-level autocast).
If `engine vs fused` tracks `perm vs fused`, the engine parts from the restatement by accumulation order alone; a tap where it
jumps ABOVE that curve names a rounding point that differs.
Usage: python tools/precision_attribution.py [latent=96] [out.json]"""
import ctypes as C
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contexture_nerf_amd import _lib as L
from contexture_nerf_amd.unet import UNet2DConditionModel
from oracle import unet_ref as U          # tools/ may use the checker: this script is a measurement, not the product path

S = int(sys.argv[1]) if len(sys.argv) > 1 else 96
out_path = sys.argv[2] if len(sys.argv) > 2 else None
dev = torch.device("cuda:0")
torch.set_num_threads(min(os.cpu_count() or 1, 16))
torch.manual_seed(0)
ref = U.randomize_affine(U.UNet2DConditionModelRef(U.SD2_DEPTH)).eval()
net = UNet2DConditionModel(device=dev, init=False)
net.load_state_dict(ref.state_dict())
g = torch.Generator().manual_seed(1)
x = torch.randn(2, 5, S, S, generator=g); ctx = torch.randn(2, 77, 1024, generator=g); t = 501.0
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())

lib = L.load()
cap = 2 * 64 * S * S * 320 + (1 << 20)                    # generous: 45 taps, the widest is [2*S*S, 320]
buf = torch.zeros(cap, dtype=torch.float16, device=dev)
L.check(lib.ctx_unet_set_taps(net._h, L.ptr(buf), cap))
t0 = time.time()
y_eng = net(x.to(dev), t, ctx.to(dev))['sample'].float().cpu()
torch.cuda.synchronize()
n = lib.ctx_unet_tap_count(net._h)
eng = []
for i in range(n):
    off, rows, ch = C.c_int64(), C.c_int32(), C.c_int32()
    L.check(lib.ctx_unet_tap_info(net._h, i, C.byref(off), C.byref(rows), C.byref(ch)))
    assert off.value + rows.value * ch.value <= cap
    hw = rows.value // 2
    side = int(round(hw ** 0.5))
    eng.append(buf[off.value:off.value + rows.value * ch.value].view(2, side, side, ch.value).permute(0, 3, 1, 2).float().cpu())
L.check(lib.ctx_unet_set_taps(net._h, None, 0))
print(f"engine forward + {n} taps: {time.time() - t0:.1f} s", flush=True)

tt = torch.tensor(t)
t0 = time.time(); o32, t32 = U.forward_taps(ref, x, tt, ctx); print(f"fp32 oracle: {time.time() - t0:.1f} s", flush=True)
tF = []; t0 = time.time(); oF = U.forward_fp16_storage(ref, x, tt, ctx, taps=tF)['sample']; print(f"fused restatement: {time.time() - t0:.1f} s", flush=True)
tP = []; t0 = time.time(); oP = U.forward_fp16_storage(ref, x, tt, ctx, taps=tP, perm=U._PermLinear(4))['sample']; print(f"permuted restatement: {time.time() - t0:.1f} s", flush=True)
assert len(tF) == n == len(t32), (len(tF), n)
rows = []
print(f"{'tap':>3} {'shape':>18}  {'engine-fused':>12} {'perm-fused':>12} {'engine-fp32':>12} {'fused-fp32':>12}")
for i in range(n):
    r = dict(tap=i, shape=list(tF[i].shape), engine_vs_fused=rel(eng[i], tF[i]), perm_vs_fused=rel(tP[i], tF[i]),
             engine_vs_fp32=rel(eng[i], t32[i]), fused_vs_fp32=rel(tF[i], t32[i]))
    rows.append(r)
    print(f"{i:>3} {str(tuple(tF[i].shape)):>18}  {r['engine_vs_fused']:12.3e} {r['perm_vs_fused']:12.3e} {r['engine_vs_fp32']:12.3e} {r['fused_vs_fp32']:12.3e}", flush=True)
final = dict(engine_vs_fused=rel(y_eng, oF), perm_vs_fused=rel(oP, oF), engine_vs_perm=rel(y_eng, oP), engine_vs_fp32=rel(y_eng, o32['sample']),
             fused_vs_fp32=rel(oF, o32['sample']))
for name, kw in (("p16", dict(p16=True)), ("temb16", dict(temb16=True)), ("p16_temb16", dict(p16=True, temb16=True)), ("autocast", dict(autocast=True))):
    o = U.forward_fp16_storage(ref, x, tt, ctx, **kw)['sample']
    final[f"engine_vs_{name}"] = rel(y_eng, o)
    final[f"{name}_vs_fused"] = rel(o, oF)
    final[f"{name}_vs_fp32"] = rel(o, o32['sample'])
ident = lambda v: v
final["fp16_weights_only_vs_fp32"] = rel(U.forward_fp16_storage(ref, x, tt, ctx, q=ident, q_res=ident)['sample'], o32['sample'])
print(json.dumps(final, indent=1))
res = dict(latent=S, timestep=t, taps=rows, final=final,
           worst_ratio_engine_over_perm=max(r['engine_vs_fused'] / max(r['perm_vs_fused'], 1e-12) for r in rows[3:]))
print("worst engine/perm ratio past the first blocks:", round(res['worst_ratio_engine_over_perm'], 3))
if out_path:
    json.dump(res, open(out_path, 'w'), indent=1)
