#!/usr/bin/env python3
"""Roofline numbers for the HBM-bound kernels of the path at BASELINE sizes (SURVEY §8d byte formulas) and for the
exact-f32 MFMA texture field.  One JSON object per line.  Usage: python tools/bench_geometry.py [--cpu 1]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from contexture_nerf_amd import kal, view_weights as vw, run_nerf_helpers as rnh, _lib as L
from contexture_nerf_amd.render import Renderer

dev = torch.device('cuda:0')
lib = L.load()
HBM = 8000.0   # GB/s spec (6290 measured copy)


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def report(name, sec, bytes_=None, flops=None, peak=None, extra=None):
    o = {"kernel": name, "ms": round(sec * 1e3, 4)}
    if bytes_ is not None:
        o.update({"algorithmic_MB": round(bytes_ / 1e6, 2), "GB_per_s": round(bytes_ / sec / 1e9, 1), "frac_hbm_8TBs": round(bytes_ / sec / 1e9 / HBM, 4)})
    if flops is not None:
        o.update({"TFLOP": round(flops / 1e12, 4), "TFLOP_per_s": round(flops / sec / 1e12, 2), "frac_peak": round(flops / sec / 1e12 / peak, 4), "peak_TFLOPs": peak})
    if extra:
        o.update(extra)
    print(json.dumps(o), flush=True)


m = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'shapes', 'meshes.npz'))
from contexture_nerf_amd.mesh import Mesh
mesh = Mesh('shapes/nascar.obj', dev).normalize_mesh(inplace=True, target_scale=0.6, dy=0.25)
B, H, W = 7, 1200, 1200
F_, V = mesh.faces.shape[0], mesh.vertices.shape[0]
theta = torch.tensor([1.0471976] * 4 + [1.9198622] * 3, device=dev)
phi = torch.deg2rad(torch.tensor([0., 30, 150, 270, 90, 210, 330], device=dev))
r = torch.full((7,), 1.5, device=dev)
ren = Renderer(dev, dim=(H, W), interpolation_mode='bilinear')
cam = ren.get_camera_from_multiple_view(theta, phi, r, 0.25)
verts = mesh.vertices[None].repeat(B, 1, 1)
uva = torch.rand(1, F_, 3, 2, device=dev)
fvc, fvi, fn = kal.render.mesh.prepare_vertices(verts, mesh.faces, ren.camera_projection, camera_transform=cam)
report("prepare_vertices B=7 nascar", timeit(lambda: kal.render.mesh.prepare_vertices(verts, mesh.faces, ren.camera_projection, camera_transform=cam)),
       bytes_=B * (V * 12 + F_ * (36 + 24 + 12)))
t = timeit(lambda: kal.render.mesh.rasterize_fused(H, W, fvc, fvi, uva, None), 5)
report("raster fused (z,uv,idx) B=7 @1200^2 nascar", t, bytes_=B * (F_ * 72 + H * W * 20), extra={"per_view_ms": round(t / B * 1e3, 4)})
t = timeit(lambda: kal.render.mesh.rasterize_fused(H, W, fvc, fvi, uva, fn), 5)
report("raster fused + normals B=7 @1200^2", t, bytes_=B * (F_ * 84 + H * W * 32))
t1 = timeit(lambda: kal.render.mesh.rasterize_fused(H, W, fvc[:1], fvi[:1], uva, None), 10)
report("raster fused B=1 @1200^2 nascar", t1, bytes_=(F_ * 72 + H * W * 20))
depth, uv, idx, nrm = kal.render.mesh.rasterize_fused(H, W, fvc, fvi, uva, fn)
report("normalize_multiple_depth B=7 @1200^2", timeit(lambda: ren.normalize_multiple_depth(depth)), bytes_=B * H * W * 12,
       extra={"note": "includes the status read-back sync the reference's asserts also pay"})
T = 1024
tex = torch.rand(1, 3, T, T, device=dev)
report("texture_mapping fwd B=7 @1200^2 T=1024", timeit(lambda: kal.render.mesh.texture_mapping(uv, tex.expand(B, -1, -1, -1), mask_idx=idx)),
       bytes_=B * H * W * (8 + 8 + 12) + 3 * T * T * 4)
texg = tex.clone().requires_grad_(True)
go = torch.rand(B, H, W, 3, device=dev)
uvc = uv.contiguous(); g = torch.zeros(3, T, T, device=dev)
report("texture_mapping bwd (atlas scatter) B=7 @1200^2", timeit(lambda: L.check(lib.ctx_texture_mapping_bwd(L.ptr(go), L.ptr(uvc), B, H * W, 3, T, L.ptr(idx), L.ptr(g), L.stream()))),
       bytes_=B * H * W * (8 + 12 + 8) + 2 * 3 * T * T * 4)
# the binned, atomics-free scatter (uvscatter.hip): plan built once per raster, then the per-backward scatter alone
plan = torch.empty(lib.ctx_texmap_bwd_plan_bytes(B, H * W, T), dtype=torch.uint8, device=dev)
wsb = torch.empty(lib.ctx_texture_mapping_bwd_binned_ws_bytes(3, T), dtype=torch.uint8, device=dev)
report("UV scatter plan (bin pixels by atlas tile; once per raster) B=7 @1200^2",
       timeit(lambda: L.check(lib.ctx_texmap_bwd_plan(L.ptr(uvc), L.ptr(idx), B, H * W, T, L.ptr(plan), L.stream()))), bytes_=B * H * W * (8 + 8) * 2)
report("texture_mapping bwd binned (LDS tiles, no global float atomics; cached plan) B=7 @1200^2",
       timeit(lambda: L.check(lib.ctx_texture_mapping_bwd_binned(L.ptr(go), L.ptr(uvc), L.ptr(idx), B, H * W, 3, T, L.ptr(plan), L.ptr(wsb), L.ptr(g), L.stream()))),
       bytes_=B * H * W * (8 + 12 + 8) + 2 * 3 * T * T * 4)
# the painted-view back-projection proper (ConTEXTure.project_back_scatter: rgb * w and w, C = 4, int64 2^-32 sums, cached plan)
go4 = torch.rand(B, H, W, 4, device=dev)
acc = torch.zeros(4, T, T, dtype=torch.int64, device=dev)
report("UV back-projection scatter, fixed point (C=4: rgb*w + w; int64 sums; cached plan) B=7 @1200^2",
       timeit(lambda: L.check(lib.ctx_uv_scatter_fixed(L.ptr(go4), L.ptr(uvc), L.ptr(idx), B, H * W, 4, T, L.ptr(plan), 32, L.ptr(acc), L.stream()))),
       bytes_=B * H * W * (8 + 16 + 8) + 2 * 4 * T * T * 8)
report("UV back-projection scatter, fixed point, no plan (one int64 atomic per tap) B=7 @1200^2",
       timeit(lambda: L.check(lib.ctx_uv_scatter_fixed(L.ptr(go4), L.ptr(uvc), L.ptr(idx), B, H * W, 4, T, None, 32, L.ptr(acc), L.stream())), 3),
       bytes_=B * H * W * (8 + 16 + 8) + 2 * 4 * T * T * 8)
fnp = fn.permute(0, 2, 1).contiguous()
idx6 = idx[1:, None].contiguous(); fn6 = fnp[1:].contiguous()
report("view weights (scatter_max seam) B=6 @1200^2", timeit(lambda: vw.view_weight_masks(idx6, fn6)), bytes_=6 * H * W * (8 + 8 + 1) + 6 * F_ * 4)
report("create_face_view_map B=6 @1200^2", timeit(lambda: L.check(0) or vw.create_face_view_map(idx6), 3),
       bytes_=6 * H * W * 8 * 2 + int((idx6 >= 0).sum()) * 32, extra={"note": "includes the n_rows read-back"})
torch.manual_seed(0)
net = rnh.NeRF2D(D=8, W=256, input_ch=42, output_ch=3, skips=[4]).to(dev)
net.packed()
report("texture field (embed+NeRF2D+tanh) 1024^2 atlas, split-fp16 MFMA (3 passes; FLOP = the network's, not the passes')", timeit(lambda: (setattr(net, "_tex_cache", None), net.texture_map(1024))[1], 5), flops=962048.0 * 1024 * 1024, peak=2500.0 / 3)
os.environ["CTX_UVMLP_EXACT_F32"] = "1"
report("texture field (embed+NeRF2D+tanh) 1024^2 atlas, exact-f32 MFMA", timeit(lambda: (setattr(net, "_tex_cache", None), net.texture_map(1024))[1], 5), flops=962048.0 * 1024 * 1024, peak=157.3)
os.environ.pop("CTX_UVMLP_EXACT_F32")
R, S = 512 * 512, 128
raw = torch.randn(R, S, 4, device=dev); z = torch.sort(torch.rand(R, S, device=dev) * 4 + 2, -1).values; d = torch.randn(R, 3, device=dev)
rr, zz, dd = raw.contiguous(), z.contiguous(), d.contiguous()
outs = [torch.empty(R, 3, device=dev), torch.empty(R, device=dev), torch.empty(R, device=dev), torch.empty(R, device=dev)]
report("ray composite 512^2 rays x 128 samples (no weights out)",
       timeit(lambda: L.check(lib.ctx_raymarch_composite_fwd(L.ptr(rr), L.ptr(zz), L.ptr(dd), R, S, 0, L.ptr(outs[0]), L.ptr(outs[1]), L.ptr(outs[2]), None, L.ptr(outs[3]), L.stream()))),
       bytes_=R * S * 20 + R * 12 + R * 24)
x = torch.rand(1024 * 1024, 2, device=dev)
emb = rnh.Embedder(2, 10)
report("positional encode (unfused) 1024^2 x 2 -> 42", timeit(lambda: emb.embed(x)), bytes_=x.numel() * 4 + 1024 * 1024 * 42 * 4)

if len(sys.argv) > 2 and sys.argv[1] == '--cpu' and sys.argv[2] == '1':
    from oracle import geometry as og
    Hc = 300
    o_cam, o_img = fvc[:1].cpu().numpy(), fvi[:1].cpu().numpy()
    t0 = time.perf_counter(); og.rasterize(Hc, Hc, o_cam[..., 2], o_img, o_cam[..., 2:3]); dt = time.perf_counter() - t0
    print(json.dumps({"cpu_baseline": "oracle brute-force raster (1 core, C)", "sample": f"1 view @{Hc}^2 nascar, 1 pass", "seconds": round(dt, 3),
                      "scaled_to_1200^2_two_passes_s": round(dt * 16 * 2, 2)}))
    # the other stages of the path on the host cores (the oracle's C / numpy restatements; one core each unless numpy threads), so
    # that every GPU stage above has a CPU point beside it: bounded samples, scaled to the GPU line's size by pixel / texel count
    import numpy as np
    from oracle import nerf as onerf

    def cpu(name, fn, sample, scale, gpu_line):
        t0 = time.perf_counter(); fn(); d = time.perf_counter() - t0
        print(json.dumps({"cpu_baseline": name, "sample": sample, "seconds": round(d, 3), "scaled_s": round(d * scale, 3), "scaled_to": gpu_line}))
    uv1, idx1 = uv[:1].cpu().numpy(), idx[:1].cpu().numpy()
    tex_np = tex[0].cpu().numpy()
    cpu("oracle texture_mapping fwd (C, 1 core)", lambda: og.texture_mapping(uv1, tex_np[None]), "1 view @1200^2, T=1024", 7, "B=7 @1200^2")
    go1 = go[:1].cpu().numpy()
    cpu("oracle texture_mapping bwd / UV scatter (C, 1 core)", lambda: og.texture_mapping_bwd(go1, uv1, T), "1 view @1200^2, T=1024", 7, "B=7 @1200^2")
    fnz6 = fn6[:, 2, :].cpu().numpy() if fn6.dim() == 3 else fn6.cpu().numpy()
    i6 = idx6[:, 0].cpu().numpy()
    cpu("oracle view weights (scatter_max restatement, C, 1 core)", lambda: og.view_weights(i6, fnz6), "B=6 @1200^2", 1, "B=6 @1200^2")
    d1 = depth[:1].cpu().numpy()
    cpu("oracle normalize_multiple_depth (numpy)", lambda: og.normalize_multiple_depth(d1), "1 view @1200^2", 7, "B=7 @1200^2")
    ws_ = [l.weight.detach().cpu().numpy() for l in net.pts_linears]; bs_ = [l.bias.detach().cpu().numpy() for l in net.pts_linears]
    e256 = onerf.embed(onerf.uv_grid(256))
    cpu("oracle texture field forward (numpy fp32, all host threads)", lambda: onerf.nerf2d_forward(e256, ws_, bs_, net.output_linear.weight.detach().cpu().numpy(),
                                                                                                   net.output_linear.bias.detach().cpu().numpy()),
        "256^2 texels", 16, "1024^2 atlas")
    Rc = 16384
    cpu("oracle raw2outputs (C, 1 core)", lambda: og.raw2outputs(rr[:Rc].cpu().numpy(), zz[:Rc].cpu().numpy(), dd[:Rc].cpu().numpy()), f"{Rc} rays x 128 samples",
        R / Rc, "512^2 rays x 128 samples")
