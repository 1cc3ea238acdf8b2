#!/usr/bin/env python3
"""Exhaustive plan search for the UNet's GEMM / conv layer list on the GPU it runs on: for every distinct shape
(tools/bench_gemm.py's list, built for one or more latent sizes) time every tile of gemm.hip, the 256x256 kernel of
gemm8.hip and a ladder of split-K factors; write the winners as contexture-nerf_amd/csrc/gemm_tuned.h (stdout with
--emit).  Device-timed back-to-back launches, bias + residual epilogues on, random operands.

  python tools/tune_gemm.py --latents 96,64 --emit gpurun_out/gemm_tuned.h"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contexture_nerf_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--latents", default="96")
ap.add_argument("--emit", default="")
ap.add_argument("--iters", type=int, default=8)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--batch", type=int, default=2, help="UNet batch (2 = CFG pair; 1 = one CFG half per stream)")
ap.add_argument("--forms", default="", help="comma list of use8 values: time ONLY these kernels (A/B of kernel variants), print all")
ap.add_argument("--convs", type=int, default=0, help="1: convolution shapes only")
args = ap.parse_args()
lib = L.load(); dev = torch.device('cuda:0')


def layer_list(Lt, Bn=2):
    """(conv, M, N, K, flags, epi, desc) of the SD2-depth UNet at latent Lt (int: square, or (H, W)), batch Bn."""
    out = set()
    ch = [320, 640, 1280, 1280]
    Hh, Ww = (Lt, Lt) if isinstance(Lt, int) else Lt
    res = [(Hh >> k, Ww >> k) for k in range(4)]
    def conv(r, cout, cin, flags=0):
        out.add((1, (Bn, r[0], r[1]), cout, cin, flags, 0))
    def lin(M, N, K, epi=0):
        out.add((0, M, N, K, 0, epi))
    # resnets: (level, cin, cout)
    for lv, cins in enumerate([[320, 320, 960, 640, 640], [320, 640, 1920, 1280, 960], [640, 1280, 2560, 2560, 1920], [1280, 1280, 2560, 2560, 2560]]):
        c = ch[lv]; r = res[lv]
        for cin in set(cins):
            conv(r, c, cin)
            if cin != c: lin(Bn * r[0] * r[1], c, cin)
        conv(r, c, c)
    for lv in range(3):
        conv(res[lv], ch[lv], ch[lv], 1)                      # downsample (stride 2), input at res[lv]
        conv(res[lv + 1], ch[lv + 1] if lv < 2 else 1280, ch[lv + 1] if lv < 2 else 1280, 2)   # upsample conv, input at res[lv+1]
    for lv in range(4):
        c = ch[lv]; M = Bn * res[lv][0] * res[lv][1]
        lin(M, c, c); lin(M, 3 * c, c); lin(M, 8 * c, c, 1); lin(M, c, 4 * c)
    lin(77 * Bn, 24960, 1024)
    return sorted(out, key=str)


shapes = []
for Lt in [(tuple(int(v) for v in x.split("x")) if "x" in x else int(x)) for x in args.latents.split(",")]:
    for s_ in layer_list(Lt, args.batch):
        if s_ not in shapes: shapes.append(s_)
g = torch.Generator(device=dev).manual_seed(0)
part = torch.empty(384 << 20, dtype=torch.uint8, device=dev)
rows = []
for s_ in shapes:
    if s_[0] == 1:
        _, (B, H, W), N, Cin, flags, epi = s_
        st, up = (2 if flags & 1 else 1), (1 if flags & 2 else 0)
        Ho, Wo = ((H << up) - 1) // st + 1, ((W << up) - 1) // st + 1
        M, K = B * Ho * Wo, 9 * Cin
        x = torch.randn(B, H, W, Cin, generator=g, device=dev).half(); cb = (B, H, W, Cin, flags)
    else:
        _, M, N, K, flags, epi = s_
        x = torch.randn(M, K, generator=g, device=dev).half(); cb = (0, 0, 0, 0, 0)
    w = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).half()
    No = N // 2 if epi else N
    y = torch.empty(M, No, dtype=torch.float16, device=dev)
    res_ = torch.randn(M, No, generator=g, device=dev).half() if not epi else None
    bias = torch.randn(N, generator=g, device=dev).half()

    def run(tile, use8, S):
        lib.ctx_gemm_tune(tile, use8)
        ts = []
        for _ in range(args.rounds):
            ms = lib.ctx_bench_gemm(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(res_), M, N, K, L.ptr(y), *cb, epi, L.ptr(part), S, args.iters, L.stream())
            if ms <= 0: return None
            ts.append(ms)
        return min(ts)
    splits = [1]
    if epi == 0 and N % 4 == 0:
        splits += [s for s in (2, 3, 4, 6, 8, 12, 16, 24, 32) if K // 32 // s >= 4 and s * M * N * 4 <= part.numel()]
    cands = []
    if args.convs and s_[0] == 0:
        continue
    if args.forms:
        best = {}
        for u8 in [int(v) for v in args.forms.split(",")]:
            for S in splits:
                if epi == 0 and K % 64 == 0 and (s_[0] == 0 or Cin % 64 == 0) and (K // 64 // S) >= 1:
                    t = run(-1, u8, S)
                    if t and (u8 not in best or t < best[u8][0]): best[u8] = (t, S)
        lib.ctx_gemm_tune(-1, -1)
        print(f"{str(s_):42s} " + "  ".join(f"[{u8}] {best[u8][0] * 1e3:7.1f} us S={best[u8][1]}" for u8 in best), flush=True)
        rows.append((s_, best))
        continue
    for S in splits:
        for tile in range(28):
            if epi == 1 and tile in (5, 6, 9, 11, 13, 15, 17, 19, 21, 23, 26, 27): continue
            if tile >= 10 and (K % 64 or (s_[0] == 1 and Cin % 64)): continue
            t = run(tile, 0, S)
            if t: cands.append((t, tile, 0, S))
        if K % 64 == 0 and (s_[0] == 0 or Cin % 64 == 0) and (K // 64 // S) >= 1:
            t = run(-1, 1, S)
            if t: cands.append((t, -1, 1, S))
        if epi == 0 and K % 64 == 0 and (s_[0] == 0 or Cin % 64 == 0) and (K // 64 // S) >= 1:
            for u8 in (4, 5, 6, 7, 8):                          # gemm144.hip: 144x160 tiles; 6 waves / 15 waves lockstep / pipelined / barrier per two stages; 8: 288x160 lockstep
                t = run(-1, u8, S)
                if t: cands.append((t, -1, u8, S))
        if s_[0] == 1 and flags == 0 and H % 16 == 0 and W % 16 == 0 and Cin % 64 == 0 and S <= Cin // 64:
            for u8 in (2, 3):                                   # halo-staged conv, 128 / 64 features per workgroup
                t = run(-1, u8, S)
                if t: cands.append((t, -1, u8, S))
    lib.ctx_gemm_tune(-1, -1)
    base = lib.ctx_bench_gemm(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(res_), M, N, K, L.ptr(y), *cb, epi, L.ptr(part), -1, args.iters, L.stream())
    cands.sort()
    t, tile, use8, S = cands[0]
    fl = 2.0 * M * N * K
    print(f"{str(s_):50s} M={M:6d} best {t * 1e3:7.1f} us ({fl / t / 1e9:6.1f} TF) tile={tile} use8={use8} S={S}   | current plan {base * 1e3:7.1f} us | runner-up {cands[1][0] * 1e3:.1f} us {cands[1][1:]}", flush=True)
    rows.append((s_[0], M, N, K, flags, epi, tile, use8, S, t, base))
lib.ctx_gemm_tune(-1, -1)
if args.forms:
    for u8 in [int(v) for v in args.forms.split(",")]:
        print(f"form {u8}: sum {sum(b[u8][0] for _, b in rows if u8 in b) * 1e3:.0f} us")
    sys.exit(0)
print(f"sum best {sum(r[9] for r in rows) * 1e3:.0f} us vs current {sum(r[10] for r in rows) * 1e3:.0f} us (one launch per distinct shape)")
if args.emit:
    with open(args.emit, "w") as f:
        f.write("// Tuned GEMM / conv plans (tools/tune_gemm.py on MI355X): exact-shape matches override the heuristics of gemm.hip.\n"
                "// flags: bit 0 stride 2, bit 1 fused x2 upsample; tile: id of gemm.hip's list (-1 heuristic); use8: gemm8.hip's 256x256 kernel.\n"
                f"// generated for latents {args.latents} (CFG batch 2)\n#pragma once\n"
                "struct TunedGemm { int conv, M, N, K, flags, epi, tile, use8, splitk; };\nstatic const TunedGemm g_tuned[] = {\n")
        for r in rows:
            f.write(f"    {{{r[0]}, {r[1]}, {r[2]}, {r[3]}, {r[4]}, {r[5]}, {r[6]}, {r[7]}, {r[8]}}},   // {r[9] * 1e3:.1f} us (was {r[10] * 1e3:.1f})\n")
        f.write("};\n")
