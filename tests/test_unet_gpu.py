"""GPU parity of the UNet-denoise kernels (through the C-ABI) against plain PyTorch fp32 references of
the same ops, and of the whole engine against the oracle's fp32 UNet (oracle/unet_ref.py).

Tolerances (stated per test): operands/outputs are fp16 with fp32 accumulation, so a single op is
compared to the fp32 result of the SAME fp16-rounded inputs within 2^-10 relative (+ small abs);
the end-to-end output is held (a) against pure fp32 to within 1.25x of what fp16 storage itself costs — measured by
the oracle's fp16-storage restatement of the reference's autocast contract — and (b) to <= 1.6e-3 relative L2 against
that restatement.  north_star's "within 1e-3 rel fp16" is below the noise floor of fp16-operand arithmetic on this
network: fp16 WEIGHTS alone put any fp16 implementation 0.84e-3 from fp32 (tests/test_precision_cpu.py)."""
import ctypes as C
import os
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    from contexture_nerf_amd import _lib as L
    return L, L.load()


def _close(got, want, rtol=2e-3, atol=2e-3, what=""):
    got, want = got.float().cpu(), want.float().cpu()
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = (err > tol).sum().item()
    assert bad == 0, f"{what}: {bad}/{got.numel()} off; max err {err.max():.4e}, rel L2 {(got - want).norm() / want.norm():.3e}"


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 320, 320), (257, 64, 128), (18, 1280, 1024), (1000, 2560, 320),
                                   (4096, 640, 1920)])
def test_gemm(dev, M, N, K):
    L, lib = _lib()
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g)).half()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half()
    bias = torch.randn(N, generator=g).half()
    res = torch.randn(M, N, generator=g).half()
    Ad, Wd, bd, rd = A.to(dev), W.to(dev), bias.to(dev), res.to(dev)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    L.check(lib.ctx_gemm_f16(L.ptr(Ad), L.ptr(Wd), L.ptr(bd), L.ptr(rd), M, N, K, L.ptr(out), L.stream()))
    want = A.float() @ W.float().T + bias.float() + res.float()
    _close(out, want, what=f"gemm {M}x{N}x{K}")
    out2 = torch.empty(M, N, dtype=torch.float16, device=dev)
    L.check(lib.ctx_gemm_f16(L.ptr(Ad), L.ptr(Wd), None, None, M, N, K, L.ptr(out2), L.stream()))
    _close(out2, A.float() @ W.float().T, what="gemm no-epilogue")


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,ups", [(2, 16, 16, 64, 64, 1, 0), (1, 9, 13, 128, 320, 1, 0),
                                                        (2, 16, 12, 64, 128, 2, 0), (2, 8, 8, 128, 64, 1, 1),
                                                        (2, 32, 32, 320, 320, 1, 0), (1, 24, 24, 1920, 640, 1, 0)])
def test_conv3x3(dev, B, H, W, Cin, Cout, stride, ups):
    L, lib = _lib()
    g = torch.Generator().manual_seed(B * H + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g).half()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).half()
    bias = torch.randn(Cout, generator=g).half()
    rowb = torch.randn(B, Cout, generator=g).half()
    xin = F.interpolate(x.float(), scale_factor=2.0, mode='nearest') if ups else x.float()
    want = F.conv2d(xin, w.float(), bias.float(), stride=stride, padding=1) + rowb.float()[:, :, None, None]
    res = torch.randn(want.shape, generator=g).half()
    want = want + res.float()
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    bd, rbd = bias.to(dev), rowb.to(dev)
    rd = res.permute(0, 2, 3, 1).contiguous().to(dev)
    Ho, Wo = want.shape[2], want.shape[3]
    y = torch.empty(B, Ho, Wo, Cout, dtype=torch.float16, device=dev)
    L.check(lib.ctx_conv3x3_f16(L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(rbd), L.ptr(rd), B, H, W, Cin, Cout, stride, ups, L.ptr(y), L.stream()))
    _close(y.permute(0, 3, 1, 2), want, rtol=3e-3, atol=4e-3, what="conv3x3")


@pytest.fixture
def force_gemm8():
    """Route every applicable GEMM / conv through the 256x256 8-wave kernel (gemm8.hip), whatever its size."""
    import os
    old = os.environ.get("CTX_GEMM8")
    os.environ["CTX_GEMM8"] = "2"
    yield
    if old is None:
        os.environ.pop("CTX_GEMM8", None)
    else:
        os.environ["CTX_GEMM8"] = old


@pytest.mark.parametrize("M,N,K", [(512, 512, 256), (300, 320, 320), (257, 72, 64), (1000, 2560, 128), (2048, 256, 1344)])
def test_gemm8_forced(dev, force_gemm8, M, N, K):
    test_gemm(dev, M, N, K)


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,ups", [(2, 16, 16, 64, 64, 1, 0), (1, 9, 13, 128, 320, 1, 0),
                                                        (2, 16, 12, 64, 128, 2, 0), (2, 8, 8, 128, 64, 1, 1),
                                                        (2, 32, 32, 320, 320, 1, 0)])
def test_conv8_forced(dev, force_gemm8, B, H, W, Cin, Cout, stride, ups):
    test_conv3x3(dev, B, H, W, Cin, Cout, stride, ups)


@pytest.mark.parametrize("tile", list(range(28)))
def test_every_tile_form(dev, tile):
    """Each tile form of gemm.hip forced through the tuning override: GEMM with ragged M / N and a convolution."""
    L, lib = _lib()
    lib.ctx_gemm_tune(tile, 0)
    try:
        test_gemm(dev, 300, 320, 320)
        test_gemm(dev, 1000, 200, 128)
        test_conv3x3(dev, 2, 16, 12, 64, 128, 1, 0)
        test_conv3x3(dev, 1, 9, 13, 128, 320, 2, 0)
    finally:
        lib.ctx_gemm_tune(-1, -1)


@pytest.mark.parametrize("ni", [2, 3])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 16, 16, 64, 64), (1, 32, 16, 128, 320), (2, 48, 48, 192, 136), (1, 16, 32, 320, 72)])
def test_conv_halo_forced(dev, ni, B, H, W, Cin, Cout):
    """Halo-staged conv kernel (conv_halo.hip) forced through the tuning override, incl. ragged feature tiles and image borders."""
    L, lib = _lib()
    lib.ctx_gemm_tune(-1, ni)
    try:
        test_conv3x3(dev, B, H, W, Cin, Cout, 1, 0)
    finally:
        lib.ctx_gemm_tune(-1, -1)


@pytest.mark.parametrize("form", [4, 5, 6, 7, 8])
def test_gemm144_forced(dev, form):
    """144x160 kernel, 6- and 15-wave forms (gemm144.hip) forced through the tuning override: whole and ragged tiles, bias + residual, the
    conv address generator (stride 1 / 2, fused x2 upsample, image borders) and fp32 split-K slabs."""
    L, lib = _lib()
    lib.ctx_gemm_tune(-1, form)
    try:
        for M, N, K in [(288, 320, 320), (300, 320, 64), (1000, 200, 128), (144, 160, 1344), (1152, 1280, 640), (77, 24, 192)]:
            test_gemm(dev, M, N, K)
        for B, H, W, Cin, Cout, stride, ups in [(2, 12, 12, 64, 160, 1, 0), (1, 9, 13, 128, 320, 1, 0), (2, 16, 12, 64, 128, 2, 0),
                                                 (2, 8, 8, 128, 64, 1, 1), (2, 24, 24, 320, 320, 1, 0)]:
            test_conv3x3(dev, B, H, W, Cin, Cout, stride, ups)
        g = torch.Generator().manual_seed(5)
        for M, N, K, splitk in [(512, 320, 448, 3), (200, 320, 1024, 5), (288, 160, 2048, 16)]:
            A = torch.randn(M, K, generator=g).half()
            W = (torch.randn(N, K, generator=g) / K ** 0.5).half(); bias = torch.randn(N, generator=g).half()
            res = torch.randn(M, N, generator=g).half()
            out = torch.zeros(M, N, dtype=torch.float16, device=dev)
            part = torch.empty(splitk * M * N, dtype=torch.float32, device=dev)
            Ad, Wd, bd, rd = A.to(dev), W.to(dev), bias.to(dev), res.to(dev)
            ms = lib.ctx_bench_gemm(L.ptr(Ad), L.ptr(Wd), L.ptr(bd), L.ptr(rd), M, N, K, L.ptr(out), 0, 0, 0, 0, 0, 0, L.ptr(part), splitk, 1, L.stream())
            assert ms > 0, lib.ctx_last_error()
            _close(out, A.float() @ W.float().T + bias.float() + res.float(), what=f"gemm144 split-K {splitk}")
    finally:
        lib.ctx_gemm_tune(-1, -1)


def test_gemm_kernels_repeat_without_races(dev):
    """Race screen (short form of tools/stress_gemm.py): the kernels whose K loops keep DMA in flight across barriers, several
    workgroups per CU, five launches each on one UNet-sized problem; an early LDS read or refill shows as a few hundred wrong
    elements in some launches only."""
    L, lib = _lib()
    g = torch.Generator(device=dev).manual_seed(3)
    M, N, K = 4608, 640, 640
    x = torch.randn(M, K, generator=g, device=dev).half(); w = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).half()
    b = torch.randn(N, generator=g, device=dev).half(); r = torch.randn(M, N, generator=g, device=dev).half()
    want = x.float() @ w.float().T + b.float() + r.float()
    part = torch.empty(2 * M * N, dtype=torch.float32, device=dev)
    try:
        for tile, u8 in [(5, 0), (14, 0), (15, 0), (19, 0), (20, 0), (25, 0), (-1, 1), (-1, 5), (-1, 6), (-1, 7)]:
            for S in (1, 2):
                lib.ctx_gemm_tune(tile, u8)
                for rep in range(5):
                    y = torch.zeros(M, N, dtype=torch.float16, device=dev)
                    ms = lib.ctx_bench_gemm(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(r), M, N, K, L.ptr(y), 0, 0, 0, 0, 0, 0, L.ptr(part), S, 1, L.stream())
                    assert ms > 0, lib.ctx_last_error()
                    _close(y, want, rtol=3e-3, atol=4e-3, what=f"tile {tile} use8 {u8} split {S} launch {rep}")
    finally:
        lib.ctx_gemm_tune(-1, -1)


@pytest.mark.parametrize("forced", [0, 1])
@pytest.mark.parametrize("M,C4,K,splitk", [(384, 256, 192, 1), (300, 96, 64, 1), (512, 0, 448, 3), (200, 0, 1024, 5)])
def test_gemm_geglu_and_splitk(dev, M, C4, K, splitk, forced):
    """GEGLU epilogue (packed [32 value | 32 gate] weight rows) and fp32 split-K slabs, through the benchmark entry."""
    import os
    L, lib = _lib()
    g = torch.Generator().manual_seed(M + C4 + K)
    old = os.environ.get("CTX_GEMM8")
    os.environ["CTX_GEMM8"] = "2" if forced else "0"
    try:
        A = torch.randn(M, K, generator=g).half()
        if C4:
            Wv = (torch.randn(C4, K, generator=g) / K ** 0.5).half(); Wg = (torch.randn(C4, K, generator=g) / K ** 0.5).half()
            bv = torch.randn(C4, generator=g).half(); bg = torch.randn(C4, generator=g).half()
            Wp = torch.empty(2 * C4, K, dtype=torch.float16); bp = torch.empty(2 * C4, dtype=torch.float16)
            for blk in range(C4 // 32):
                Wp[64 * blk:64 * blk + 32] = Wv[32 * blk:32 * blk + 32]; Wp[64 * blk + 32:64 * blk + 64] = Wg[32 * blk:32 * blk + 32]
                bp[64 * blk:64 * blk + 32] = bv[32 * blk:32 * blk + 32]; bp[64 * blk + 32:64 * blk + 64] = bg[32 * blk:32 * blk + 32]
            out = torch.zeros(M, C4, dtype=torch.float16, device=dev)
            Ad, Wd, bd = A.to(dev), Wp.to(dev), bp.to(dev)
            ms = lib.ctx_bench_gemm(L.ptr(Ad), L.ptr(Wd), L.ptr(bd), None, M, 2 * C4, K, L.ptr(out), 0, 0, 0, 0, 0, 1, None, 1, 1, L.stream())
            assert ms > 0, lib.ctx_last_error()
            want = (A.float() @ Wv.float().T + bv.float()) * F.gelu(A.float() @ Wg.float().T + bg.float())
            _close(out, want, rtol=3e-3, atol=3e-3, what="gemm+GEGLU")
        else:
            N = 320
            W = (torch.randn(N, K, generator=g) / K ** 0.5).half(); bias = torch.randn(N, generator=g).half()
            res = torch.randn(M, N, generator=g).half()
            out = torch.zeros(M, N, dtype=torch.float16, device=dev)
            part = torch.empty(splitk * M * N, dtype=torch.float32, device=dev)
            Ad, Wd, bd, rd = A.to(dev), W.to(dev), bias.to(dev), res.to(dev)
            ms = lib.ctx_bench_gemm(L.ptr(Ad), L.ptr(Wd), L.ptr(bd), L.ptr(rd), M, N, K, L.ptr(out), 0, 0, 0, 0, 0, 0, L.ptr(part), splitk, 1, L.stream())
            assert ms > 0, lib.ctx_last_error()
            _close(out, A.float() @ W.float().T + bias.float() + res.float(), what=f"gemm split-K {splitk}")
    finally:
        if old is None:
            os.environ.pop("CTX_GEMM8", None)
        else:
            os.environ["CTX_GEMM8"] = old


def _tuned_entries():
    import os, re
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "contexture-nerf_amd", "csrc", "gemm_tuned.h")
    out = []
    for line in open(path):
        m = re.match(r"\s*\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (-?\d+), (-?\d+), (\d+)\}", line)
        if m:
            out.append(tuple(int(x) for x in m.groups()))
    return out


def test_tuned_plans_against_torch(dev):
    """Every plan of the tuned table (tile form, 256x256 kernel, split-K) at its exact full-size shape against an fp32 torch
    reference computed on the same GPU from the same fp16-rounded operands (bias + residual epilogue, GEGLU where planned)."""
    L, lib = _lib()
    entries = _tuned_entries()
    assert len(entries) > 50
    g = torch.Generator(device=dev).manual_seed(11)
    part = torch.empty(384 << 20, dtype=torch.uint8, device=dev)
    checked = 0
    for conv, M, N, K, flags, epi, tile, use8, S in entries:
        if conv:
            Cin = K // 9
            st, up = (2 if flags & 1 else 1), (1 if flags & 2 else 0)
            # recover the output grid from M = Bn * Ho * Wo: square with Bn = 2 (CFG pair) or 1, or the 3:2 view grid of
            # Zero123++ (120 x 80 and its halvings) with Bn = 2
            Ho = Wo = int(round((M // 2) ** 0.5)); Bn = 2
            if Bn * Ho * Wo != M:
                Ho = Wo = int(round(M ** 0.5)); Bn = 1
            if Bn * Ho * Wo != M:
                k = int(round((M / 12) ** 0.5)); Ho, Wo, Bn = 3 * k, 2 * k, 2
            assert Bn * Ho * Wo == M, (M, N, K)
            H, W_ = (Ho // 2, Wo // 2) if up else ((Ho * 2, Wo * 2) if st == 2 else (Ho, Wo))
            x = torch.randn(Bn, H, W_, Cin, generator=g, device=dev).half()
            cb = (Bn, H, W_, Cin, flags)
        else:
            x = torch.randn(M, K, generator=g, device=dev).half()
            cb = (0, 0, 0, 0, 0)
        w = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).half()
        No = N // 2 if epi else N
        y = torch.zeros(M, No, dtype=torch.float16, device=dev)
        res = torch.randn(M, No, generator=g, device=dev).half() if not epi else None
        bias = torch.randn(N, generator=g, device=dev).half()
        ms = lib.ctx_bench_gemm(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(res), M, N, K, L.ptr(y), *cb, epi, L.ptr(part), -1, 1, L.stream())
        assert ms > 0, lib.ctx_last_error()
        if conv:
            xin = x.float().permute(0, 3, 1, 2)
            if up:
                xin = F.interpolate(xin, scale_factor=2.0, mode='nearest')
            wt = w.float().view(N, 3, 3, Cin).permute(0, 3, 1, 2)
            want = F.conv2d(xin, wt, bias.float(), stride=st, padding=1).permute(0, 2, 3, 1).reshape(M, N) + res.float()
        elif epi:
            full = x.float() @ w.float().T + bias.float()                       # packed [32 value | 32 gate] per 64 rows
            full = full.view(M, N // 64, 2, 32)
            want = (full[:, :, 0] * F.gelu(full[:, :, 1])).reshape(M, N // 2)
        else:
            want = x.float() @ w.float().T + bias.float() + res.float()
        err = (y.float() - want).abs()
        tol = 4e-3 + 3e-3 * want.abs()
        bad = int((err > tol).sum())
        assert bad == 0, f"plan {(conv, M, N, K, flags, epi, tile, use8, S)}: {bad} off, max err {float(err.max()):.3e}"
        checked += 1
        del x, w, y, res, bias, want, err
    assert checked == len(entries)


@pytest.mark.parametrize("B,HW,Cc,G,silu", [(2, 256, 64, 32, 1), (2, 1024, 320, 32, 1), (1, 144, 1280, 32, 0),
                                            (2, 400, 2560, 32, 1), (2, 576, 1920, 32, 1), (2, 100, 960, 32, 0),
                                            (2, 576, 1280, 32, 1), (2, 1229, 1280, 32, 0),
                                            # r3 thread mappings (a thread keeps one 8-channel column): batch 12, pixel counts that are no
                                            # multiple of lanes x chunks, column counts that do not divide 256, 8 channels, 4 groups, C = 4096
                                            (12, 2304, 640, 32, 1), (12, 576, 1280, 32, 1), (3, 1237, 320, 32, 1), (1, 9216, 960, 32, 1),
                                            (2, 77, 8, 1, 0), (2, 513, 96, 4, 1), (1, 301, 4096, 32, 0), (5, 37, 2560, 32, 1),
                                            (2, 3001, 24, 1, 1), (1, 2304, 1920, 32, 1)])
def test_groupnorm(dev, B, HW, Cc, G, silu):
    L, lib = _lib()
    g = torch.Generator().manual_seed(Cc)
    x = (torch.randn(B, HW, Cc, generator=g) * 2 + 0.5).half()
    ga = (1 + 0.2 * torch.randn(Cc, generator=g)).half()
    be = (0.2 * torch.randn(Cc, generator=g)).half()
    want = F.group_norm(x.float().permute(0, 2, 1), G, ga.float(), be.float(), eps=1e-5)
    if silu:
        want = F.silu(want)
    want = want.permute(0, 2, 1)
    xd, gd, bd = x.to(dev), ga.to(dev), be.to(dev)
    y = torch.empty_like(xd)
    ws = torch.empty(lib.ctx_groupnorm_ws_bytes(B, G), dtype=torch.uint8, device=dev)
    L.check(lib.ctx_groupnorm_f16(L.ptr(xd), L.ptr(gd), L.ptr(bd), B, HW, Cc, G, 1e-5, silu, L.ptr(y), L.ptr(ws), L.stream()))
    _close(y, want, what="groupnorm")


def test_layernorm_geglu(dev):
    L, lib = _lib()
    g = torch.Generator().manual_seed(3)
    # (rows, C): both kernel families — 8 lanes per row (C % 64 == 0, C <= 640, rows >= 64; ragged last group of 8 rows and of 32) and one wave per row
    for rows, Cc in [(77, 320), (513, 1280), (9, 64), (18432, 320), (4609, 640), (64, 64), (1001, 448), (129, 576), (300, 1920), (63, 320), (200, 200)]:
        x = (torch.randn(rows, Cc, generator=g) * 3 - 1).half()
        ga = (1 + 0.2 * torch.randn(Cc, generator=g)).half(); be = (0.2 * torch.randn(Cc, generator=g)).half()
        xd, gd, bd = x.to(dev), ga.to(dev), be.to(dev)
        y = torch.empty_like(xd)
        L.check(lib.ctx_layernorm_f16(L.ptr(xd), L.ptr(gd), L.ptr(bd), rows, Cc, 1e-5, L.ptr(y), L.stream()))
        _close(y, F.layer_norm(x.float(), (Cc,), ga.float(), be.float(), 1e-5), what="layernorm")
    h = torch.randn(100, 2 * 256, generator=g).half()
    hd = h.to(dev)
    y = torch.empty(100, 256, dtype=torch.float16, device=dev)
    L.check(lib.ctx_geglu_f16(L.ptr(hd), 100, 256, L.ptr(y), L.stream()))
    a, gt = h.float().chunk(2, -1)
    _close(y, a * F.gelu(gt), what="geglu")


@pytest.mark.parametrize("B,Sq,Skv,heads", [(2, 256, 256, 1), (1, 200, 200, 5), (2, 1024, 77, 2), (2, 96, 7, 1), (1, 2304, 2304, 5)])
def test_attention(dev, B, Sq, Skv, heads):
    L, lib = _lib()
    g = torch.Generator().manual_seed(Sq + Skv)
    Cc = heads * 64
    q = torch.randn(B, Sq, Cc, generator=g).half()
    k = torch.randn(B, Skv, Cc, generator=g).half()
    v = torch.randn(B, Skv, Cc, generator=g).half()
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    o = torch.empty(B, Sq, Cc, dtype=torch.float16, device=dev)
    ws = torch.empty(lib.ctx_attention_ws_bytes(B, Skv, heads), dtype=torch.uint8, device=dev)
    L.check(lib.ctx_attention_f16(L.ptr(qd), L.ptr(kd), L.ptr(vd), B, Sq, Skv, heads, Cc, Cc, 0.125, L.ptr(o), Cc, L.ptr(ws), L.stream()))
    qh = q.float().view(B, Sq, heads, 64).transpose(1, 2)
    kh = k.float().view(B, Skv, heads, 64).transpose(1, 2)
    vh = v.float().view(B, Skv, heads, 64).transpose(1, 2)
    want = (torch.softmax(qh @ kh.transpose(-1, -2) * 0.125, -1) @ vh).transpose(1, 2).reshape(B, Sq, Cc)
    # P is rounded to fp16 before the second product: 2^-11 relative per term, averaged over Skv terms
    _close(o, want, rtol=3e-3, atol=3e-3, what="attention")


def test_attention_rescale_branch(dev):
    """Online-softmax rescale forced: one key per later tile dominates the running max of specific queries."""
    L, lib = _lib()
    B, S, heads = 1, 256, 1
    g = torch.Generator().manual_seed(9)
    q = torch.randn(B, S, 64, generator=g).half()
    k = torch.randn(B, S, 64, generator=g).half()
    v = torch.randn(B, S, 64, generator=g).half()
    k[0, 70] = q[0, 5] * 4          # spikes in tile 1 and tile 3
    k[0, 200] = q[0, 37] * 6
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    o = torch.empty(B, S, 64, dtype=torch.float16, device=dev)
    ws = torch.empty(lib.ctx_attention_ws_bytes(B, S, heads), dtype=torch.uint8, device=dev)
    L.check(lib.ctx_attention_f16(L.ptr(qd), L.ptr(kd), L.ptr(vd), B, S, S, heads, 64, 64, 0.125, L.ptr(o), 64, L.ptr(ws), L.stream()))
    want = torch.softmax(q.double() @ k.double().transpose(-1, -2) * 0.125, -1) @ v.double()
    _close(o, want, rtol=3e-3, atol=3e-3, what="attention rescale")


def test_attention_creeping_maximum_lazy_rescale(dev):
    """The lazily updated maximum (attention.hip: O and l are rescaled only when some row's maximum moved by more than 2^8): scores that
    rise a little with every key keep the subtracted maximum STALE for several tiles, so P runs up to 2^8 before a rescale; rows that
    jump, rows that never move and a ragged last tile in the same workgroup.  Against float64 softmax, and against CTX_ATTN_LAZY=0
    (rescale on every change) run in a child process: the two agree to fp16 rounding of P."""
    import subprocess, sys, tempfile
    L, lib = _lib()
    B, S, Skv, heads = 1, 192, 1000, 2
    g = torch.Generator().manual_seed(4)
    q = torch.randn(B, S, heads * 64, generator=g).half()
    k = (0.05 * torch.randn(B, Skv, heads * 64, generator=g)).half()
    v = torch.randn(B, Skv, heads * 64, generator=g).half()
    ramp = torch.linspace(0.0, 6.0, Skv)                                    # log2-scores creep up by ~0.4 per 64-key tile ...
    k[0, :, :64] += (ramp[:, None] * q[0, 3:4, :64].float() / q[0, 3, :64].float().pow(2).sum() * 8 / 1.4427).half()
    k[0, 700, 64:] = q[0, 9, 64:] * 5                                       # ... and one row of the other head jumps once
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    o = torch.empty(B, S, heads * 64, dtype=torch.float16, device=dev)
    L.check(lib.ctx_attention_f16(L.ptr(qd), L.ptr(kd), L.ptr(vd), B, S, Skv, heads, heads * 64, heads * 64, 0.125, L.ptr(o), heads * 64, None, L.stream()))
    qh = q.double().view(B, S, heads, 64).transpose(1, 2); kh = k.double().view(B, Skv, heads, 64).transpose(1, 2); vh = v.double().view(B, Skv, heads, 64).transpose(1, 2)
    want = (torch.softmax(qh @ kh.transpose(-1, -2) * 0.125, -1) @ vh).transpose(1, 2).reshape(B, S, heads * 64)
    assert torch.isfinite(o.float()).all()
    _close(o, want, rtol=3e-3, atol=3e-3, what="attention, creeping maximum")
    # the textbook recurrence in a fresh process (the threshold is read once per process)
    with tempfile.TemporaryDirectory() as td:
        torch.save({"q": q, "k": k, "v": v}, td + "/in.pt")
        code = ("import torch, sys; sys.path.insert(0, %r); from contexture_nerf_amd import _lib as L; lib = L.load(); d = torch.load(%r); dev = torch.device('cuda:0');"
                "q, k, v = (d[n].to(dev) for n in 'qkv'); o = torch.empty_like(q);"
                "L.check(lib.ctx_attention_f16(L.ptr(q), L.ptr(k), L.ptr(v), 1, %d, %d, %d, %d, %d, 0.125, L.ptr(o), %d, None, L.stream())); torch.save(o.cpu(), %r)"
                % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), td + "/in.pt", S, Skv, heads, heads * 64, heads * 64, heads * 64, td + "/out.pt"))
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CTX_ATTN_LAZY="0"), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        o0 = torch.load(td + "/out.pt")
    d = (o.cpu().float() - o0.float()).abs()
    assert float(d.max()) <= 4e-3 and float((d > 0).float().mean()) > 0.0, f"lazy vs eager rescale: max {float(d.max()):.3e}"


def test_cfg_plms_scheduler(dev):
    from contexture_nerf_amd.scheduler import PNDMScheduler
    from oracle.scheduler import PNDMRef, cfg
    for n in (50, 2, 7):
        sch, ref = PNDMScheduler(), PNDMRef()
        sch.set_timesteps(n); ts = ref.set_timesteps(n)
        assert np.array_equal(sch.timesteps.numpy(), ts)
        g = torch.Generator().manual_seed(n)
        x = torch.randn(1, 4, 8, 8, generator=g)
        xr = x.numpy().copy(); xd = x.to(dev)
        for t in ts:
            pair = torch.randn(2, 4, 8, 8, generator=g)
            xd = sch.step_cfg(pair.to(dev), 10.0, int(t), xd)['prev_sample']
            xr = ref.step(cfg(pair.numpy(), 10.0), t, xr)
            d = np.linalg.norm(xd.cpu().numpy() - xr) / np.linalg.norm(xr)
            assert d < 2e-6, d      # float32 op-order differences only (|x| grows to ~1e2 at guidance 10)
    # diffusers-shaped step() == step_cfg with identical halves; add_noise
    sch, ref = PNDMScheduler(), PNDMRef()
    sch.set_timesteps(5); ref.set_timesteps(5)
    x = torch.randn(1, 4, 4, 4); e = torch.randn(1, 4, 4, 4)
    out = sch.step(e.to(dev), int(sch.timesteps[0]), x.to(dev))['prev_sample']
    np.testing.assert_allclose(out.cpu().numpy(), ref.step(e.numpy(), ref.timesteps[0], x.numpy()), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(sch.add_noise(x, e, torch.tensor([981])).numpy(), ref.add_noise(x.numpy(), e.numpy(), 981), rtol=1e-6, atol=1e-6)


def _rel(got, want):
    return ((got.float().cpu() - want).norm() / want.norm()).item()


@pytest.mark.parametrize("cfgname,h,w,L", [("tiny", 16, 16, 7), ("tiny", 24, 8, 77), ("mid", 16, 16, 77)])
def test_unet_vs_oracle(dev, cfgname, h, w, L):
    """Whole engine vs the fp32 oracle UNet on seeded random-init weights (norm affines perturbed)."""
    from contexture_nerf_amd.unet import UNet2DConditionModel
    from oracle import unet_ref
    cfg = unet_ref.tiny_config() if cfgname == "tiny" else unet_ref.tiny_config(ch=(64, 128, 256, 256), heads=(1, 2, 4, 4), ctx_dim=128)
    torch.manual_seed(1)
    ref = unet_ref.randomize_affine(unet_ref.UNet2DConditionModelRef(cfg)).eval()
    net = UNet2DConditionModel(cfg, device=dev, init=False)
    assert net.param_shapes() == {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    net.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 5, h, w, generator=g)
    ctx = torch.randn(2, L, cfg['cross_attention_dim'], generator=g)
    for t in (981.0, 1.0):
        with torch.no_grad():
            want32 = ref(x, torch.tensor(t), ctx)['sample']
        want16 = unet_ref.forward_fp16_storage(ref, x, torch.tensor(t), ctx)['sample']     # the reference's autocast contract
        got = net(x.to(dev), t, ctx.to(dev))['sample']
        assert got.shape == want32.shape and torch.isfinite(got).all()
        r16, r32, floor = _rel(got, want16), _rel(got, want32), _rel(want16, want32)
        print(f"unet {cfgname} {h}x{w} t={t}: rel L2 vs fp16-storage restatement {r16:.3e} | vs fp32 {r32:.3e} | "
              f"restatement vs fp32 {floor:.3e}")
        # Two fp16-storage evaluations that differ only in fp32 accumulation order decorrelate at every rounding point, so
        # they sit ~sqrt(2) x the activation-rounding noise apart (measured 1.26-1.34e-3 here; the restatement itself is
        # 1.3-1.5e-3 from fp32 and fp16 WEIGHTS alone cost 0.84e-3: tests/test_precision_cpu.py).  north_star's "1e-3 rel"
        # is therefore below the noise floor of fp16-operand arithmetic on this network; the gates are the floor-relative
        # ones below ("UNet tolerance").
        assert r16 < 1.6e-3, f"UNet {cfgname} t={t}: {r16:.3e} vs the fp16-storage restatement"      # measured 1.26-1.34e-3 x 1.15
        assert r32 < 1.25 * floor + 2e-4, f"UNet {cfgname} t={t}: {r32:.3e} vs fp32 exceeds the fp16-storage floor {floor:.3e}"


def test_unet_sd2_depth_shapes_and_determinism(dev):
    """Full SD2-depth architecture (random init): 866 M parameters, runs at latent 32^2, deterministic, finite;
    algorithmic FLOP accounting equals SURVEY §8d (181.1 GFLOP per sample-forward at 32^2)."""
    from contexture_nerf_amd.unet import UNet2DConditionModel
    net = UNet2DConditionModel(device=dev, seed=0)
    assert abs(net.num_parameters() - 865.9e6) < 1.0e6
    fl = net.flops(2, 32, 32, 77)
    total = sum(v[1] for v in fl.values())
    assert abs(total / 2 - 181.1e9) / 181.1e9 < 2e-3
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 32, 32, generator=g).to(dev)
    ctx = torch.randn(2, 77, 1024, generator=g).to(dev)
    a = net(x, 501.0, ctx)['sample'].clone()
    b = net(x, 501.0, ctx)['sample']
    assert a.shape == (2, 4, 32, 32) and torch.isfinite(a).all()
    assert torch.equal(a, b)
    assert a.std() > 1e-3


@pytest.mark.parametrize("cfgname,h,w", [("tiny", 8, 8), ("tiny", 16, 8), ("sd", 8, 8)])
def test_vae_decode_vs_oracle(dev, cfgname, h, w):
    """VAE decoder engine vs the oracle's fp32 AutoencoderKL.decode restatement (random init, affines perturbed)."""
    from contexture_nerf_amd.vae import AutoencoderKL
    from oracle import vae_ref, unet_ref
    cfg = dict(vae_ref.SD_VAE) if cfgname == "sd" else dict(latent_channels=4, out_channels=3, block_out_channels=(64, 128), layers_per_block=1, groups=32)
    torch.manual_seed(3)
    ref = unet_ref.randomize_affine(vae_ref.AutoencoderKLDecodeRef(cfg)).eval()
    vae = AutoencoderKL(cfg, device=dev, init=False)
    vae.load_state_dict(ref.state_dict())
    z = torch.randn(1, 4, h, w)
    with torch.no_grad():
        want = ref.decode(z)
    got = vae.decode(z.to(dev)).sample
    up = 2 ** (len(cfg['block_out_channels']) - 1)
    assert got.shape == (1, 3, h * up, w * up) and torch.isfinite(got).all()
    r = _rel(got, want)
    print(f"vae {cfgname} {h}x{w}: rel L2 vs fp32 = {r:.3e}")
    assert r < 3e-3, r            # same fp16-storage noise floor as the UNet (tests/test_precision_cpu.py)
    assert torch.equal(got, vae.decode(z.to(dev)).sample)


@pytest.mark.parametrize("cfgname,H,W", [("tiny", 128, 128), ("tiny", 64, 32), ("sd", 64, 64)])
def test_vae_encode_vs_oracle(dev, cfgname, H, W):
    """VAE encoder engine (stride-2 convs with diffusers' one-sided padding) vs the oracle's fp32 AutoencoderKL.encode moments;
    a decoder-only checkpoint leaves encode off with a clear error."""
    from contexture_nerf_amd.vae import AutoencoderKL
    from contexture_nerf_amd._lib import CtxError
    from oracle import vae_ref, unet_ref
    cfg = dict(vae_ref.SD_VAE) if cfgname == "sd" else dict(latent_channels=4, out_channels=3, block_out_channels=(64, 128), layers_per_block=1, groups=32)
    torch.manual_seed(4)
    ref = unet_ref.randomize_affine(vae_ref.AutoencoderKLRef(cfg)).eval()
    vae = AutoencoderKL(cfg, device=dev, init=False)
    vae.load_state_dict(ref.state_dict())
    x = torch.rand(1, 3, H, W) * 2 - 1
    with torch.no_grad():
        want = ref.encode_moments(x)
    dist = vae.encode(x.to(dev)).latent_dist
    got = dist.parameters
    f = 2 ** (len(cfg['block_out_channels']) - 1)
    assert got.shape == (1, 8, H // f, W // f) and torch.isfinite(got).all()
    r = _rel(got, want)
    print(f"vae encode {cfgname} {H}x{W}: rel L2 vs fp32 = {r:.3e}")
    assert r < 3e-3, r
    g = torch.Generator(device=dev).manual_seed(1)
    smp = dist.sample(generator=g)
    assert smp.shape == (1, 4, H // f, W // f) and torch.equal(dist.mode(), got[:, :4])
    # decode still works on the same engine, and a decoder-only state_dict switches encode off
    z = torch.randn(1, 4, 8, 8)
    with torch.no_grad():
        wd = ref.decode(z)
    assert _rel(vae.decode(z.to(dev)).sample, wd) < 4e-3       # sanity only (test_vae_decode_vs_oracle holds the decode gate)
    dec_only = {k: v for k, v in ref.state_dict().items() if k.startswith(('decoder.', 'post_quant_conv.'))}
    vae.load_state_dict(dec_only)
    with pytest.raises(CtxError):
        vae.encode(x.to(dev))


@pytest.mark.parametrize("cfg_guidance", [True, False])
def test_ref_only_attention_vs_oracle(dev, cfg_guidance):
    """Zero123++'s reference-only self-attention as two engine passes ('w' over the noised condition latent, 'r' over a
    non-square sample with the extra K/V tokens) vs the oracle's restatement of src/zero123plus.py:127-237."""
    from contexture_nerf_amd.unet import UNet2DConditionModel
    from contexture_nerf_amd.zero123plus import RefOnlyNoisedUNet
    from contexture_nerf_amd.scheduler import DDPMScheduler
    from oracle import unet_ref
    cfg = unet_ref.tiny_config(in_channels=4)
    torch.manual_seed(6)
    ref = unet_ref.randomize_affine(unet_ref.UNet2DConditionModelRef(cfg)).eval()
    net = UNet2DConditionModel(cfg, device=dev, init=False)
    net.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(8)
    B = 2 if cfg_guidance else 1
    x = torch.randn(B, 4, 16, 8, generator=g)                  # the 3x2 view grid is not square (120 x 80 at full size)
    cond = 3 * torch.randn(1, 4, 16, 16, generator=g)          # noised condition latent: 256 extra tokens at level 0, 64 at level 1
    ctx = torch.randn(B, 9, cfg['cross_attention_dim'], generator=g)
    with torch.no_grad():
        want = unet_ref.ref_only_forward(ref, x, torch.tensor(321.0), ctx, cond, cfg_guidance)['sample']
        plain = ref(x, torch.tensor(321.0), ctx)['sample']
    # engine passes directly
    _, bank = net.forward_ref(cond.to(dev), 321.0, (ctx[1:] if cfg_guidance else ctx).to(dev), 'w')
    got = net.forward_ref(x.to(dev), 321.0, ctx.to(dev), 'r', bank=bank, ref_row0=1 if cfg_guidance else 0)[0]['sample']
    r = _rel(got, want)
    print(f"ref-only cfg={cfg_guidance}: rel L2 vs fp32 = {r:.3e}; reference tokens move the output by {_rel(plain.to(dev), want):.3e}")
    assert r < 3e-3, r
    assert _rel(plain.to(dev), want) > 4 * r                    # the parked tokens matter (well above the fp16 noise)
    # a plain forward afterwards is unaffected by the passes
    assert _rel(net(x.to(dev), 321.0, ctx.to(dev))['sample'], plain) < 3e-3
    # the host mirror of RefOnlyNoisedUNet: same two passes behind the reference's call shape (its own noise draw on cond_lat)
    wrap = RefOnlyNoisedUNet(net, DDPMScheduler(), DDPMScheduler()).eval()
    out = wrap(x.to(dev), torch.tensor([321]), ctx.to(dev), cross_attention_kwargs=dict(cond_lat=cond.to(dev), is_cfg_guidance=cfg_guidance))
    assert out['sample'].shape == x.shape and torch.isfinite(out['sample']).all()


def test_controlnet_residuals_vs_oracle(dev):
    """ControlNet engine (conditioning embedding + zero convolutions) and the UNet's residual injection vs the oracle's
    ControlNetModel / UNet restatement; then the DepthControlUNet(RefOnlyNoisedUNet(...)) mirror end to end."""
    from contexture_nerf_amd.unet import UNet2DConditionModel, ControlNetModel
    from contexture_nerf_amd.zero123plus import RefOnlyNoisedUNet, DepthControlUNet
    from contexture_nerf_amd.scheduler import DDPMScheduler
    from oracle import unet_ref
    cfg = unet_ref.tiny_config(in_channels=4)
    torch.manual_seed(12)
    ref = unet_ref.randomize_affine(unet_ref.UNet2DConditionModelRef(cfg)).eval()
    cref = unet_ref.randomize_affine(unet_ref.ControlNetModelRef(cfg), seed=1).eval()
    net = UNet2DConditionModel(cfg, device=dev, init=False); net.load_state_dict(ref.state_dict())
    cnet = ControlNetModel(cfg, device=dev, init=False); cnet.load_state_dict(cref.state_dict())
    assert set(cnet.param_shapes()) == set(cref.state_dict().keys())
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 4, 16, 8, generator=g)
    ctx = torch.randn(2, 9, cfg['cross_attention_dim'], generator=g)
    depth = torch.rand(2, 3, 128, 64, generator=g)
    with torch.no_grad():
        wd, wm = cref(x, torch.tensor(77.0), ctx, depth, conditioning_scale=2.0)
        want = ref(x, torch.tensor(77.0), ctx, down_block_additional_residuals=wd, mid_block_additional_residual=wm)['sample']
        plain = ref(x, torch.tensor(77.0), ctx)['sample']
    res, mid = cnet(x.to(dev), 77.0, encoder_hidden_states=ctx.to(dev), controlnet_cond=depth.to(dev), conditioning_scale=2.0)
    assert len(res) == len(wd) + 1
    for k, w in enumerate(wd + [wm]):
        got = res[k].float().permute(0, 3, 1, 2) * 2.0                  # the engine hands the residuals over unscaled
        assert _rel(got, w) < 3e-3, (k, _rel(got, w))
    got = net(x.to(dev), 77.0, encoder_hidden_states=ctx.to(dev), down_block_additional_residuals=res, mid_block_additional_residual=mid)['sample']
    r = _rel(got, want)
    print(f"controlnet + unet: rel L2 vs fp32 = {r:.3e}; the residuals move the output by {_rel(plain, want):.3e}")
    assert r < 3e-3 and _rel(plain, want) > 4 * r
    assert _rel(net(x.to(dev), 77.0, encoder_hidden_states=ctx.to(dev))['sample'], plain) < 3e-3      # switched off again
    # same conditioning tensor again: the cached embedding path gives the same bits; a changed image is picked up
    dd = depth.to(dev)
    r1, _ = cnet(x.to(dev), 77.0, encoder_hidden_states=ctx.to(dev), controlnet_cond=dd)
    b1 = r1.buffer.clone()
    r2, _ = cnet(x.to(dev), 77.0, encoder_hidden_states=ctx.to(dev), controlnet_cond=dd)
    assert torch.equal(b1, r2.buffer)
    dd.mul_(0.5)
    r3, _ = cnet(x.to(dev), 77.0, encoder_hidden_states=ctx.to(dev), controlnet_cond=dd)
    assert not torch.equal(b1, r3.buffer)
    # the reference's wrapper stack: DepthControlUNet(RefOnlyNoisedUNet(unet)) with cond_lat / control_depth / is_cfg_guidance
    stack = DepthControlUNet(RefOnlyNoisedUNet(net, DDPMScheduler(), DDPMScheduler()).eval(), cnet, conditioning_scale=2.0).eval()
    cond_lat = torch.randn(1, 4, 8, 8, generator=g)
    out = stack(x.to(dev), torch.tensor([77]), ctx.to(dev),
                cross_attention_kwargs=dict(cond_lat=cond_lat.to(dev), control_depth=depth.to(dev), is_cfg_guidance=True))
    assert out['sample'].shape == x.shape and torch.isfinite(out['sample']).all()


def test_new_entry_points_fail_loudly(dev):
    """error behaviour of this round's entry points: wrong shapes / wrong order raise CtxError with a message, nothing is
    silently computed on a fallback path."""
    from contexture_nerf_amd.unet import UNet2DConditionModel, ControlNetModel
    from contexture_nerf_amd.vae import AutoencoderKL
    from contexture_nerf_amd import run_nerf_helpers as rnh
    from contexture_nerf_amd._lib import CtxError
    from oracle import unet_ref
    cfg = unet_ref.tiny_config(in_channels=4)
    net = UNet2DConditionModel(cfg, device=dev, seed=1)
    x = torch.randn(2, 4, 16, 8, device=dev); ctx = torch.randn(2, 9, cfg['cross_attention_dim'], device=dev)
    with pytest.raises(CtxError, match="bank"):
        net.forward_ref(x, 10.0, ctx, 'r')                               # no bank at all
    other = UNet2DConditionModel(cfg, device=dev, seed=2)
    _, bank = other.forward_ref(x[:1], 10.0, ctx[:1], 'w')
    with pytest.raises(CtxError, match="reference bank"):
        net.forward_ref(x, 10.0, ctx, 'r', bank=bank, ref_row0=1)        # this engine never ran a 'w' pass
    with pytest.raises(CtxError):
        net.forward_ref(x, 10.0, ctx, 'q')
    cnet = ControlNetModel(cfg, device=dev, seed=3)
    with pytest.raises(CtxError, match="8x the latent grid"):
        cnet(x, 10.0, encoder_hidden_states=ctx, controlnet_cond=torch.rand(2, 3, 64, 64, device=dev))
    with pytest.raises(CtxError):
        net(x, 10.0, encoder_hidden_states=ctx, down_block_additional_residuals=[torch.zeros(1, device=dev)])
    vae = AutoencoderKL(dict(latent_channels=4, out_channels=3, block_out_channels=(64, 128), layers_per_block=1, groups=32), device=dev)
    with pytest.raises(CtxError, match="multiples"):
        vae.encode(torch.rand(1, 3, 30, 30, device=dev))
    with pytest.raises(CtxError):
        rnh.NeRF2D(D=8, W=256, input_ch=70, output_ch=3, skips=[4]).to(dev).packed()      # beyond the fused kernel's envelope
    field = rnh.NeRF2D(D=8, W=64, input_ch=63, output_ch=4, skips=[4]).to(dev)
    with pytest.raises(CtxError):
        with torch.no_grad():
            field.texture_map(16)                                         # the atlas grid is a 2-D input; this field is 3-D


def test_unet_full_size_vs_oracle(dev):
    """BASELINE configs[1] at full size: the SD2-depth UNet (866 M parameters, seeded random init shared through the state_dict),
    CFG batch 2, latent 96 x 96, 77 context tokens — engine (tuned plans, 256x256 kernel, split-K, fused GroupNorm...) against
      * the fp32 oracle,
      * the fp16-storage restatement of the reference's autocast contract (oracle.unet_ref.forward_fp16_storage: the same rounding
        points as the engine), and
      * that restatement with every K sum taken in another order (perm): how far two CORRECT implementations of one contract are
        from each other (tests/test_precision_cpu.py shows why: fp32-order noise flips fp16 roundings, the flips decorrelate).
    Gates, self-calibrating: the engine may be no further from the contract than 1.25 x what the contract is from itself, and no
    further from fp32 than 1.15 x what the contract is (measured: 1.18e-3 vs 1.12e-3; 1.39e-3 vs 1.36e-3;
    profiles/r03_precision_attribution.json has the same comparison after every one of the 45 blocks)."""
    from contexture_nerf_amd.unet import UNet2DConditionModel
    from oracle import unet_ref
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    torch.manual_seed(0)
    ref = unet_ref.randomize_affine(unet_ref.UNet2DConditionModelRef(unet_ref.SD2_DEPTH)).eval()
    net = UNet2DConditionModel(device=dev, init=False)
    net.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 5, 96, 96, generator=g); ctx = torch.randn(2, 77, 1024, generator=g)
    t = torch.tensor(501.0)
    with torch.no_grad():
        want = ref(x, t, ctx)['sample']
    fused = unet_ref.forward_fp16_storage(ref, x, t, ctx)['sample']
    perm = unet_ref.forward_fp16_storage(ref, x, t, ctx, perm=unet_ref._PermLinear(4))['sample']
    got = net(x.to(dev), 501.0, ctx.to(dev))['sample']
    r32, r16, floor16, floor32 = _rel(got, want), _rel(got, fused), _rel(perm, fused), _rel(fused, want)
    print(f"full-size UNet (latent 96, batch 2): engine vs fp32 {r32:.3e} | engine vs fp16 contract {r16:.3e} | contract vs itself in another "
          f"summation order {floor16:.3e} | contract vs fp32 {floor32:.3e}")
    assert torch.isfinite(got).all()
    assert r16 < 1.25 * floor16, (r16, floor16)
    assert r32 < 1.15 * floor32 and r32 < 1.6e-3, (r32, floor32)


def test_zero123pp_full_size_vs_oracle(dev):
    """BASELINE configs[2] at FULL size: the Zero123++-shaped denoise evaluation — SD2-family UNet (in_channels 4, 320/640/1280/1280),
    CFG batch 2 on the 3x2 view grid latent [2,4,120,80], the noised 40x40 condition latent parked by the 'w' pass (1 600 reference
    tokens appended to the level-0 self-attention K/V of the conditional row; 400 / 100 at the deeper levels), depth ControlNet
    over the 960x640 grid with conditioning scale 2 and its 13 residuals injected into the 'r' pass — against the fp32 oracle's
    restatement of src/zero123plus.py:127-298 on the host cores.  Gate: measured x 1.15 (the fp16 contract's own floor is 1.3-1.5e-3)."""
    from contexture_nerf_amd.unet import UNet2DConditionModel, ControlNetModel
    from oracle import unet_ref
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    cfg = dict(unet_ref.SD2_DEPTH, in_channels=4)
    torch.manual_seed(3)
    ref = unet_ref.randomize_affine(unet_ref.UNet2DConditionModelRef(cfg)).eval()
    cref = unet_ref.randomize_affine(unet_ref.ControlNetModelRef(cfg), seed=1).eval()
    net = UNet2DConditionModel(cfg, device=dev, init=False); net.load_state_dict(ref.state_dict())
    cnet = ControlNetModel(cfg, device=dev, init=False); cnet.load_state_dict(cref.state_dict())
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 4, 120, 80, generator=g)
    cond = 2 * torch.randn(1, 4, 40, 40, generator=g)
    ctx = torch.randn(2, 77, 1024, generator=g)
    depth = torch.rand(1, 3, 960, 640, generator=g).expand(2, -1, -1, -1).contiguous()
    t = 515.0
    with torch.no_grad():
        wd, wm = cref(x, torch.tensor(t), ctx, depth, conditioning_scale=2.0)
        want = unet_ref.ref_only_forward(ref, x, torch.tensor(t), ctx, cond, True, down_res=wd, mid_res=wm)['sample']
    res, mid = cnet(x.to(dev), t, encoder_hidden_states=ctx.to(dev), controlnet_cond=depth.to(dev), conditioning_scale=2.0)
    for k, w in enumerate(wd + [wm]):
        rk = _rel(res[k].float().permute(0, 3, 1, 2) * 2.0, w)
        assert rk < 3e-3, (k, rk)
    _, bank = net.forward_ref(cond.to(dev), t, ctx[1:].to(dev), 'w')
    with net.residuals(res):
        got = net.forward_ref(x.to(dev), t, ctx.to(dev), 'r', bank=bank, ref_row0=1)[0]['sample']
    r = _rel(got, want)
    with net.residuals(res):
        no_ref = net(x.to(dev), t, encoder_hidden_states=ctx.to(dev))['sample']
    print(f"full-size Zero123++ evaluation (latent 120x80, 1600 ref tokens, ControlNet): rel L2 vs fp32 = {r:.3e}; "
          f"without the reference tokens {_rel(no_ref, want):.3e}")
    assert torch.isfinite(got).all() and r < 1.7e-3, r                # measured 1.33e-3 (x 1.15 + box / plan-table spread)
    assert _rel(no_ref, want) > 4 * r


def test_engines_load_safetensors_files(dev, tmp_path):
    """Checkpoint-from-file path (the local stand-in for `from_pretrained`, src/stable_diffusion_depth.py:58-88): a diffusers-layout
    directory written by this test (unet/ and vae/ safetensors in fp16 and fp32) loads through safetensors_io into the engines and
    gives the same bits as load_state_dict of the same tensors; a file with a missing tensor is refused."""
    from contexture_nerf_amd.unet import UNet2DConditionModel
    from contexture_nerf_amd.vae import AutoencoderKL
    from contexture_nerf_amd.stable_diffusion_depth import StableDiffusion
    from contexture_nerf_amd import safetensors_io as sio
    from contexture_nerf_amd._lib import CtxError
    from oracle import unet_ref, vae_ref
    cfg = unet_ref.tiny_config()
    torch.manual_seed(4)
    ref = unet_ref.randomize_affine(unet_ref.UNet2DConditionModelRef(cfg)).eval()
    sd16 = {k: v.half() for k, v in ref.state_dict().items()}
    os.makedirs(tmp_path / "unet"); os.makedirs(tmp_path / "vae")
    up = str(tmp_path / "unet" / "diffusion_pytorch_model.safetensors")
    sio.save_file(sd16, up)
    a = UNet2DConditionModel(cfg, device=dev, init=False); a.load_state_dict({k: v.float() for k, v in sd16.items()})
    b = UNet2DConditionModel.from_file(up, cfg, device=dev)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 5, 16, 16, generator=g).to(dev); ctx = torch.randn(2, 7, cfg['cross_attention_dim'], generator=g).to(dev)
    assert torch.equal(a(x, 300.0, ctx)['sample'], b(x, 300.0, ctx)['sample'])
    vcfg = dict(latent_channels=4, out_channels=3, block_out_channels=(64, 128), layers_per_block=1, groups=32)
    vref = vae_ref.AutoencoderKLRef(vcfg).eval()
    vp = str(tmp_path / "vae" / "diffusion_pytorch_model.safetensors")
    sio.save_file(vref.state_dict(), vp)
    va = AutoencoderKL(vcfg, device=dev, init=False); va.load_state_dict(vref.state_dict())
    vb = AutoencoderKL.from_file(vp, vcfg, device=dev)
    z = torch.randn(1, 4, 8, 8, generator=g).to(dev)
    assert torch.equal(va.decode(z).sample, vb.decode(z).sample)
    bad = dict(sd16); bad.pop("conv_in.weight")
    sio.save_file(bad, up)
    with pytest.raises(CtxError, match="missing"):
        UNet2DConditionModel.from_file(up, cfg, device=dev)


@pytest.mark.parametrize("cfgname,B,H,W", [("tiny", 1, 128, 128), ("tiny", 2, 64, 32), ("sd", 1, 64, 64), ("sd", 1, 192, 128)])
def test_vae_encode_backward_vs_oracle(dev, cfgname, B, H, W):
    """Autograd of `vae.encode` (the link between the SDS loss and the rendered views, src/training/trainer.py:732, 866): the
    engine's input gradient (conv / linear data-gradients on the forward's MFMA kernels with transposed packs, the stride-2
    downsamplers through the zero-inserted grid, GroupNorm+SiLU and softmax-attention backward) vs torch autograd over the
    oracle's fp32 Encoder, for a random cotangent on the moments.  fp16 gradients with a power-of-two scale: gate 1e-2 rel L2."""
    from contexture_nerf_amd.vae import AutoencoderKL
    from oracle import vae_ref, unet_ref
    cfg = dict(vae_ref.SD_VAE) if cfgname == "sd" else dict(latent_channels=4, out_channels=3, block_out_channels=(64, 128), layers_per_block=1, groups=32)
    torch.manual_seed(9)
    ref = unet_ref.randomize_affine(vae_ref.AutoencoderKLRef(cfg)).eval()
    vae = AutoencoderKL(cfg, device=dev, init=False)
    vae.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(2)
    x = (torch.rand(B, 3, H, W, generator=g) * 2 - 1)
    f = 2 ** (len(cfg['block_out_channels']) - 1)
    cot = torch.randn(B, 8, H // f, W // f, generator=g) * 0.03              # the SDS cotangent is O(0.01-0.1)
    xr = x.clone().requires_grad_(True)
    (ref.encode_moments(xr) * cot).sum().backward()
    xg = x.to(dev).requires_grad_(True)
    mom = vae.encode_moments_with_grad(xg)
    with torch.no_grad():
        assert _rel(mom, ref.encode_moments(x)) < 3e-3
    (mom * cot.to(dev)).sum().backward()
    r = _rel(xg.grad, xr.grad)
    print(f"vae encode backward {cfgname} B={B} {H}x{W}: rel L2 of d(image) vs fp32 autograd = {r:.3e}, |grad| = {float(xr.grad.norm()):.3e}")
    assert torch.isfinite(xg.grad).all() and r < 1e-2, r
    # linear in the cotangent (gscale drops out), and the public seam: encode() of a tensor that requires grad carries autograd
    xg2 = x.to(dev).requires_grad_(True)
    (vae.encode(xg2).latent_dist.parameters * (4.0 * cot.to(dev))).sum().backward()
    assert _rel(xg2.grad, (4.0 * xg.grad).cpu()) < 2e-3
    # a no-grad call between a training forward and its backward runs on a sibling handle (same weights, own workspace): the
    # tape survives it (the SDS loop encodes the condition image between the grid's encode and loss.backward())
    xg3 = x.to(dev).requires_grad_(True)
    m3 = vae.encode_moments_with_grad(xg3)
    other = vae.encode(x.to(dev)).latent_dist.parameters
    assert torch.equal(other, mom.detach())
    (m3 * cot.to(dev)).sum().backward()
    assert torch.equal(xg3.grad, xg.grad)
    # at the C-ABI the tape is single-use: a second backward without a forward is refused
    from contexture_nerf_amd._lib import CtxError
    L, _ = _lib()
    with pytest.raises(CtxError, match="tape"):
        L.check(vae._lib.ctx_vae_encode_bwd(vae._h, L.ptr(cot.to(dev).contiguous()), 1.0, L.ptr(torch.empty_like(xg.grad)), L.stream()))


def test_unet_fp32_residual_stream(dev):
    """The fp32-residual-stream variant (block outputs, skip tensors and the transformer's running sums kept in fp32; operands
    and weights fp16) against the fp32 oracle, beside the default fp16 stream: small and mid configurations here, the SD2-depth
    UNet at latent 96 in tools/bench_precision.py (numbers in DESIGN section 7).  The variant must not be worse than the default."""
    from contexture_nerf_amd.unet import UNet2DConditionModel
    from oracle import unet_ref
    for cfg, hw in ((unet_ref.tiny_config(), (16, 16)), (unet_ref.tiny_config(ch=(64, 128, 256, 256), heads=(1, 2, 4, 4), ctx_dim=128), (32, 32))):
        torch.manual_seed(8)
        ref = unet_ref.randomize_affine(unet_ref.UNet2DConditionModelRef(cfg)).eval()
        net = UNet2DConditionModel(cfg, device=dev, init=False); net.load_state_dict(ref.state_dict())
        g = torch.Generator().manual_seed(2)
        x = torch.randn(2, 5, *hw, generator=g); ctx = torch.randn(2, 11, cfg['cross_attention_dim'], generator=g)
        with torch.no_grad():
            want = ref(x, torch.tensor(333.0), ctx)['sample']
        r16 = _rel(net(x.to(dev), 333.0, ctx.to(dev))['sample'], want)
        net.set_residual_fp32(True)
        y32 = net(x.to(dev), 333.0, ctx.to(dev))['sample']
        r32 = _rel(y32, want)
        assert torch.equal(y32, net(x.to(dev), 333.0, ctx.to(dev))['sample'])            # deterministic
        net.set_residual_fp32(False)
        assert _rel(net(x.to(dev), 333.0, ctx.to(dev))['sample'], want) == r16            # switches back cleanly
        print(f"residual stream fp16 {r16:.3e} vs fp32 {r32:.3e} (rel L2 vs the fp32 oracle, channels {cfg['block_out_channels']})")
        assert r32 < 2.5e-3 and r32 <= r16 * 1.05
