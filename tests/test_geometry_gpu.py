"""GPU parity: HIP raster / uv / view-weight / texture-field kernels (through the C-ABI) against the
oracle on the same seeded inputs.  Integer and float32 geometry results must be BIT-EXACT (both sides run
uncontracted IEEE binary32 in the same order); transcendental paths carry an explicit tolerance."""
import numpy as np
import pytest
import torch

from oracle import geometry as og, nerf as onerf

pytestmark = pytest.mark.gpu

ZP_THETA = np.float32([1.0471975803375244] * 4 + [1.919862151145935] * 3)       # Zero123PlusDataset (golden meta)
ZP_PHI = np.deg2rad(np.float32([0, 30, 150, 270, 90, 210, 330])).astype(np.float32)


def _scene(meshes, name, B, dy=0.25):
    v = og.normalize_mesh(meshes[name + '_v'], 0.6, dy)
    f = meshes[name + '_f'].astype(np.int64)
    cam = og.get_camera_from_multiple_view(ZP_THETA[:B], ZP_PHI[:B], np.full(B, 1.5, np.float32), dy)
    proj = og.generate_perspective_projection(np.pi / 3)
    return np.repeat(v[None], B, 0), f, cam, proj


def _uv_attr(meshes, name, F, seed=0):
    vt, ft = meshes[name + '_vt'], meshes[name + '_ft']
    if vt.shape[0] and ft.min() >= 0:
        return vt[ft.astype(np.int64)][None].astype(np.float32)                     # [1,F,3,2]
    return np.random.default_rng(seed).random((1, F, 3, 2), dtype=np.float32)


def test_mfma_lane_maps(dev):
    """The fragment maps every MFMA kernel relies on, with asymmetric integer data (exact in f16/f32)."""
    from contexture_nerf_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(0)
    A = torch.randint(-4, 5, (32, 16), generator=g).float()
    B = torch.randint(-4, 5, (32, 16), generator=g).float()
    C = torch.zeros(32, 32, device=dev)
    Ad, Bd = A.half().to(dev), B.half().to(dev)          # keep alive across the call
    L.check(lib.ctx_probe_mfma(0, L.ptr(Ad), L.ptr(Bd), L.ptr(C), L.stream()))
    assert torch.equal(C.cpu(), A @ B.T)
    A2 = torch.randint(-9, 10, (32, 2), generator=g).float()
    B2 = torch.randint(-9, 10, (32, 2), generator=g).float()
    A2d, B2d = A2.to(dev), B2.to(dev)
    L.check(lib.ctx_probe_mfma(1, L.ptr(A2d), L.ptr(B2d), L.ptr(C), L.stream()))
    assert torch.equal(C.cpu(), A2 @ B2.T)


def test_tr16_read_map(dev):
    """The transposing LDS read the attention kernel feeds its PV MFMAs with (V tile kept [key][d], no transpose pass)."""
    from contexture_nerf_amd import _lib as L
    lib = L.load()
    A = torch.arange(8 * 32, dtype=torch.float32).view(8, 32).half()
    Ad = A.to(dev); Cm = torch.zeros(64, 4, device=dev)
    L.check(lib.ctx_probe_mfma(2, L.ptr(Ad), L.ptr(Ad), L.ptr(Cm), L.stream()))
    got = Cm.cpu()
    for l in range(64):
        a, h, i = (l >> 4) & 1, l >> 5, l & 15
        want = A[4 * h:4 * h + 4, 16 * a + i].float()
        assert torch.equal(got[l], want), (l, got[l], want)


@pytest.mark.parametrize("name,B,H,W", [("sphere", 2, 64, 64), ("spot_triangulated", 3, 256, 256),
                                        ("nascar", 7, 200, 200), ("bunny", 1, 97, 131),
                                        ("blub_no_texture", 2, 320, 320), ("env_sphere", 1, 128, 128)])
def test_prepare_and_raster_bit_exact(dev, meshes, name, B, H, W):
    from contexture_nerf_amd import kal
    verts, f, cam, proj = _scene(meshes, name, B)
    o_cam, o_img, o_fn = og.prepare_vertices(verts, f, proj, cam)
    g_cam, g_img, g_fn = kal.render.mesh.prepare_vertices(torch.tensor(verts, device=dev), torch.tensor(f, device=dev),
                                                          torch.tensor(proj), camera_transform=torch.tensor(cam, device=dev))
    assert np.array_equal(g_cam.cpu().numpy(), o_cam)
    assert np.array_equal(g_img.cpu().numpy(), o_img)
    assert np.array_equal(g_fn.cpu().numpy(), o_fn)
    uva = _uv_attr(meshes, name, f.shape[0])
    # reference form: two passes (depth feature, then uv feature)
    o_d, o_i = og.rasterize(H, W, o_cam[..., 2], o_img, o_cam[..., 2:3])
    o_uv, o_i2 = og.rasterize(H, W, o_cam[..., 2], o_img, np.repeat(uva, B, 0))
    assert np.array_equal(o_i, o_i2)
    g_d, g_i = kal.render.mesh.rasterize(H, W, g_cam[..., 2], g_img, g_cam[..., 2:3])
    g_uv, g_i2 = kal.render.mesh.rasterize(H, W, g_cam[..., 2], g_img, torch.tensor(uva, device=dev).repeat(B, 1, 1, 1))
    assert g_i.dtype == torch.int64
    assert np.array_equal(g_i.cpu().numpy(), o_i), f"{(g_i.cpu().numpy() != o_i).sum()} face ids differ"
    assert np.array_equal(g_i2.cpu().numpy(), o_i)
    assert np.array_equal(g_d.cpu().numpy(), o_d)
    assert np.array_equal(g_uv.cpu().numpy(), o_uv)
    # fused single pass == both passes + normals gather
    f_d, f_uv, f_i, f_n = kal.render.mesh.rasterize_fused(H, W, g_cam, g_img, torch.tensor(uva, device=dev), g_fn)
    assert np.array_equal(f_i.cpu().numpy(), o_i)
    assert np.array_equal(f_d.cpu().numpy(), o_d)
    assert np.array_equal(f_uv.cpu().numpy(), o_uv)
    assert np.array_equal(f_n.cpu().numpy(), og.gather_normals(o_i, o_fn))
    assert (o_i >= 0).mean() > 0.02


@pytest.mark.parametrize("name", ["front", "side"])
def test_product_raster_chain_vs_reference_depth_fixtures(dev, meshes, golden, golden_meta, name):
    """PINNED against kaolin's own output as the reference holds it (shapes/spot_depth_{front,side}.pt, stored by
    tests/golden/make_golden.py): the PRODUCT chain Mesh.normalize_mesh -> Renderer.get_camera_from_multiple_view ->
    kal prepare_vertices (HIP) -> rasterize_fused (HIP) at the reference's default 1200^2 grid -> Renderer.normalize_multiple_depth
    (HIP) re-based to [0.5, 1] -> utils.get_nonzero_region_tuple.  Silhouette equal on every pixel of the crop, depth within 1e-4
    (to rounding once the fixture's own min / max are fitted); and the 1200^2 face ids / depths equal the oracle's bit for bit."""
    from contexture_nerf_amd import kal, utils
    from contexture_nerf_amd.mesh import Mesh
    from contexture_nerf_amd.render import Renderer
    from test_oracle_golden import check_against_spot_fixture
    meta = golden_meta['spot_depth_' + name]
    ref = golden['spot_depth_' + name]
    G = meta['grid']
    mesh = Mesh(device=dev, arrays=(meshes['spot_triangulated_v'], meshes['spot_triangulated_f'],
                                    meshes['spot_triangulated_vt'], meshes['spot_triangulated_ft']))
    mesh = mesh.normalize_mesh(inplace=True, target_scale=meta['scale'], dy=meta['dy'])
    R = Renderer(dev, dim=(G, G))
    theta = torch.tensor([np.deg2rad(meta['theta_deg'])], dtype=torch.float32, device=dev)
    phi = torch.tensor([np.deg2rad(meta['phi_deg'])], dtype=torch.float32, device=dev)
    radius = torch.tensor([meta['radius']], dtype=torch.float32, device=dev)
    cam = R.get_camera_from_multiple_view(theta, phi, radius, look_at_height=meta['dy'])
    fvc, fvi, fn = kal.render.mesh.prepare_vertices(mesh.vertices[None], mesh.faces, R.camera_projection, camera_transform=cam)
    uva = kal.ops.mesh.index_vertices_by_faces(mesh.vt[None].to(dev), mesh.ft.to(dev))
    raw, uv, idx, normals = kal.render.mesh.rasterize_fused(G, G, fvc, fvi, uva, fn)
    depth = R.normalize_multiple_depth(raw)                               # render.py:48-74 with min_val = 0
    mask = idx > -1
    depth = torch.where(mask[..., None], 0.5 * depth + 0.5, depth)         # the fixtures' min_val = 0.5 (render.py:64-67)
    h0, w0, h1, w1 = utils.get_nonzero_region_tuple(mask[0].float())
    assert (h1 - h0, w1 - w0) == ref.shape
    check_against_spot_fixture(depth[0, h0:h1, w0:w1, 0].cpu().numpy(), ref)
    # the same 1200^2 raster against the oracle on the product's own vertices and camera: bit for bit
    o_cam, o_img, o_fn = og.prepare_vertices(mesh.vertices[None].cpu().numpy(), mesh.faces.cpu().numpy(),
                                             R.camera_projection.cpu().numpy(), cam.cpu().numpy())
    assert np.array_equal(fvc.cpu().numpy(), o_cam) and np.array_equal(fvi.cpu().numpy(), o_img)
    o_d, o_i = og.rasterize(G, G, o_cam[..., 2], o_img, o_cam[..., 2:3])
    o_uv, _ = og.rasterize(G, G, o_cam[..., 2], o_img, uva.cpu().numpy())
    assert np.array_equal(idx.cpu().numpy(), o_i), f"{(idx.cpu().numpy() != o_i).sum()} face ids differ at 1200^2"
    assert np.array_equal(raw.cpu().numpy(), o_d) and np.array_equal(uv.cpu().numpy(), o_uv)
    assert np.array_equal(normals.cpu().numpy(), og.gather_normals(o_i, o_fn))
    assert np.array_equal(R.normalize_multiple_depth(raw).cpu().numpy(), og.normalize_multiple_depth(o_d))


@pytest.mark.parametrize("name,B", [("nascar", 7), ("blub_no_texture", 1)])
def test_raster_default_grid_bit_exact(dev, meshes, name, B):
    """The reference's default render grid (1200^2, src/configs/train_config.py:11) with all 7 Zero123++ views: the size at
    which k_raster_bin splits a tile's face scan into ranges.  Oracle = the face-major walk (== brute force, CPU test)."""
    from contexture_nerf_amd import kal
    verts, f, cam, proj = _scene(meshes, name, B)
    o_cam, o_img, o_fn = og.prepare_vertices(verts, f, proj, cam)
    g_cam, g_img, g_fn = kal.render.mesh.prepare_vertices(torch.tensor(verts, device=dev), torch.tensor(f, device=dev),
                                                          torch.tensor(proj), camera_transform=torch.tensor(cam, device=dev))
    uva = _uv_attr(meshes, name, f.shape[0])
    H = W = 1200
    o_d, o_i = og.rasterize(H, W, o_cam[..., 2], o_img, o_cam[..., 2:3])
    o_uv, _ = og.rasterize(H, W, o_cam[..., 2], o_img, np.repeat(uva, B, 0))
    f_d, f_uv, f_i, f_n = kal.render.mesh.rasterize_fused(H, W, g_cam, g_img, torch.tensor(uva, device=dev), g_fn)
    assert np.array_equal(f_i.cpu().numpy(), o_i), f"{(f_i.cpu().numpy() != o_i).sum()} face ids differ"
    assert np.array_equal(f_d.cpu().numpy(), o_d)
    assert np.array_equal(f_uv.cpu().numpy(), o_uv)
    assert np.array_equal(f_n.cpu().numpy(), og.gather_normals(o_i, o_fn))
    g_d, g_i = kal.render.mesh.rasterize(H, W, g_cam[..., 2], g_img, g_cam[..., 2:3])
    assert np.array_equal(g_i.cpu().numpy(), o_i) and np.array_equal(g_d.cpu().numpy(), o_d)


def test_raster_edge_cases(dev):
    """Degenerate (zero-area) faces, exact depth ties (lowest index must win), faces off-screen, 1x1 image."""
    from contexture_nerf_amd import kal
    fxy = np.float32([[[-0.5, -0.5], [0.5, -0.5], [0.0, 0.5]],        # face 0
                      [[-0.5, -0.5], [0.5, -0.5], [0.0, 0.5]],        # face 1 == face 0 (tie)
                      [[0.2, 0.2], [0.2, 0.2], [0.2, 0.2]],           # degenerate
                      [[3.0, 3.0], [4.0, 3.0], [3.5, 4.0]],           # off-screen
                      [[-0.9, 0.1], [-0.1, 0.1], [-0.5, 0.9]]])[None]  # partly overlapping, nearer
    fz = np.float32([[-2, -2, -2], [-2, -2, -2], [-1, -1, -1], [-1, -1, -1], [-1.5, -1.2, -1.7]])[None]
    feat = np.random.default_rng(0).random((1, 5, 3, 3), dtype=np.float32)
    for (H, W) in [(33, 47), (1, 1), (8, 300)]:
        o, oi = og.rasterize(H, W, fz, fxy, feat)
        g, gi = kal.render.mesh.rasterize(H, W, torch.tensor(fz, device=dev), torch.tensor(fxy, device=dev),
                                          torch.tensor(feat, device=dev))
        assert np.array_equal(gi.cpu().numpy(), oi)
        assert np.array_equal(g.cpu().numpy(), o)
    assert 1 not in np.unique(oi)          # tie: face 0 hides its duplicate
    # all-background view
    o, oi = og.rasterize(16, 16, fz[:, 3:4], fxy[:, 3:4], feat[:, 3:4])
    g, gi = kal.render.mesh.rasterize(16, 16, torch.tensor(fz[:, 3:4], device=dev), torch.tensor(fxy[:, 3:4], device=dev),
                                      torch.tensor(feat[:, 3:4], device=dev))
    assert (gi.cpu().numpy() == -1).all() and np.array_equal(g.cpu().numpy(), o)


def test_normalize_depth(dev, meshes, golden):
    from contexture_nerf_amd import render
    d = torch.tensor(golden['depth_raw'], device=dev)
    out = render.Renderer.normalize_multiple_depth(None, d)
    assert np.array_equal(out.cpu().numpy(), golden['depth_norm'])          # vs the reference's own output
    with pytest.raises(AssertionError, match='negative'):
        render.Renderer.normalize_multiple_depth(None, -d)
    with pytest.raises(AssertionError, match='empty'):
        render.Renderer.normalize_multiple_depth(None, torch.zeros_like(d))
    big = -(torch.rand(3, 300, 301, 1, device=dev) + 0.25)
    big[:, :50] = 0
    assert np.array_equal(render.Renderer.normalize_multiple_depth(None, big).cpu().numpy(),
                          og.normalize_multiple_depth(big.cpu().numpy()))


def test_texture_mapping_fwd_bwd(dev):
    from contexture_nerf_amd import kal
    g = torch.Generator().manual_seed(0)
    uv = torch.rand(3, 70, 90, 2, generator=g) * 1.1 - 0.05
    tex = torch.rand(1, 3, 64, 64, generator=g)
    for mode in ('bilinear', 'nearest'):
        o = og.texture_mapping(uv.numpy(), tex.numpy(), mode)
        out = kal.render.mesh.texture_mapping(uv.to(dev), tex.to(dev).expand(3, -1, -1, -1), mode=mode)
        assert np.array_equal(out.cpu().numpy(), o)                       # bit-exact float32
        ref = torch.nn.functional.grid_sample(tex.expand(3, -1, -1, -1), torch.stack([uv[..., 0], 1 - uv[..., 1]], -1) * 2 - 1,
                                              mode=mode, align_corners=False, padding_mode='border').permute(0, 2, 3, 1)
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)
    # few pixels per texel -> the planar kernel; many -> the texel-interleaved one (3x70x90 above); both bit-exact, also masked
    uv2 = torch.rand(1, 30, 40, 2, generator=g) * 1.1 - 0.05
    for uvx in (uv2, uv):
        mi = (torch.rand(uvx.shape[:3], generator=g) > 0.3).long() * 5 - 1
        o = og.texture_mapping(uvx.numpy(), tex.numpy(), 'bilinear') * (mi.numpy() > -1)[..., None]
        out = kal.render.mesh.texture_mapping(uvx.to(dev), tex.to(dev).expand(uvx.shape[0], -1, -1, -1), mask_idx=mi.to(dev))
        assert np.array_equal(out.cpu().numpy(), o.astype(np.float32))
    # backward (float atomics: order-dependent rounding -> tolerance, stated here: 1e-5 abs on O(10) sums)
    texd = tex.to(dev).requires_grad_(True)
    y = kal.render.mesh.texture_mapping(uv.to(dev), texd.expand(3, -1, -1, -1))
    go = torch.rand(y.shape, generator=g)
    (y * go.to(dev)).sum().backward()
    gt = og.texture_mapping_bwd(go.numpy(), uv.numpy(), 64)
    np.testing.assert_allclose(texd.grad[0].cpu().numpy(), gt, rtol=1e-4, atol=2e-5)


def test_view_weights_vs_reference_vectors(dev, golden):
    from contexture_nerf_amd import view_weights as vw
    fi = torch.tensor(golden['vw_face_idx'], device=dev)
    fn = torch.tensor(golden['vw_face_normals'], device=dev)
    masks = vw.compare_face_normals_between_views(None, fn, fi)
    assert masks.dtype == torch.bool and masks.shape == fi.shape
    assert np.array_equal(masks.cpu().numpy(), golden['vw_masks'])          # the reference's own output
    fvm = vw.create_face_view_map(fi)
    assert np.array_equal(fvm.cpu().numpy(), golden['vw_face_view_map'])
    toy = torch.tensor([[[[0, -1], [1, 1]]], [[[1, 0], [-1, 2]]]], device=dev)
    assert np.array_equal(vw.create_face_view_map(toy).cpu().numpy(), golden['vw_toy_map'])


def test_view_weights_full_size(dev, meshes):
    """B=7 @ 600^2 on nascar: bit-exact masks vs oracle, plus size-independent properties."""
    from contexture_nerf_amd import kal, view_weights as vw
    verts, f, cam, proj = _scene(meshes, 'nascar', 7)
    g_cam, g_img, g_fn = kal.render.mesh.prepare_vertices(torch.tensor(verts, device=dev), torch.tensor(f, device=dev),
                                                          torch.tensor(proj), camera_transform=torch.tensor(cam, device=dev))
    _, _, idx, _ = kal.render.mesh.rasterize_fused(600, 600, g_cam, g_img, torch.rand(1, f.shape[0], 3, 2, device=dev))
    fn = g_fn.permute(0, 2, 1).contiguous()
    masks = vw.view_weight_masks(idx[:, None], fn)
    mz, om = og.view_weights(idx.cpu().numpy(), g_fn[..., 2].cpu().numpy())
    assert np.array_equal(masks[:, 0].cpu().numpy(), om)
    m = masks[:, 0]
    assert bool(m[idx < 0].all())                                          # background stays True
    # every visible face is owned by at least one view; shard-and-reduce (2 "ranks") gives the same masks
    a, fnz_a = vw.local_max_z(idx[:4, None], fn[:4])
    b, fnz_b = vw.local_max_z(idx[4:, None], fn[4:])
    red = torch.maximum(a, b)
    assert torch.equal(red[torch.isfinite(red)], torch.tensor(mz, device=dev)[torch.isfinite(red)])
    m2 = torch.cat([vw.masks_from_max_z(idx[:4, None], fnz_a, red), vw.masks_from_max_z(idx[4:, None], fnz_b, red)])
    assert torch.equal(m2, masks)
    fvm = vw.create_face_view_map(idx[:, None])
    assert fvm.shape[0] == int((idx >= 0).sum())
    assert bool((fvm[1:, 1] >= fvm[:-1, 1]).all())                          # view-major order


def test_embed_and_texture_field(dev, golden):
    from contexture_nerf_amd import run_nerf_helpers as rnh
    embed, odim = rnh.get_embedder(10)
    assert odim == 42
    x = torch.tensor(golden['embed_x'], device=dev)
    e = embed(x)
    np.testing.assert_allclose(e.cpu().numpy(), golden['embed_y'], rtol=0, atol=3e-6)    # sin/cos of args up to 512
    # small net with the reference's stored weights
    small = rnh.NeRF2D(D=8, W=64, input_ch=42, output_ch=3, skips=[4])
    sd = {k[len('small_'):]: torch.tensor(golden[k]) for k in golden.files if k.startswith('small_') and k != 'small_y'}
    small.load_state_dict(sd)
    small.to(dev)
    y = small(torch.tensor(golden['embed_y'], device=dev))
    np.testing.assert_allclose(y.detach().cpu().numpy(), golden['small_y'], rtol=1e-4, atol=2e-5)
    y2 = small.forward_uv(x)                                                # fused embed
    np.testing.assert_allclose(y2.detach().cpu().numpy(), golden['small_y'], rtol=1e-4, atol=5e-5)
    # full-size net: same seed => same init as the reference (init order parity) => same outputs
    torch.manual_seed(1234)
    net = rnh.NeRF2D(D=8, W=256, input_ch=42, output_ch=3, skips=[4])
    assert sum(p.numel() for p in net.parameters()) == 483075
    net.to(dev)
    y = net(torch.tensor(golden['embed_y'], device=dev))
    np.testing.assert_allclose(y.detach().cpu().numpy(), golden['nerf2d_seed1234_y'], rtol=1e-4, atol=2e-5)
    # texture_map(res) == explicit uv grid path == oracle
    res = 40
    tex, raw = net.texture_map(res)
    uvg = onerf.uv_grid(res)
    assert np.array_equal(uvg, torch.stack(torch.meshgrid(torch.linspace(0, 1, res), torch.linspace(0, 1, res),
                                                          indexing='xy'), -1).reshape(-1, 2).numpy())
    # the in-kernel grid follows the DEVICE linspace formula (start + i*step | end - (n-1-i)*step), which differs
    # from the vectorised CPU linspace by <= 1 ulp at a few nodes; at frequency 2^9 that is <= 3e-5 in a sin/cos
    # argument, hence the 5e-4 absolute tolerance on the raw MLP output below.
    uvd = torch.stack(torch.meshgrid(torch.linspace(0, 1, res, device=dev), torch.linspace(0, 1, res, device=dev),
                                     indexing='xy'), -1).reshape(-1, 2)
    raw_explicit = net.forward_uv(uvd)
    np.testing.assert_allclose(raw.detach().cpu().numpy(), raw_explicit.detach().cpu().numpy(), rtol=1e-4, atol=5e-4)
    ws = [l.weight.detach().cpu().numpy() for l in net.pts_linears]
    bs = [l.bias.detach().cpu().numpy() for l in net.pts_linears]
    o = onerf.nerf2d_forward(onerf.embed(uvg), ws, bs, net.output_linear.weight.detach().cpu().numpy(),
                             net.output_linear.bias.detach().cpu().numpy(), dtype=np.float64)
    np.testing.assert_allclose(raw.detach().cpu().numpy(), o, rtol=1e-3, atol=5e-4)
    np.testing.assert_allclose(tex.detach().cpu().numpy(), onerf.texture_from_mlp(o, res), rtol=0, atol=3e-4)
    assert tex.shape == (1, 3, res, res)


def _field_grads(net):
    return [l.weight.grad for l in net.pts_linears] + [net.output_linear.weight.grad], \
           [l.bias.grad for l in net.pts_linears] + [net.output_linear.bias.grad]


def _oracle_field_grads(net, e, grad_raw=None, grad_tex=None):
    ws = [l.weight.detach().cpu().numpy() for l in net.pts_linears]
    bs = [l.bias.detach().cpu().numpy() for l in net.pts_linears]
    return onerf.nerf2d_backward(e, ws, bs, net.output_linear.weight.detach().cpu().numpy(),
                                 net.output_linear.bias.detach().cpu().numpy(), grad_raw=grad_raw, grad_tex=grad_tex)


def _close(a, b, rel, what):
    a = a.detach().cpu().numpy().astype(np.float64)
    err = np.abs(a - b).max()
    scale = np.abs(b).max() + 1e-30
    assert err <= rel * scale, f"{what}: max abs err {err:.3e} vs scale {scale:.3e}"


def test_texture_field_backward_golden(dev, golden):
    """ctx_uvmlp_bwd vs the REFERENCE's autograd (tests/golden/make_golden.py).  fp32 sums in a different order than
    torch's CPU GEMM: 2e-5 of each gradient tensor's largest entry."""
    from contexture_nerf_amd import run_nerf_helpers as rnh
    e = torch.tensor(golden['embed_y'], device=dev)
    # small net, stored reference weights, tanh head
    small = rnh.NeRF2D(D=8, W=64, input_ch=42, output_ch=3, skips=[4])
    sd = {k[len('small_'):]: torch.tensor(golden[k]) for k in golden.files if k.startswith('small_') and k != 'small_y'}
    small.load_state_dict(sd)
    small.to(dev)
    y = small(e)
    assert y.requires_grad
    (((torch.tanh(y) + 1) / 2) * torch.linspace(-1, 1, y.numel(), device=dev).reshape(y.shape)).sum().backward()
    for k, v in small.named_parameters():
        _close(v.grad, golden['smallgrad_' + k], 2e-5, 'small ' + k)
    # full-size net, seeded init (init-order parity with the reference), linear loss on the raw output
    torch.manual_seed(1234)
    net = rnh.NeRF2D(D=8, W=256, input_ch=42, output_ch=3, skips=[4]).to(dev)
    y = net(e)
    (y * torch.linspace(-1, 1, y.numel(), device=dev).reshape(y.shape)).sum().backward()
    _close(net.pts_linears[0].weight.grad, golden['nerf2d_seed1234_gw0'], 2e-5, 'gw0')
    _close(net.pts_linears[5].weight.grad, golden['nerf2d_seed1234_gw5'], 2e-5, 'gw5')
    _close(net.pts_linears[7].weight.grad, golden['nerf2d_seed1234_gw7'], 2e-5, 'gw7')
    _close(net.output_linear.weight.grad, golden['nerf2d_seed1234_gw_out'], 2e-5, 'gw_out')
    _close(net.output_linear.bias.grad, golden['nerf2d_seed1234_gb_out'], 2e-5, 'gb_out')
    for i in range(8):
        _close(net.pts_linears[i].bias.grad, golden[f'nerf2d_seed1234_gb{i}'], 2e-5, f'gb{i}')


def _field_bwd_abi(net, uv, c_raw, res=0, c_tex=None):
    """training forward + backward straight through the C-ABI -> (gws, gbs, saved activations [D,N,W]).
    uv None: the res x res atlas grid; c_tex [3,N]: gradient wrt the (tanh+1)/2 atlas."""
    import ctypes as C
    from contexture_nerf_amd import _lib as L
    lib = L.load()
    dev = c_raw.device
    N, D, W = c_raw.shape[0], net.D, net.W
    blob = net.packed()
    raw = torch.empty(N, net.output_ch, device=dev)
    saved = torch.zeros(lib.ctx_uvmlp_saved_bytes(N, D, W, net.input_ch) // 4, device=dev)
    L.check(lib.ctx_uvmlp_fwd_save(L.ptr(uv), None, N, res, L.ptr(blob), D, W, net.dims, net.multires, net.output_ch, 4, L.ptr(raw), None,
                                   L.ptr(saved), L.stream()))
    ws = torch.empty(lib.ctx_uvmlp_bwd_ws_bytes(N, D, W), dtype=torch.uint8, device=dev)
    layers = list(net.pts_linears) + [net.output_linear]
    gws = [torch.empty_like(l.weight) for l in layers]
    gbs = [torch.empty_like(l.bias) for l in layers]
    gwp = (C.c_void_p * (D + 1))(*[L.ptr(t).value for t in gws])
    gbp = (C.c_void_p * (D + 1))(*[L.ptr(t).value for t in gbs])
    L.check(lib.ctx_uvmlp_bwd(L.ptr(c_raw), L.ptr(c_tex), L.ptr(raw), N, L.ptr(blob), D, W, net.dims, net.multires, net.output_ch, 4,
                              L.ptr(saved), L.ptr(ws), gwp, gbp, L.stream()))
    ep = 48 if net.input_ch <= 48 else 64
    return gws, gbs, saved[N * ep:N * ep + D * N * W].reshape(D, N, W)     # (the ReLU bit masks follow)


def test_texture_field_split_fp16_vs_exact_f32(dev, golden):
    """The default forward of the 2-D texture field (k_uvmlp_fwd16: fp16 hi + lo split operands, three MFMA passes, fp32 accumulate)
    against the exact-f32 kernel behind CTX_UVMLP_EXACT_F32=1 and against the REFERENCE's stored outputs at the f32 path's own
    tolerance; ragged N (not a multiple of the 128-texel tile), the training forward's saved tensors, and the backward through
    either forward."""
    import os
    from contexture_nerf_amd import run_nerf_helpers as rnh
    torch.manual_seed(1234)
    net = rnh.NeRF2D(D=8, W=256, input_ch=42, output_ch=3, skips=[4]).to(dev)
    e = torch.tensor(golden['embed_y'], device=dev)
    y_fast = net(e)
    np.testing.assert_allclose(y_fast.detach().cpu().numpy(), golden['nerf2d_seed1234_y'], rtol=1e-4, atol=2e-5)     # the reference's own forward
    os.environ["CTX_UVMLP_EXACT_F32"] = "1"
    try:
        y_exact = net(e)
    finally:
        os.environ.pop("CTX_UVMLP_EXACT_F32")
    d = float((y_fast - y_exact).abs().max() / y_exact.abs().max())
    assert d < 2e-6, d                                              # 22-bit split operands: fp32-grade
    assert not torch.equal(y_fast, y_exact)                         # ... and really another kernel
    for N in (1, 127, 129, 1000, 66000):
        g = torch.Generator().manual_seed(N)
        uv = torch.rand(N, 2, generator=g).to(dev)
        with torch.no_grad():
            a = net.forward_uv(uv) if hasattr(net, 'forward_uv') else net(rnh.get_embedder(10)[0](uv))
            os.environ["CTX_UVMLP_EXACT_F32"] = "1"
            try:
                b = net.forward_uv(uv) if hasattr(net, 'forward_uv') else net(rnh.get_embedder(10)[0](uv))
            finally:
                os.environ.pop("CTX_UVMLP_EXACT_F32")
        assert a.shape == b.shape and float((a - b).abs().max()) < 3e-6 * max(1.0, float(b.abs().max())), N
    # training forward + backward: parameter gradients through the fast forward's saved tensors == through the exact one's
    grads = []
    for exact in (False, True):
        if exact:
            os.environ["CTX_UVMLP_EXACT_F32"] = "1"
        try:
            net.zero_grad(set_to_none=True)
            tex, raw = net.texture_map(64)
            (tex * torch.linspace(-1, 1, tex.numel(), device=dev).reshape(tex.shape)).sum().backward()
            grads.append([p.grad.clone() for p in net.parameters()])
        finally:
            os.environ.pop("CTX_UVMLP_EXACT_F32", None)
    # not bit-for-bit and not 1e-6 either: a pre-activation within ~1e-7 of zero takes the other side of its ReLU in the other kernel,
    # and such a flip moves the gradients it feeds by a finite step (measured: one or two flips among 8.4 M activations, 1e-4 relative)
    for ga, gb in zip(*grads):
        assert float((ga - gb).abs().max()) <= 1e-3 * max(float(gb.abs().max()), 1e-6)


@pytest.mark.parametrize("W,N", [(64, 1), (128, 517), (256, 4133), (256, 64 * 300), (256, 64 * 700 + 5)])
def test_texture_field_backward_vs_oracle(dev, W, N):
    """ragged texel counts (tile tails, fewer texel ranges than workgroups, more tiles than persistent workgroups), the
    fused-uv seam, against the float64 oracle.  The ReLU derivative pattern is taken from the device's saved activations
    (checked against the oracle's pre-activations: they may only differ where |pre-activation| < 1e-5), because a unit
    that rounds to the other side of 0 moves a gradient by a whole term (~1/sqrt(N) relative)."""
    from contexture_nerf_amd import run_nerf_helpers as rnh
    torch.manual_seed(5 + W)
    net = rnh.NeRF2D(D=8, W=W, input_ch=42, output_ch=3, skips=[4]).to(dev)
    g = torch.Generator().manual_seed(N)
    uv = torch.rand(N, 2, generator=g)
    c_raw = torch.randn(N, 3, generator=g)
    hw, hb, acts = _field_bwd_abi(net, uv.to(dev), c_raw.to(dev))
    acts = acts.cpu().numpy()
    masks = [acts[i] > 0 for i in range(8)]
    ws = [l.weight.detach().cpu().numpy() for l in net.pts_linears]
    bs = [l.bias.detach().cpu().numpy() for l in net.pts_linears]
    gws, gbs, pre = onerf.nerf2d_backward(onerf.embed(uv.numpy()), ws, bs, net.output_linear.weight.detach().cpu().numpy(),
                                          net.output_linear.bias.detach().cpu().numpy(), grad_raw=c_raw.numpy(), masks=masks,
                                          return_pre=True)
    for i in range(8):
        assert np.abs(acts[i] - np.maximum(pre[i], 0)).max() <= 2e-5 * max(1.0, np.abs(pre[i]).max()), f'saved activations {i}'
        flipped = masks[i] != (pre[i] > 0)
        assert np.all(np.abs(pre[i][flipped]) < 1e-5), f'ReLU pattern of layer {i}'
    for i in range(9):
        _close(hw[i], gws[i], 3e-5, f'W={W} N={N} gw{i}')
        _close(hb[i], gbs[i], 3e-5, f'W={W} N={N} gb{i}')
    # the autograd seam gives the same bits (fixed-order partial sums => deterministic)
    raw = net.forward_uv(uv.to(dev))
    (raw * c_raw.to(dev)).sum().backward()
    aw, ab = _field_grads(net)
    assert all(torch.equal(a, b) for a, b in zip(hw, aw)) and all(torch.equal(a, b) for a, b in zip(hb, ab))


def test_texture_map_backward_and_fit(dev):
    """texture_map(res) -> atlas gradient -> parameters (the texture side of the SDS loop, trainer.py:644-907) and a few
    Adam steps towards a target atlas."""
    from contexture_nerf_amd import run_nerf_helpers as rnh
    torch.manual_seed(3)
    net = rnh.NeRF2D(D=8, W=256, input_ch=42, output_ch=3, skips=[4]).to(dev)
    res = 48
    g = torch.Generator().manual_seed(9)
    c_tex = torch.randn(1, 3, res, res, generator=g)
    c_raw = torch.randn(res * res, 3, generator=g) * 0.1
    tex, raw = net.texture_map(res)
    ((tex * c_tex.to(dev)).sum() + (raw * c_raw.to(dev)).sum()).backward()
    hw, hb = _field_grads(net)
    # the same call through the C-ABI gives the same bits, and its saved activations give the ReLU pattern for the oracle
    dw, db, acts = _field_bwd_abi(net, None, c_raw.to(dev), res=res, c_tex=c_tex[0].reshape(3, -1).contiguous().to(dev))
    assert all(torch.equal(a, b) for a, b in zip(hw, dw)) and all(torch.equal(a, b) for a, b in zip(hb, db))
    acts = acts.cpu().numpy()
    gt = c_tex[0].permute(1, 2, 0).reshape(-1, 3).numpy()                   # [N,3] in texel order (row i <-> v, col j <-> u)
    ws = [l.weight.detach().cpu().numpy() for l in net.pts_linears]
    bs = [l.bias.detach().cpu().numpy() for l in net.pts_linears]
    gws, gbs = onerf.nerf2d_backward(onerf.embed(onerf.uv_grid(res)), ws, bs, net.output_linear.weight.detach().cpu().numpy(),
                                     net.output_linear.bias.detach().cpu().numpy(), grad_raw=c_raw.numpy(), grad_tex=gt,
                                     masks=[acts[i] > 0 for i in range(8)])
    for i in range(9):
        _close(hw[i], gws[i], 2e-4, f'gw{i}')        # the device linspace differs from the CPU one by 1 ulp at a few nodes (see above)
        _close(hb[i], gbs[i], 2e-4, f'gb{i}')
    target = torch.rand(1, 3, 1, 1, device=dev).expand(1, 3, res, res)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        tex, _ = net.texture_map(res)
        loss = ((tex - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.5 * losses[0], losses
    with torch.no_grad():
        t2, _ = net.texture_map(res)                 # inference path sees the updated weights (packed blob re-made)
    assert abs(((t2 - target) ** 2).mean().item() - losses[-1]) < losses[0]


def test_volume_render_3d_field(dev, golden):
    """BASELINE configs[4] in small: get_rays -> stratified samples -> 3-D Fourier embed (63) -> NeRF2D(input_ch 63, output_ch 4)
    -> raw2outputs, against the oracle (numpy embed / MLP in float64, C compositing).  Parity unpinned vs a reference run
    (the reference has no ray-march body); the pieces are pinned separately (embed, NeRF2D, get_rays: golden)."""
    from contexture_nerf_amd import run_nerf_helpers as rnh
    H = W_ = 24
    f = 12.0 / np.tan(np.pi / 6)
    K = np.array([[f, 0, W_ / 2], [0, f, H / 2], [0, 0, 1]], np.float32)
    c2w = torch.tensor(golden['rays_c2w'], device=dev)
    ro, rd = rnh.get_rays(H, W_, K, c2w)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    torch.manual_seed(21)
    field = rnh.NeRF2D(D=8, W=256, input_ch=63, output_ch=4, skips=[4]).to(dev)
    assert (field.dims, field.multires) == (3, 10)
    S = 40
    with torch.no_grad():
        rgb, disp, acc, wts, depth = rnh.render_rays(field, ro, rd, near=0.5, far=2.5, N_samples=S)
        # the fused embedding == ctx_embed_fwd followed by the embedded-input seam
        t = torch.linspace(0., 1., S, device=dev)
        z = (0.5 * (1 - t) + 2.5 * t).expand(ro.shape[0], S).contiguous()
        pts = ro[:, None, :] + rd[:, None, :] * z[:, :, None]
        embed3 = rnh.Embedder(3, 10)
        raw_seam = field(embed3.embed(pts.reshape(-1, 3)))
        raw_fused = field.forward_pts(pts).reshape(-1, 4)
    np.testing.assert_allclose(raw_fused.cpu().numpy(), raw_seam.cpu().numpy(), rtol=1e-4, atol=2e-4)
    ws = [l.weight.detach().cpu().numpy() for l in field.pts_linears]
    bs = [l.bias.detach().cpu().numpy() for l in field.pts_linears]
    e = onerf.embed(pts.reshape(-1, 3).cpu().numpy())
    assert e.shape[1] == 63
    raw_o = onerf.nerf2d_forward(e, ws, bs, field.output_linear.weight.detach().cpu().numpy(),
                                 field.output_linear.bias.detach().cpu().numpy(), dtype=np.float64)
    # sin/cos arguments reach |x| * 2^9 ~ 1e3: the fp32 embedding differs from the float64 one by ~1e-4 there
    np.testing.assert_allclose(raw_fused.cpu().numpy(), raw_o, rtol=2e-3, atol=2e-3)
    o = og.raw2outputs(raw_fused.reshape(-1, S, 4).cpu().numpy(), z.cpu().numpy(), rd.cpu().numpy(), white_bkgd=False)
    for a, b in zip((rgb, disp, acc, wts, depth), o):
        np.testing.assert_allclose(a.cpu().numpy(), b, rtol=2e-4, atol=2e-6)
    assert rgb.shape == (H * W_, 3) and torch.isfinite(rgb).all()


def test_field_3d_backward(dev):
    """the 3-D field (padded embedding 64, 4 outputs) through the same backward, ragged N."""
    from contexture_nerf_amd import run_nerf_helpers as rnh
    torch.manual_seed(33)
    net = rnh.NeRF2D(D=8, W=256, input_ch=63, output_ch=4, skips=[4]).to(dev)
    N = 64 * 37 + 11
    g = torch.Generator().manual_seed(4)
    pts = torch.rand(N, 3, generator=g) * 2 - 1
    c_raw = torch.randn(N, 4, generator=g)
    hw, hb, acts = _field_bwd_abi(net, pts.to(dev), c_raw.to(dev))
    acts = acts.cpu().numpy()
    ws = [l.weight.detach().cpu().numpy() for l in net.pts_linears]
    bs = [l.bias.detach().cpu().numpy() for l in net.pts_linears]
    gws, gbs = onerf.nerf2d_backward(onerf.embed(pts.numpy()), ws, bs, net.output_linear.weight.detach().cpu().numpy(),
                                     net.output_linear.bias.detach().cpu().numpy(), grad_raw=c_raw.numpy(),
                                     masks=[acts[i] > 0 for i in range(8)])
    for i in range(9):
        _close(hw[i], gws[i], 1e-4, f'3-D field gw{i}')
        _close(hb[i], gbs[i], 1e-4, f'3-D field gb{i}')
    raw = net.forward_pts(pts.to(dev))
    (raw * c_raw.to(dev)).sum().backward()
    aw, ab = _field_grads(net)
    assert all(torch.equal(a, b) for a, b in zip(hw, aw)) and all(torch.equal(a, b) for a, b in zip(hb, ab))


def test_texture_field_full_size_properties(dev):
    """The 1024^2 atlas of the reference (1 048 576 texels) through size-independent properties: the fused grid path
    equals the explicit-uv path and the float64 oracle on a random subset of texels; the backward is linear in the upstream
    gradient, and the gradients of two disjoint texel sets add up to the gradient of their union (every texel range of the
    split-K weight-gradient GEMMs and every tile of the dZ chain contributes exactly once)."""
    from contexture_nerf_amd import run_nerf_helpers as rnh
    torch.manual_seed(17)
    net = rnh.NeRF2D(D=8, W=256, input_ch=42, output_ch=3, skips=[4]).to(dev)
    res = 1024
    with torch.no_grad():
        tex, raw = net.texture_map(res)
    assert tex.shape == (1, 3, res, res) and torch.isfinite(raw).all()
    g = torch.Generator().manual_seed(3)
    idx = torch.randint(0, res * res, (4096,), generator=g)
    lin = torch.linspace(0, 1, res, device=dev)
    uv = torch.stack([lin[idx.to(dev) % res], lin[idx.to(dev) // res]], -1)          # (u, v) = (column, row) of the 'xy' meshgrid
    with torch.no_grad():
        sub = net.forward_uv(uv)
    np.testing.assert_allclose(raw[idx.to(dev)].cpu().numpy(), sub.cpu().numpy(), rtol=1e-4, atol=5e-4)
    ws = [l.weight.detach().cpu().numpy() for l in net.pts_linears]
    bs = [l.bias.detach().cpu().numpy() for l in net.pts_linears]
    o = onerf.nerf2d_forward(onerf.embed(uv.cpu().numpy()), ws, bs, net.output_linear.weight.detach().cpu().numpy(),
                             net.output_linear.bias.detach().cpu().numpy(), dtype=np.float64)
    np.testing.assert_allclose(sub.cpu().numpy(), o, rtol=1e-3, atol=5e-4)
    # backward properties at full size
    gt = torch.randn(1, 3, res, res, generator=g).to(dev)
    half = (torch.arange(res * res, device=dev) % 7 < 3).reshape(1, 1, res, res).float()          # an irregular split of the texels

    def grads(gtex):
        net.zero_grad(set_to_none=True)
        t, _ = net.texture_map(res)
        t.backward(gtex)
        return [p.grad.clone() for p in net.parameters()]
    ga, gb, gab = grads(gt * half), grads(gt * (1 - half)), grads(gt)
    g2 = grads(2.5 * gt)
    for a, b, ab, s2 in zip(ga, gb, gab, g2):
        scale = ab.abs().max().item() + 1e-20
        assert ((a + b) - ab).abs().max().item() <= 2e-5 * scale           # fp32 partial sums in a different grouping
        assert (s2 - 2.5 * ab).abs().max().item() <= 2e-5 * scale
    assert all(torch.equal(x, y) for x, y in zip(gab, grads(gt)))            # deterministic at full size


def test_rays_and_composite(dev, golden):
    from contexture_nerf_amd import run_nerf_helpers as rnh
    ro, rd = rnh.get_rays(6, 8, golden['rays_K'], torch.tensor(golden['rays_c2w'], device=dev))
    np.testing.assert_allclose(rd.cpu().numpy(), golden['rays_d'], rtol=1e-6, atol=1e-6)
    assert np.array_equal(ro.cpu().numpy(), golden['rays_o'])
    no, nd = rnh.ndc_rays(6, 8, 5.0, 1.0, ro, rd)
    np.testing.assert_allclose(no.cpu().numpy(), golden['ndc_o'], rtol=1e-5, atol=1e-5)
    s = rnh.sample_pdf(torch.tensor(golden['pdf_bins'], device=dev), torch.tensor(golden['pdf_w'], device=dev), 24, det=True)
    np.testing.assert_allclose(s.cpu().numpy(), golden['pdf_det'], rtol=1e-5, atol=1e-5)
    s = rnh.sample_pdf(torch.tensor(golden['pdf_bins'], device=dev), torch.tensor(golden['pdf_w'], device=dev), 24, pytest=True)
    np.testing.assert_allclose(s.cpu().numpy(), golden['pdf_pytest'], rtol=1e-5, atol=1e-5)
    g = torch.Generator().manual_seed(2)
    for (R, S) in [(37, 128), (5, 33), (3, 64), (2, 200)]:
        raw = torch.randn(R, S, 4, generator=g) * 2
        z = torch.sort(torch.rand(R, S, generator=g) * 4 + 2, -1).values
        d = torch.randn(R, 3, generator=g)
        o = og.raw2outputs(raw.numpy(), z.numpy(), d.numpy(), white_bkgd=True)
        out = rnh.raw2outputs(raw.to(dev), z.to(dev), d.to(dev), white_bkgd=True)
        # tolerance: wave-parallel prefix product / sums vs sequential float32 oracle, expf ulp differences
        for a, b in zip(out, o):
            np.testing.assert_allclose(a.cpu().numpy(), b, rtol=2e-4, atol=2e-6)


def test_volume_render_full_size_vs_oracle(dev):
    """BASELINE configs[4] at FULL size: 512^2 rays x 128 samples through the 3-D field NeRF2D(D 8, W 256, input_ch 63, output_ch 4)
    and raw2outputs on the HIP path (33.5 M field evaluations, one launch chain), checked (a) against the float64 numpy oracle
    field + the C compositing oracle on a subset of image rows (top, centre, bottom: 3 x 512 rays x 128 samples), (b) through the
    size-independent property that a row-tile render equals the same rows of the whole-image render bit for bit (rays are
    independent: the multi-GPU row sharding of SURVEY section 8e).  Parity unpinned vs a reference run (no ray-march body upstream)."""
    from contexture_nerf_amd import run_nerf_helpers as rnh, volume_render as vr
    H = W_ = 512
    S = 128
    torch.manual_seed(77)
    field = rnh.NeRF2D(D=8, W=256, input_ch=63, output_ch=4, skips=[4]).to(dev)
    with torch.no_grad():
        field.output_linear.bias[3] = 1.0                       # some density, so that weights / acc are not ~0 everywhere
    c2w = torch.tensor([[1, 0, 0, 0.05], [0, 1, 0, -0.02], [0, 0, 1, 1.5]], dtype=torch.float32, device=dev)
    K = vr.pinhole(H, W_)
    full = vr.render_image(field, H, W_, K, c2w, 0.5, 2.5, S)
    # disp = 1 / max(1e-10, depth / acc) is 0/0 = NaN wherever a ray accumulates nothing (nerf-pytorch's formula): not checked
    assert full['rgb'].shape == (H, W_, 3) and all(torch.isfinite(full[k]).all() for k in ('rgb', 'depth', 'acc'))
    rows = (0, 255, 511)
    ro, rd = rnh.get_rays(H, W_, K, c2w)
    t = torch.linspace(0., 1., S, device=dev)
    ws = [l.weight.detach().cpu().numpy() for l in field.pts_linears]
    bs = [l.bias.detach().cpu().numpy() for l in field.pts_linears]
    wo, bo = field.output_linear.weight.detach().cpu().numpy(), field.output_linear.bias.detach().cpu().numpy()
    for r in rows:
        o, d = ro[r].reshape(-1, 3), rd[r].reshape(-1, 3)
        z = (0.5 * (1 - t) + 2.5 * t).expand(W_, S).contiguous()
        pts = (o[:, None, :] + d[:, None, :] * z[:, :, None]).reshape(-1, 3).cpu().numpy()
        raw = onerf.nerf2d_forward(onerf.embed(pts), ws, bs, wo, bo, dtype=np.float64).reshape(W_, S, 4)
        want = og.raw2outputs(raw.astype(np.float32), z.cpu().numpy(), d.cpu().numpy(), white_bkgd=False)
        tile = vr.render_image(field, H, W_, K, c2w, 0.5, 2.5, S, rows=(r, r + 1))
        for k in ('rgb', 'depth', 'acc'):
            assert torch.equal(tile[k][0], full[k][r]), (k, r)                     # (b)
        # (a): the fp32 embedding of |x| 2^9 ~ 1e3 arguments differs from the float64 one by ~1e-4 (as at the small size);
        # after compositing over 128 samples the image-level quantities agree to ~1e-3
        np.testing.assert_allclose(full['rgb'][r].cpu().numpy(), want[0], rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(full['acc'][r].cpu().numpy(), want[2], rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(full['depth'][r].cpu().numpy(), want[4], rtol=2e-3, atol=4e-3)
    tiles = [vr.render_image(field, H, W_, K, c2w, 0.5, 2.5, S, rows=vr.shard_rows(H, k, 8)) for k in (0, 3, 7)]
    for k, tl in zip((0, 3, 7), tiles):
        r0, r1 = vr.shard_rows(H, k, 8)
        assert torch.equal(tl['rgb'], full['rgb'][r0:r1]) and torch.equal(tl['depth'], full['depth'][r0:r1])


def test_project_back_scatter_vs_oracle(dev, meshes):
    """The UV back-projection scatter of ConTEXTure.project_back_scatter (C = 3 colours + weight, face-index mask) == the C
    oracle's texture_mapping backward (`orc_texture_mapping_bwd`, the restatement of grid_sample's backward) on the same pixels:
    a real raster of a bundled mesh at 600^2 into a 256^2 atlas, with a half-empty weight mask."""
    from contexture_nerf_amd import kal
    from contexture_nerf_amd.trainer import ConTEXTure
    v = og.normalize_mesh(meshes["spot_triangulated_v"], 0.6, 0.25); f = meshes["spot_triangulated_f"].astype(np.int64)
    vt, ft = meshes["spot_triangulated_vt"], meshes["spot_triangulated_ft"].astype(np.int64)
    B, Hh, Ww, T = 2, 600, 600, 256
    cam = og.get_camera_from_multiple_view(np.float32([1.0471976, 1.9198622]), np.float32([0.5, 3.6]), np.float32([1.5, 1.5]), 0.25)
    proj = og.generate_perspective_projection(np.pi / 3)
    g_cam, g_img, g_fn = kal.render.mesh.prepare_vertices(torch.tensor(np.repeat(v[None], B, 0), device=dev), torch.tensor(f, device=dev),
                                                          torch.tensor(proj), camera_transform=torch.tensor(cam, device=dev))
    uv_attr = torch.tensor(vt[ft][None].repeat(B, 0), device=dev)
    depth, uv, face_idx, normals = kal.render.mesh.rasterize_fused(Hh, Ww, g_cam, g_img, uv_attr, g_fn)
    rng = np.random.default_rng(3)
    rgb = torch.tensor(rng.random((B, 3, Hh, Ww), dtype=np.float32), device=dev)
    wmask = torch.tensor(rng.random((B, 1, Hh, Ww)) > 0.5, device=dev)
    tr = ConTEXTure.__new__(ConTEXTure)
    tr.cfg = type('C', (), {'guide': type('G', (), {'texture_resolution': T})()})()
    contrib = tr.project_back_scatter(dict(uv_features=uv, face_idx=face_idx), rgb, wmask)
    assert contrib.shape == (4, T, T) and contrib.dtype == torch.int64
    w = wmask.float().permute(0, 2, 3, 1)
    go = torch.cat([rgb.permute(0, 2, 3, 1) * w, w], -1).cpu().numpy()
    # BIT-EXACT against the integer oracle (2^-32 fixed point; integer sums are order-free)
    want_i = og.uv_scatter_fixed(go, uv.cpu().numpy(), face_idx.cpu().numpy(), T, kal.SCATTER_FRAC_BITS)
    assert np.array_equal(contrib.cpu().numpy(), want_i), f"{(contrib.cpu().numpy() != want_i).sum()} texel sums differ"
    # a second view added into the same accumulator == the oracle fed both; conversions agree bit for bit
    tr.project_back_scatter(dict(uv_features=uv[:1].contiguous(), face_idx=face_idx[:1].contiguous()), rgb[:1], wmask[:1], acc=contrib)
    og.uv_scatter_fixed(go[:1], uv[:1].cpu().numpy(), face_idx[:1].cpu().numpy(), T, kal.SCATTER_FRAC_BITS, acc=want_i)
    assert np.array_equal(contrib.cpu().numpy(), want_i)
    as_float = kal.fixed_to_float(contrib)
    assert np.array_equal(as_float.cpu().numpy(), (want_i.astype(np.float64) * 2.0 ** -32).astype(np.float32))
    assert torch.equal(as_float, (contrib.to(torch.float64) * 2.0 ** -32).to(torch.float32))         # dist.merge_atlas' conversion
    # and it is the float scatter the oracle's grid_sample backward describes
    go2 = np.concatenate([go, go[:1]], 0) * (np.concatenate([face_idx.cpu().numpy(), face_idx[:1].cpu().numpy()], 0) >= 0)[..., None]
    want = og.texture_mapping_bwd(go2, np.concatenate([uv.cpu().numpy(), uv[:1].cpu().numpy()], 0), T)
    np.testing.assert_allclose(as_float.cpu().numpy(), want, rtol=0, atol=4e-6 * max(np.abs(want).max(), 1.0))
    assert float(as_float[3].sum()) > 1000


def test_uv_back_projection_default_sizes_bit_exact(dev, meshes):
    """The UV back-projection at the reference's default sizes: nascar, the 7 Zero123++ views rastered at 1200^2 (src/configs/train_config.py:11),
    rgb * weight and weight scattered into the 1024^2 atlas (:63) — int64 sums bit-equal to the integer oracle, through the cached tile
    plan, twice (accumulating), and per view == all views at once (what makes the sharded paint independent of the dealing)."""
    from contexture_nerf_amd import kal
    verts, f, cam, proj = _scene(meshes, "nascar", 7)
    g_cam, g_img, g_fn = kal.render.mesh.prepare_vertices(torch.tensor(verts, device=dev), torch.tensor(f, device=dev),
                                                          torch.tensor(proj), camera_transform=torch.tensor(cam, device=dev))
    uva = _uv_attr(meshes, "nascar", f.shape[0])
    H = W = 1200; T = 1024
    _, uv, idx, _ = kal.render.mesh.rasterize_fused(H, W, g_cam, g_img, torch.tensor(uva, device=dev), g_fn)
    gen = torch.Generator(device=dev).manual_seed(0)
    rgb = torch.rand(7, H, W, 3, generator=gen, device=dev)
    w = (torch.rand(7, H, W, 1, generator=gen, device=dev) > 0.3).float()
    vals = torch.cat([rgb * w, w], -1).contiguous()
    acc = torch.zeros(4, T, T, dtype=torch.int64, device=dev)
    kal.scatter_fixed(vals, uv, idx, acc)
    want = og.uv_scatter_fixed(vals.cpu().numpy(), uv.cpu().numpy(), idx.cpu().numpy(), T, kal.SCATTER_FRAC_BITS)
    assert np.array_equal(acc.cpu().numpy(), want), f"{(acc.cpu().numpy() != want).sum()} texel sums differ"
    assert float((want[3] > 0).mean()) > 0.05
    per_view = torch.zeros_like(acc)
    for b in (3, 0, 6, 1, 5, 2, 4):                                  # any order of views
        kal.scatter_fixed(vals[b:b + 1].contiguous(), uv[b:b + 1].contiguous(), idx[b:b + 1].contiguous(), per_view)
    assert torch.equal(per_view, acc)
    kal.scatter_fixed(vals, uv, idx, acc)
    assert torch.equal(acc, 2 * per_view)


@pytest.mark.parametrize("B,H,W,C,T", [(1, 64, 64, 4, 100), (2, 300, 300, 3, 2304), (1, 128, 128, 6, 64)])
def test_uv_scatter_fixed_small_and_planless(dev, B, H, W, C, T):
    """ctx_uv_scatter_fixed on rasters below the binning threshold, on an atlas beyond the plan's LDS histogram (T = 2304 > 2272:
    the plan-less kernel, one int64 atomic per tap) and with more channels than the tile kernel takes: same integer sums."""
    from contexture_nerf_amd import kal
    rng = np.random.default_rng(T + C)
    uv = (rng.random((B, H, W, 2)) * 1.2 - 0.1).astype(np.float32)
    val = rng.random((B, H, W, C)).astype(np.float32)
    fidx = np.where(rng.random((B, H, W)) > 0.3, 2, -1).astype(np.int64)
    assert kal.binned_fits(C, T) == (C <= 4 and T <= 2272)
    acc = torch.zeros(C, T, T, dtype=torch.int64, device=dev)
    kal.scatter_fixed(torch.tensor(val, device=dev), torch.tensor(uv, device=dev), torch.tensor(fidx, device=dev), acc)
    assert np.array_equal(acc.cpu().numpy(), og.uv_scatter_fixed(val, uv, fidx, T, kal.SCATTER_FRAC_BITS))


def test_uv_scatter_binned_tiny_and_huge_gradients(dev):
    """The binned backward picks its fixed-point unit from max|grad_out| of the call: gradients of 1e-8 (a mean-reduced loss, a
    low-noise DreamTime step) and of 1e+6 come out with the same RELATIVE accuracy as O(1) ones; a non-finite gradient poisons the
    output instead of being rounded away; a raster rewritten behind a cached plan is detected."""
    from contexture_nerf_amd import kal
    rng = np.random.default_rng(5)
    B, H, W, C, T = 2, 300, 300, 3, 128
    uv = rng.random((B, H, W, 2)).astype(np.float32)
    go = rng.standard_normal((B, H, W, C)).astype(np.float32)
    go[0, :50] *= 1e-4                                       # a wide dynamic range inside one call
    uvd = torch.tensor(uv, device=dev)
    ref = og.texture_mapping_bwd(go, uv, T).astype(np.float64)
    for sc in (1.0, 1e-8, 1e-20, 1e6):
        g = kal.scatter_add_texture(torch.tensor(go * np.float32(sc), device=dev), uvd, None, torch.zeros(C, T, T, device=dev), binned=True)
        got = g.cpu().numpy().astype(np.float64) / sc
        err = np.abs(got - ref).max() / np.abs(ref).max()
        assert err < 2e-6, (sc, err)
        rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30)
        assert np.median(rel) < 3e-7, (sc, np.median(rel))
    bad = go.copy(); bad[1, 7, 9, 1] = np.inf
    g = kal.scatter_add_texture(torch.tensor(bad, device=dev), uvd, None, torch.zeros(C, T, T, device=dev), binned=True)
    assert bool(torch.isnan(g).all())
    # stale plan: cache a plan for uvd, then overwrite the raster in place through .data (no version bump)
    kal.clear_scatter_plans()
    god = torch.tensor(go, device=dev)
    g1 = kal.scatter_add_texture(god, uvd, None, torch.zeros(C, T, T, device=dev), binned=True, reuse=True)
    assert len(kal._PLANS) == 1 and not bool(torch.isnan(g1).any())
    uvd.data.copy_(torch.tensor(rng.random((B, H, W, 2)).astype(np.float32), device=dev))
    g2 = kal.scatter_add_texture(god, uvd, None, torch.zeros(C, T, T, device=dev), binned=True, reuse=True)
    assert bool(torch.isnan(g2).all()), "a rewritten raster must not be scattered along the old plan"
    plan = next(iter(kal._PLANS.values()))[0]
    from contexture_nerf_amd import _lib as L
    assert L.load().ctx_texmap_plan_stale(L.ptr(plan), L.stream()) == 1
    kal.clear_scatter_plans()
    assert len(kal._PLANS) == 0


@pytest.mark.parametrize("B,H,W,C,T", [(2, 300, 300, 3, 128), (1, 257, 129, 4, 100), (3, 256, 256, 1, 64)])
def test_uv_scatter_binned_vs_oracle_and_atomics(dev, B, H, W, C, T):
    """uvscatter.hip (pixels binned by atlas tile, LDS-resident int64 fixed-point tiles, one store per texel) against the C oracle's
    grid_sample backward and the float-atomics kernel: random uv incl. values outside [0,1] (border clamp), a masked half, atlas
    sizes that are not multiples of the tile; bit-identical results on repeated calls (integer sums) and with a reused plan."""
    from contexture_nerf_amd import kal
    rng = np.random.default_rng(B * 1000 + T)
    uv = (rng.random((B, H, W, 2)) * 1.2 - 0.1).astype(np.float32)
    go = rng.standard_normal((B, H, W, C)).astype(np.float32)
    fidx = np.where(rng.random((B, H, W)) > 0.4, 5, -1).astype(np.int64)
    want = og.texture_mapping_bwd(go * (fidx >= 0)[..., None], uv, T)
    uvd, god, fd = torch.tensor(uv, device=dev), torch.tensor(go, device=dev), torch.tensor(fidx, device=dev)
    kal.clear_scatter_plans()
    g1 = kal.scatter_add_texture(god, uvd, fd, torch.zeros(C, T, T, device=dev), binned=True, reuse=True)
    g2 = kal.scatter_add_texture(god, uvd, fd, torch.zeros(C, T, T, device=dev), binned=True, reuse=True)       # plan reused
    assert len(kal._PLANS) == 1
    assert torch.equal(g1, kal.scatter_add_texture(god, uvd, fd, torch.zeros(C, T, T, device=dev), binned=True))   # one-shot plan
    assert len(kal._PLANS) == 1
    kal.clear_scatter_plans()
    ga = kal.scatter_add_texture(god, uvd, fd, torch.zeros(C, T, T, device=dev), binned=False)
    assert torch.equal(g1, g2)
    scale = max(1.0, float(np.abs(want).max()))
    np.testing.assert_allclose(g1.cpu().numpy(), want, rtol=0, atol=4e-6 * scale)
    np.testing.assert_allclose(ga.cpu().numpy(), want, rtol=0, atol=2e-5 * scale)
    # accumulates into a caller-filled gradient
    base = torch.full((C, T, T), 0.5, device=dev)
    g3 = kal.scatter_add_texture(god, uvd, fd, base.clone(), binned=True)
    np.testing.assert_allclose((g3 - 0.5).cpu().numpy(), g1.cpu().numpy(), rtol=0, atol=1e-6 * scale)


def test_uv_scatter_binned_heavy_tile_and_autograd(dev):
    """A raster whose pixels crowd into a few atlas tiles (several chunks per tile: the int64 global accumulators and the finish
    pass) and an all-background view; then texture_mapping's autograd takes the binned path on a large raster and agrees with
    torch's grid_sample backward."""
    from contexture_nerf_amd import kal
    rng = np.random.default_rng(11)
    B, H, W, C, T = 2, 512, 512, 3, 256
    uv = (0.40 + 0.08 * rng.random((B, H, W, 2))).astype(np.float32)           # ~524k pixels into a 20 x 20 texel patch
    go = rng.standard_normal((B, H, W, C)).astype(np.float32)
    want = og.texture_mapping_bwd(go, uv, T)
    uvd, god = torch.tensor(uv, device=dev), torch.tensor(go, device=dev)
    g = kal.scatter_add_texture(god, uvd, None, torch.zeros(C, T, T, device=dev), binned=True)
    np.testing.assert_allclose(g.cpu().numpy(), want, rtol=2e-6, atol=2e-4)     # sums of ~1300 terms per texel: the oracle's own rounding
    assert torch.equal(g, kal.scatter_add_texture(god, uvd, None, torch.zeros(C, T, T, device=dev), binned=True))
    none = kal.scatter_add_texture(god, uvd, torch.full((B, H, W), -1, dtype=torch.int64, device=dev), torch.zeros(C, T, T, device=dev), binned=True)
    assert float(none.abs().max()) == 0.0
    tex = torch.rand(1, C, T, T, device=dev, requires_grad=True)
    uv2 = torch.rand(B, H, W, 2, device=dev)
    out = kal.render.mesh.texture_mapping(uv2, tex.expand(B, -1, -1, -1), mode='bilinear')
    cot = torch.randn_like(out)
    (out * cot).sum().backward()
    tex_t = tex.detach().clone().requires_grad_(True)
    grid = torch.stack([uv2[..., 0], 1 - uv2[..., 1]], -1) * 2 - 1
    ref = torch.nn.functional.grid_sample(tex_t.expand(B, -1, -1, -1), grid, mode='bilinear', padding_mode='border', align_corners=False)
    (ref.permute(0, 2, 3, 1) * cot).sum().backward()
    np.testing.assert_allclose(tex.grad.cpu().numpy(), tex_t.grad.cpu().numpy(), rtol=1e-4, atol=2e-4)
