"""CPU: the oracle's restatements against vectors produced by the reference's own functions
(tests/golden/make_golden.py).  Exact where the arithmetic is integer/compare, tight float tolerance otherwise."""
import numpy as np
import torch
from oracle import nerf as onerf, geometry as og


def test_embed_matches_reference(golden, golden_meta):
    y = onerf.embed(golden['embed_x'], 10)
    assert y.shape[1] == golden_meta['embed_out_dim'] == 42
    np.testing.assert_allclose(y, golden['embed_y'], rtol=0, atol=2e-6)
    # layout: [u, v, sin u, sin v, cos u, cos v, sin 2u, ...]
    x = golden['embed_x']
    np.testing.assert_allclose(golden['embed_y'][:, 6], np.sin(2 * x[:, 0]), atol=1e-6)


def test_nerf2d_small_stored_weights(golden):
    ws = [golden[f'small_pts_linears.{i}.weight'] for i in range(8)]
    bs = [golden[f'small_pts_linears.{i}.bias'] for i in range(8)]
    y = onerf.nerf2d_forward(golden['embed_y'], ws, bs, golden['small_output_linear.weight'],
                             golden['small_output_linear.bias'])
    np.testing.assert_allclose(y, golden['small_y'], rtol=1e-4, atol=1e-5)


def test_nerf2d_backward_small_stored_weights(golden):
    """oracle backward vs the reference's autograd (tests/golden/make_golden.py: loss = sum(((tanh(y)+1)/2) * linspace))."""
    ws = [golden[f'small_pts_linears.{i}.weight'] for i in range(8)]
    bs = [golden[f'small_pts_linears.{i}.bias'] for i in range(8)]
    c = np.linspace(-1, 1, 96 * 3, dtype=np.float32).reshape(96, 3)
    gws, gbs = onerf.nerf2d_backward(golden['embed_y'], ws, bs, golden['small_output_linear.weight'],
                                     golden['small_output_linear.bias'], grad_tex=c)
    for i in range(8):
        np.testing.assert_allclose(gws[i], golden[f'smallgrad_pts_linears.{i}.weight'], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(gbs[i], golden[f'smallgrad_pts_linears.{i}.bias'], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(gws[8], golden['smallgrad_output_linear.weight'], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(gbs[8], golden['smallgrad_output_linear.bias'], rtol=2e-4, atol=2e-5)


def test_rays_and_sampling(golden):
    ro, rd = onerf.get_rays(6, 8, golden['rays_K'], golden['rays_c2w'])
    np.testing.assert_allclose(rd, golden['rays_d'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ro, golden['rays_o'], rtol=0, atol=0)
    np.testing.assert_allclose(golden['rays_d_np'], golden['rays_d'], rtol=1e-6, atol=1e-6)
    no, nd = onerf.ndc_rays(6, 8, 5.0, 1.0, golden['rays_o'], golden['rays_d'])
    np.testing.assert_allclose(no, golden['ndc_o'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(nd, golden['ndc_d'], rtol=1e-5, atol=1e-5)
    s = onerf.sample_pdf(golden['pdf_bins'], golden['pdf_w'], 24, det=True)
    np.testing.assert_allclose(s, golden['pdf_det'], rtol=1e-5, atol=1e-5)
    s = onerf.sample_pdf(golden['pdf_bins'], golden['pdf_w'], 24, det=False, pytest=True)
    np.testing.assert_allclose(s, golden['pdf_pytest'], rtol=1e-5, atol=1e-5)
    s = onerf.sample_pdf(golden['pdf_bins'], golden['pdf_w'], 24, det=True, pytest=True)
    np.testing.assert_allclose(s, golden['pdf_det_pytest'], rtol=1e-5, atol=1e-5)


def test_view_weights_bit_exact(golden):
    fi = golden['vw_face_idx']                      # [B,1,H,W]
    fn = golden['vw_face_normals']                  # [B,3,F]
    mz, mask = og.view_weights(fi, fn[:, 2, :])
    assert np.array_equal(mask, golden['vw_masks'])
    # the per-face maxima equal a scatter-max over the reference's own face-view map rows
    fvm = golden['vw_face_view_map']
    z = fn[fvm[:, 1], 2, fvm[:, 0]]
    for f in np.unique(fvm[:, 0]):
        assert mz[f] == z[fvm[:, 0] == f].max()


def test_normalize_depth_and_mesh(golden):
    d = og.normalize_multiple_depth(golden['depth_raw'])
    assert np.array_equal(d, golden['depth_norm'])
    n, a = og.calculate_face_normals(golden['mesh_v'], golden['mesh_f'])
    np.testing.assert_allclose(n, golden['mesh_fn'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(a, golden['mesh_area'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(og.normalize_mesh(golden['mesh_v'], 0.6, 0.25), golden['mesh_v_norm'], rtol=1e-6, atol=1e-6)


def test_texture_mapping_matches_grid_sample():
    """kaolin's texture_mapping is F.grid_sample(tex, (u,1-v)*2-1, align_corners=False, padding 'border')."""
    g = torch.Generator().manual_seed(0)
    uv = torch.rand(2, 9, 11, 2, generator=g) * 1.2 - 0.1           # a few samples outside [0,1] -> border
    tex = torch.rand(1, 3, 16, 16, generator=g)
    grid = torch.stack([uv[..., 0], 1 - uv[..., 1]], -1) * 2 - 1
    for mode in ('bilinear', 'nearest'):
        ref = torch.nn.functional.grid_sample(tex.expand(2, -1, -1, -1), grid, mode=mode, align_corners=False,
                                              padding_mode='border').permute(0, 2, 3, 1)
        out = og.texture_mapping(uv.numpy(), tex.numpy(), mode)
        np.testing.assert_allclose(out, ref.numpy(), rtol=1e-5, atol=1e-6)
    # backward: oracle scatter == autograd of grid_sample
    tex2 = tex.clone().requires_grad_(True)
    y = torch.nn.functional.grid_sample(tex2.expand(2, -1, -1, -1), grid, mode='bilinear', align_corners=False,
                                        padding_mode='border').permute(0, 2, 3, 1)
    go = torch.rand(y.shape, generator=g)
    (y * go).sum().backward()
    gt = og.texture_mapping_bwd(go.numpy(), uv.numpy(), 16)
    np.testing.assert_allclose(gt, tex2.grad[0].numpy(), rtol=1e-4, atol=1e-5)


def test_raster_small_scene_properties(meshes):
    """Brute-force raster on the bundled sphere: silhouette is a disc, depth negative, indices in range."""
    v = og.normalize_mesh(meshes['sphere_v'], 0.6, 0.0)
    f = meshes['sphere_f'].astype(np.int64)
    cam = og.get_camera_from_multiple_view(np.float32([np.pi / 3]), np.float32([0.3]), np.float32([1.5]), 0.0)
    proj = og.generate_perspective_projection(np.pi / 3)
    fvc, fvi, fn = og.prepare_vertices(v[None], f, proj, cam)
    d, idx = og.rasterize(48, 48, fvc[..., 2], fvi, fvc[..., 2:3])
    assert idx.max() < f.shape[0] and idx.min() == -1
    assert (d[idx >= 0] < 0).all() and (d[idx < 0] == 0).all()
    cover = (idx[0] >= 0).mean()
    assert 0.15 < cover < 0.6
    # visible faces face the camera (z-normal > 0) for a closed convex mesh
    assert (fn[0, np.unique(idx[idx >= 0]), 2] > -1e-3).all()


def test_raw2outputs_against_torch_formula():
    g = torch.Generator().manual_seed(1)
    raw = torch.randn(5, 33, 4, generator=g)
    z = torch.sort(torch.rand(5, 33, generator=g) * 4 + 2, -1).values
    rd = torch.randn(5, 3, generator=g)
    dists = torch.cat([z[..., 1:] - z[..., :-1], torch.full((5, 1), 1e10)], -1) * rd.norm(dim=-1, keepdim=True)
    alpha = 1. - torch.exp(-torch.relu(raw[..., 3]) * dists)
    w = alpha * torch.cumprod(torch.cat([torch.ones(5, 1), 1. - alpha + 1e-10], -1), -1)[:, :-1]
    rgb = (w[..., None] * torch.sigmoid(raw[..., :3])).sum(-2)
    o = og.raw2outputs(raw.numpy(), z.numpy(), rd.numpy())
    np.testing.assert_allclose(o[0], rgb.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(o[3], w.numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(o[4], (w * z).sum(-1).numpy(), rtol=1e-5, atol=1e-6)
