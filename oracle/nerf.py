"""ORACLE (test infrastructure) — numpy restatement of src/run_nerf_helpers.py.

  embed          : run_nerf_helpers.py:15-65   (Embedder, get_embedder(multires); freq = 2**linspace(0,L-1,L), no pi)
  nerf2d_forward : run_nerf_helpers.py:68-135  (NeRF2D; skip-cat AFTER layer `skips`, input first)
  nerf2d_backward: reverse-mode derivative of the same forward (what autograd computes for run_nerf_helpers.py:106-135,
                   with the (tanh+1)/2 head of textured_mesh.py:298-301 when grad_tex is given)
  get_rays       : run_nerf_helpers.py:139-148
  ndc_rays       : run_nerf_helpers.py:162-180
  sample_pdf     : run_nerf_helpers.py:184-225
Pinned by tests/golden/reference_vectors.npz (outputs of the reference's own functions)."""
import numpy as np


def embed(x, multires=10):
    x = np.asarray(x, np.float32)
    freqs = (2.0 ** np.linspace(0.0, multires - 1, multires)).astype(np.float32)
    outs = [x]
    for f in freqs:
        outs.append(np.sin(x * f))
        outs.append(np.cos(x * f))
    return np.concatenate(outs, -1).astype(np.float32)


def nerf2d_forward(e, weights, biases, out_w, out_b, skips=(4,), dtype=np.float32):
    """weights[i]: [out,in] like nn.Linear; returns pre-activation outputs [N, output_ch]."""
    e = np.asarray(e, dtype)
    h = e
    for i, (w, b) in enumerate(zip(weights, biases)):
        h = h @ np.asarray(w, dtype).T + np.asarray(b, dtype)
        h = np.maximum(h, 0)
        if i in skips:
            h = np.concatenate([e, h], -1)
    return h @ np.asarray(out_w, dtype).T + np.asarray(out_b, dtype)


def nerf2d_backward(e, weights, biases, out_w, out_b, grad_raw=None, grad_tex=None, skips=(4,), dtype=np.float64, masks=None,
                    return_pre=False):
    """Parameter gradients of NeRF2D.  grad_raw: d loss / d raw [N,C]; grad_tex: d loss / d ((tanh(raw)+1)/2) [N,C].
    Returns (gws, gbs) with index D = output_linear, nn.Linear layouts.
    masks (optional, list of bool [N,W]): the ReLU derivative pattern to use instead of (pre-activation > 0) — a comparison
    against an fp32 implementation must share the pattern, since a unit whose pre-activation is ~1e-7 may round to either
    side and one flipped unit moves a gradient entry by a whole term.  return_pre: also return the pre-activations."""
    e = np.asarray(e, dtype)
    ins, acts, pre = [], [], []
    h = e
    for i, (w, b) in enumerate(zip(weights, biases)):
        ins.append(h)
        z = h @ np.asarray(w, dtype).T + np.asarray(b, dtype)
        pre.append(z)
        h = np.maximum(z, 0)
        acts.append(h)
        if i in skips:
            h = np.concatenate([e, h], -1)
    raw = h @ np.asarray(out_w, dtype).T + np.asarray(out_b, dtype)
    g = np.zeros_like(raw)
    if grad_raw is not None:
        g = g + np.asarray(grad_raw, dtype)
    if grad_tex is not None:
        g = g + np.asarray(grad_tex, dtype) * 0.5 * (1 - np.tanh(raw) ** 2)
    D = len(weights)
    gws, gbs = [None] * (D + 1), [None] * (D + 1)
    gws[D] = g.T @ h
    gbs[D] = g.sum(0)
    dh = g @ np.asarray(out_w, dtype)                   # wrt the input of output_linear (no skip-cat after the last layer)
    for i in range(D - 1, -1, -1):
        if i in skips:
            dh = dh[:, e.shape[1]:]                     # the cat put the (non-trainable) embedding first
        dz = dh * ((acts[i] > 0) if masks is None else masks[i])
        gws[i] = dz.T @ ins[i]
        gbs[i] = dz.sum(0)
        dh = dz @ np.asarray(weights[i], dtype)
    return (gws, gbs, pre) if return_pre else (gws, gbs)


def texture_from_mlp(mlp_out, res):
    """src/models/textured_mesh.py:298-301: (tanh+1)/2, [res*res,3] -> [1,3,res,res]."""
    t = (np.tanh(mlp_out.astype(np.float32)) + 1) / 2
    return t.reshape(1, res, res, 3).transpose(0, 3, 1, 2)


def uv_grid(res):
    """textured_mesh.py:269-273: meshgrid(linspace, linspace, indexing='xy') -> row i <-> v, col j <-> u."""
    import torch  # torch.linspace's float32 rounding (symmetric fill) is what the reference feeds the embedder
    l = torch.linspace(0, 1, res).numpy()
    u, v = np.meshgrid(l, l, indexing='xy')
    return np.stack([u, v], -1).reshape(-1, 2)


def get_rays(H, W, K, c2w):
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    dirs = np.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -np.ones_like(i)], -1)
    c2w = np.asarray(c2w, np.float32)
    rays_d = np.sum(dirs[..., None, :] * c2w[:3, :3], -1)
    rays_o = np.broadcast_to(c2w[:3, -1], rays_d.shape)
    return rays_o.astype(np.float32), rays_d.astype(np.float32)


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    o0 = -1. / (W / (2. * focal)) * rays_o[..., 0] / rays_o[..., 2]
    o1 = -1. / (H / (2. * focal)) * rays_o[..., 1] / rays_o[..., 2]
    o2 = 1. + 2. * near / rays_o[..., 2]
    d0 = -1. / (W / (2. * focal)) * (rays_d[..., 0] / rays_d[..., 2] - rays_o[..., 0] / rays_o[..., 2])
    d1 = -1. / (H / (2. * focal)) * (rays_d[..., 1] / rays_d[..., 2] - rays_o[..., 1] / rays_o[..., 2])
    d2 = -2. * near / rays_o[..., 2]
    return np.stack([o0, o1, o2], -1).astype(np.float32), np.stack([d0, d1, d2], -1).astype(np.float32)


def sample_pdf(bins, weights, N_samples, det=False, pytest=False, u=None):
    bins = np.asarray(bins, np.float32)
    weights = np.asarray(weights, np.float32) + np.float32(1e-5)
    pdf = weights / np.sum(weights, -1, keepdims=True)
    cdf = np.cumsum(pdf, -1, dtype=np.float32)
    cdf = np.concatenate([np.zeros_like(cdf[..., :1]), cdf], -1)
    shape = list(cdf.shape[:-1]) + [N_samples]
    if u is None:
        if det:
            u = np.broadcast_to(np.linspace(0., 1., N_samples, dtype=np.float32), shape)
        else:
            u = np.random.rand(*shape).astype(np.float32)
        if pytest:
            np.random.seed(0)
            if det:
                u = np.broadcast_to(np.linspace(0., 1., N_samples), shape).astype(np.float32)
            else:
                u = np.random.rand(*shape).astype(np.float32)
    u = np.ascontiguousarray(u, np.float32)
    out = np.empty(shape, np.float32)
    nb = cdf.shape[-1]
    flat_c = cdf.reshape(-1, nb); flat_b = bins.reshape(-1, nb); flat_u = u.reshape(-1, N_samples)
    flat_o = out.reshape(-1, N_samples)
    for r in range(flat_c.shape[0]):
        inds = np.searchsorted(flat_c[r], flat_u[r], side='right')
        below = np.maximum(0, inds - 1)
        above = np.minimum(nb - 1, inds)
        c0, c1 = flat_c[r][below], flat_c[r][above]
        b0, b1 = flat_b[r][below], flat_b[r][above]
        denom = c1 - c0
        denom = np.where(denom < 1e-5, np.float32(1.0), denom)
        t = (flat_u[r] - c0) / denom
        flat_o[r] = b0 + t * (b1 - b0)
    return out
