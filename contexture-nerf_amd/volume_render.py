"""BASELINE configs[4]: "NeRF volume render 512^2 rays x 128 samples + SD2-depth refine".

The reference names this path in north_star but holds no body for it (its `get_rays` / `sample_pdf` are dead code and the
ray-march is absent), so this module is the thin host glue over the pieces that do exist on the HIP path:

  get_rays (src/run_nerf_helpers.py:139-148)  ->  stratified samples  ->  fused 3-D embed + NeRF2D(63 -> 4)
  ->  raw2outputs (nerf-pytorch; the compositing step run_nerf_helpers.py:130-133 points to)
  ->  depth normalised like Renderer.normalize_multiple_depth's output convention (closer = larger, background 0)
  ->  StableDiffusion.img2img_step (src/stable_diffusion_depth.py:284-578) on the rendered image + depth.

Multi-GPU: rays shard by contiguous row tiles with NO exchange until the image gather (`all_gather` of the
tiles); the refine step is one UNet denoise per image, i.e. replicas only.
"""
import numpy as np
import torch
import torch.distributed as dist
from . import run_nerf_helpers as rnh


def pinhole(H, W, fovy=np.pi / 3):
    """K = [[f,0,W/2],[0,f,H/2],[0,0,1]] with f = (H/2)/tan(fovy/2) (SURVEY §8d cfg 5)."""
    f = (H / 2) / np.tan(fovy / 2)
    return np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], np.float32)


def shard_rows(H, rank, world):
    """Row range [r0, r1) of this rank: contiguous tiles, remainder rows to the first ranks."""
    base, rem = divmod(H, world)
    r0 = rank * base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)


@torch.no_grad()
def render_image(field, H, W, K, c2w, near, far, N_samples, white_bkgd=False, rows=None, N_importance=0):
    """-> dict(rgb [h,W,3], depth [h,W], acc [h,W], disp [h,W]) for the row range `rows` (default: all).
    N_importance > 0 adds nerf-pytorch's hierarchical pass: sample_pdf(det=True) on the coarse weights, merged and sorted
    with the coarse samples, evaluated by the same field."""
    ro, rd = rnh.get_rays(H, W, K, c2w)
    r0, r1 = (0, H) if rows is None else rows
    ro, rd = ro[r0:r1].reshape(-1, 3), rd[r0:r1].reshape(-1, 3)
    rgb, disp, acc, wts, depth = rnh.render_rays(field, ro, rd, near, far, N_samples, white_bkgd=white_bkgd)
    if N_importance > 0:
        t = torch.linspace(0., 1., N_samples, device=ro.device)
        z = (near * (1. - t) + far * t).expand(ro.shape[0], N_samples)
        z_mid = .5 * (z[..., 1:] + z[..., :-1])
        z_fine = rnh.sample_pdf(z_mid, wts[..., 1:-1], N_importance, det=True)
        z_all, _ = torch.sort(torch.cat([z, z_fine], -1), -1)
        rgb, disp, acc, wts, depth = rnh.render_rays(field, ro, rd, near, far, z_all.shape[-1], white_bkgd=white_bkgd,
                                                     z_vals=z_all.contiguous())
    h = r1 - r0
    return {'rgb': rgb.reshape(h, W, 3), 'depth': depth.reshape(h, W), 'acc': acc.reshape(h, W), 'disp': disp.reshape(h, W)}


def gather_rows(tile, H, group=None):
    """all_gather of the ranks' row tiles [h_r, ...] -> [H, ...] (ragged tiles padded to the largest)."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return tile
    world = dist.get_world_size(group)
    hmax = -(-H // world)
    pad = torch.zeros((hmax,) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
    pad[:tile.shape[0]] = tile
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = []
    for r in range(world):
        r0, r1 = shard_rows(H, r, world)
        out.append(parts[r][:r1 - r0])
    return torch.cat(out, 0)


def depth_for_diffusion(depth, acc, thresh=0.5):
    """Rendered z-depth -> the depth-map convention img2img_step receives from the raster path (render.py:48-74):
    foreground in [0.5, 1] with closer = larger, background 0."""
    fg = acc > thresh
    out = torch.zeros_like(depth)
    if fg.any():
        d = depth[fg]
        lo, hi = d.min(), d.max()
        out[fg] = 1.0 - 0.5 * (d - lo) / (hi - lo).clamp_min(1e-8)
    return out


@torch.no_grad()
def render_and_refine(field, sd, text_z, H, W, c2w, near=0.5, far=2.5, N_samples=128, guidance_scale=7.5, strength=1.0,
                      num_inference_steps=50, fixed_seed=0, image_size=512, rank=0, world=1, group=None):
    """configs[4] end to end on this rank's rows; every rank returns the refined image [1,3,S,S] and the gathered render."""
    K = pinhole(H, W)
    tile = render_image(field, H, W, K, c2w, near, far, N_samples, rows=shard_rows(H, rank, world))
    rgb = gather_rows(tile['rgb'], H, group)
    depth = gather_rows(tile['depth'], H, group)
    acc = gather_rows(tile['acc'], H, group)
    dmap = depth_for_diffusion(depth, acc)[None, None]
    img = rgb.permute(2, 0, 1)[None].clamp(0, 1).contiguous()
    mask = torch.ones_like(dmap)
    refined, _ = sd.img2img_step(text_z, img, dmap, guidance_scale=guidance_scale, strength=strength,
                                 num_inference_steps=num_inference_steps, update_mask=mask, fixed_seed=fixed_seed,
                                 image_size=image_size)
    return refined, {'rgb': rgb, 'depth': depth, 'acc': acc}
