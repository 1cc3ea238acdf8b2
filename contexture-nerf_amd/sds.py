"""The denoise side of one iteration of the reference's Zero123++ SDS loop (src/training/trainer.py:700-850), on the HIP
engines: six rendered views -> 3x2 grid -> scale_image -> VAE encode -> scale_latents -> DDPM noise at the DreamTime t ->
Zero123++ pipeline, one explicit step, `noise_pred` (a v prediction) through the step-end callback -> v target, Fisher-divergence
monitor, SDS gradient, `targets = z0 - grad` and the per-tile loss value.

`sds_iteration_targets` is the no-grad form (everything up to `targets`: the part that costs the UNet evaluations);
`sds_iteration` is the whole iteration body of trainer.py:700-866 with autograd: the clean latents keep their graph through
`vae.encode` (AutoencoderKL.encode_moments_with_grad: the HIP encoder backward), so `loss.backward()` reaches the rendered views and,
through texture_mapping's and the texture field's backward kernels, the UV-MLP's parameters."""
import random
import torch
from .utils import merge_tensor_with_6_elements_to_3x2_grid, split_3x2_grid_to_tensor_with_6_elements, scale_image, scale_latents

VAE_SCALING = 0.18215


def to_rgb_image(rgba):
    """The reference's to_rgb_image (src/training/trainer.py:533-543) on tensors: an RGBA image [1,4,H,W] in [0,1] is quantised to
    8 bits (torchvision's to_pil_image: mul(255).byte()) and pasted with its alpha over a grey (127) background, with PIL's
    rounding ((x*a + 127*(255-a) + 128) folded by (t + (t >> 8)) >> 8; checked against PIL in tests/test_host_cpu.py).  -> [1,3,H,W] uint8-valued float in [0,1]."""
    q = (rgba.clamp(0, 1) * 255).to(torch.uint8).to(torch.int32)
    rgb, a = q[:, :3], q[:, 3:4]

    t = rgb * a + 127 * (255 - a) + 128                       # PIL's paste: one rounded division of the blended sum by 255
    out = (t + (t >> 8)) >> 8
    return out.clamp(0, 255).float() / 255.0


def build_depth_grid(depth_maps, object_masks, size=320):
    """paint_zero123plus set-up (src/training/trainer.py:575-600): for the six novel views (batch rows 1..6 of the 7-view render)
    crop depth (3 copies) + mask-as-alpha to the mask's square region, resize to size^2 (bilinear), lay the tiles out 3 x 2
    (tile index = 3*col + row) and grey the transparent part.  depth_maps, object_masks: [7,1,H,W] -> [1,3,3*size,2*size]."""
    from .utils import get_nonzero_region_tuple
    rgba = torch.cat((depth_maps, depth_maps, depth_maps, object_masks), dim=1)
    tiles = []
    for i in range(1, depth_maps.shape[0]):
        min_h, min_w, max_h, max_w = get_nonzero_region_tuple(object_masks[i, 0])
        tiles.append(torch.nn.functional.interpolate(rgba[i:i + 1, :, min_h:max_h, min_w:max_w], (size, size), mode='bilinear',
                                                     align_corners=False))
    grid = merge_tensor_with_6_elements_to_3x2_grid(torch.cat(tiles, 0), size)
    return to_rgb_image(grid)


def views_to_grid(six, tile):
    """[6,C,t,t] (dataset order, index = 3*col + row as utils.py:326-371) -> [1,C,3t,2t]."""
    return merge_tensor_with_6_elements_to_3x2_grid(six, tile)


def encode_views(pipe, rendered_six):
    """six rendered views [6,3,s,s] in [0,1] (autograd allowed) -> scaled clean latents z0 [1,4,3s/8,2s/8] (trainer.py:722-735)."""
    tile = rendered_six.shape[-1]
    grid = views_to_grid(rendered_six, tile)                                          # trainer.py:722-727
    grid = scale_image(grid * 2 - 1)                                                  # :729-730
    z0 = pipe.vae.encode(grid).latent_dist.sample() * VAE_SCALING                     # :732-733 (autograd through the encoder)
    return scale_latents(z0), grid                                                    # :735


@torch.no_grad()
def sds_targets(pipe, z0, cond_image, depth_grid, prompt_embeds, t, alphas_cumprod, add_noise, guidance_scale=10.0, grad_scale=0.2,
                ikl_running_avg=None, size=None):
    """z0 (detached clean latents) -> dict(latents_noisy, v_pred, v, grad, targets, ikl_running_avg) (trainer.py:740-838)."""
    dev = z0.device
    tt = torch.tensor([int(t)])
    noise = torch.randn_like(z0)                                                      # :742
    latents_noisy = add_noise(z0, noise, tt)                                          # :746  x_t = sqrt(abar) z0 + sqrt(1-abar) eps
    seen = {}

    def on_step_end(p, i, ts, kw):                                                    # :774-784
        seen['v_pred'] = kw['noise_pred']
        return kw
    H, W = (z0.shape[-2] * 8, z0.shape[-1] * 8) if size is None else size
    pipe(cond_image, prompt_embeds=prompt_embeds, depth_image=depth_grid, guidance_scale=guidance_scale, num_inference_steps=1,
         timesteps=[float(t)], latents=latents_noisy, width=W, height=H, output_type='latent',
         callback_on_step_end=on_step_end, callback_on_step_end_tensor_inputs=["latents", "noise_pred"])      # :786-795
    v_pred = seen['v_pred']
    ac = alphas_cumprod[int(t)].to(dev)
    sa, sb = torch.sqrt(ac).reshape(1, 1, 1, 1), torch.sqrt(1.0 - ac).reshape(1, 1, 1, 1)
    v = sa * noise - sb * z0                                                          # :802
    fisher = torch.sum((sa.clamp(min=1e-8) / sb.clamp(min=1e-8)) ** 2 * torch.abs(v_pred - v) ** 2).item()   # :817-820
    ikl_running_avg = fisher if ikl_running_avg is None else 0.99 * ikl_running_avg + 0.01 * fisher             # :823-827
    grad = torch.nan_to_num(grad_scale * (1.0 - ac) * sa * (v_pred - v))              # :829-834
    targets = (z0 - grad).float()                                                     # :837
    return dict(latents_noisy=latents_noisy, v_pred=v_pred, v=v, grad=grad, targets=targets, ikl_running_avg=ikl_running_avg,
                fisher=fisher)


def tile_loss(z0, targets, index_to_train=None):
    """0.5 * sum-MSE on ONE random 1/6 tile of the latent grid (trainer.py:840-854)."""
    lt = z0.shape[-1] // 2
    zs, ts_ = split_3x2_grid_to_tensor_with_6_elements(z0.float(), lt), split_3x2_grid_to_tensor_with_6_elements(targets, lt)
    index = random.randint(0, 5) if index_to_train is None else index_to_train       # :842
    loss = 0.5 * torch.nn.functional.mse_loss(zs[index], ts_[index], reduction='sum') / z0.shape[0]   # :845-854
    return loss, index


def sds_iteration(pipe, rendered_six, cond_image, depth_grid, prompt_embeds, t, alphas_cumprod, add_noise, guidance_scale=10.0,
                  grad_scale=0.2, ikl_running_avg=None, index_to_train=None):
    """One SDS iteration body with autograd (trainer.py:700-866 minus the optimizer calls): returns the dict of
    sds_iteration_targets whose `loss` carries the graph back to `rendered_six`."""
    z0, _ = encode_views(pipe, rendered_six)
    r = sds_targets(pipe, z0.detach(), cond_image, depth_grid, prompt_embeds, t, alphas_cumprod, add_noise, guidance_scale, grad_scale,
                    ikl_running_avg)
    loss, index = tile_loss(z0, r['targets'], index_to_train)
    r.update(z0=z0, loss=loss, index=index)
    return r


@torch.no_grad()
def sds_iteration_targets(pipe, rendered_six, cond_image, depth_grid, prompt_embeds, t, alphas_cumprod, add_noise,
                          guidance_scale=10.0, grad_scale=0.2, ikl_running_avg=None, index_to_train=None):
    """rendered_six [6,3,s,s] in [0,1]; cond_image [1,3,H,W] in [-1,1]; depth_grid [1,3,8h,8w] in [0,1]; t: int timestep;
    alphas_cumprod: the training scheduler's table (DDPMScheduler); add_noise: its add_noise.
    -> dict(z0, latents_noisy, v_pred, v, grad, targets, loss, index, ikl_running_avg).  No autograd (see sds_iteration)."""
    return sds_iteration(pipe, rendered_six, cond_image, depth_grid, prompt_embeds, t, alphas_cumprod, add_noise, guidance_scale,
                         grad_scale, ikl_running_avg, index_to_train)
