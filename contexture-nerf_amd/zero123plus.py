"""Zero123++ wrapper pieces on the HIP UNet engine.  The reference keeps this file as a commented-out specification
(src/zero123plus.py; its live path loads the same code remotely as `custom_pipeline="sudo-ai/zero123plus-pipeline"`,
src/training/trainer.py:296-315); what is built here is the part that sits on the denoise hot path:

  RefOnlyNoisedUNet  (src/zero123plus.py:164-237) with ReferenceOnlyAttnProc (:127-161) as two engine passes:
      'w' over the noised condition latent parks every attn1 input, 'r' over the sample appends them to the self-attention K/V
      (the unconditional row of a CFG batch attends without them, `is_cfg_guidance`),
  scale_latents / unscale_latents / scale_image / unscale_image (:240-257; mirrors in utils.py).

  DepthControlUNet  (:260-298): a ControlNetModel engine whose residuals the UNet adds to its skip tensors / mid output.

  Zero123PlusPipeline.__call__ / run_sd_pipeline's denoising loop (:411-746, 748-833) over tensors: condition-image latent,
      CFG pair, EulerAncestral steps (or the explicit one-step schedule of the SDS loop) with `callback_on_step_end` exposing
      `noise_pred`, unscale_latents -> vae.decode -> unscale_image.  The PIL / CLIP preprocessing and the vision / text encoders are
      not part of it (no weights offline): `prompt_embeds` / `global_embeds` come in as tensors (seeded stand-ins otherwise); the
      inpaint / blend extension of run_sd_pipeline (:436-440, 650-708) is not built.
"""
import types
import torch
from . import _lib as L
from .utils import scale_latents, unscale_latents, scale_image, unscale_image   # noqa: F401  (re-exported like the reference module)


class RefOnlyNoisedUNet(torch.nn.Module):
    def __init__(self, unet, train_sched, val_sched):
        super().__init__()
        self.unet = unet
        self.train_sched = train_sched
        self.val_sched = val_sched
        self._bank = None

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.__dict__['unet'] if 'unet' in self.__dict__ else super().__getattr__('unet'), name)

    def forward_cond(self, noisy_cond_lat, timestep, encoder_hidden_states, class_labels, ref_dict, is_cfg_guidance, **kwargs):
        if is_cfg_guidance:
            encoder_hidden_states = encoder_hidden_states[1:]
        _, self._bank = self.unet.forward_ref(noisy_cond_lat, timestep, encoder_hidden_states, 'w', bank=self._bank)
        ref_dict['bank'] = self._bank

    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, *args, cross_attention_kwargs,
                down_block_res_samples=None, mid_block_res_sample=None, **kwargs):
        cond_lat = cross_attention_kwargs['cond_lat']
        is_cfg_guidance = cross_attention_kwargs.get('is_cfg_guidance', False)
        noise = torch.randn_like(cond_lat)
        sched = self.train_sched if self.training else self.val_sched
        t = timestep.reshape(-1) if isinstance(timestep, torch.Tensor) else torch.tensor([timestep])
        noisy_cond_lat = sched.add_noise(cond_lat, noise, t.cpu())
        noisy_cond_lat = sched.scale_model_input(noisy_cond_lat, t)
        ref_dict = {}
        self.forward_cond(noisy_cond_lat, float(t[0]), encoder_hidden_states, class_labels, ref_dict, is_cfg_guidance)
        bank = ref_dict.pop('bank')
        if down_block_res_samples is not None:
            with self.unet.residuals(down_block_res_samples):
                out, _ = self.unet.forward_ref(sample, float(t[0]), encoder_hidden_states, 'r', bank=bank,
                                               ref_row0=1 if is_cfg_guidance else 0)
        else:
            out, _ = self.unet.forward_ref(sample, float(t[0]), encoder_hidden_states, 'r', bank=bank, ref_row0=1 if is_cfg_guidance else 0)
        return out


class DepthControlUNet(torch.nn.Module):
    def __init__(self, unet, controlnet=None, conditioning_scale=1.0):
        super().__init__()
        self.unet = unet
        if controlnet is None:
            from .unet import ControlNetModel
            inner = unet.unet
            controlnet = ControlNetModel(inner.config, device=inner.device)     # from_unet topology, fresh (seeded) weights offline
        self.controlnet = controlnet
        self.conditioning_scale = conditioning_scale

    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, *args, cross_attention_kwargs, **kwargs):
        cross_attention_kwargs = dict(cross_attention_kwargs)
        control_depth = cross_attention_kwargs.pop('control_depth')
        t = float(timestep.reshape(-1)[0]) if isinstance(timestep, torch.Tensor) else float(timestep)
        down_block_res_samples, mid_block_res_sample = self.controlnet(
            sample, t, encoder_hidden_states=encoder_hidden_states, controlnet_cond=control_depth,
            conditioning_scale=self.conditioning_scale, return_dict=False)
        return self.unet(sample, timestep, encoder_hidden_states=encoder_hidden_states,
                         down_block_res_samples=down_block_res_samples, mid_block_res_sample=mid_block_res_sample,
                         cross_attention_kwargs=cross_attention_kwargs)


class Zero123PlusPipeline:
    """Tensor-level mirror of the pipeline's __call__ (src/zero123plus.py:748-833) and of the denoising loop of run_sd_pipeline
    (:604-746, without the inpaint / blend branch).  unet = DepthControlUNet(RefOnlyNoisedUNet(...)) or RefOnlyNoisedUNet(...)."""

    @property
    def inpaint_unet(self):
        """`pipeline.inpaint_unet = self.diffusion.inpaint_unet` (trainer.py:312): an engine set by the caller, or resolved on first
        use from `inpaint_unet_source` (a StableDiffusion, which builds its 9-channel UNet lazily: it never runs on the live path)."""
        v = self.__dict__.get('_inpaint_unet')
        src = self.__dict__.get('inpaint_unet_source')
        return v if v is not None or src is None else src.inpaint_unet

    @inpaint_unet.setter
    def inpaint_unet(self, v):
        self.__dict__['_inpaint_unet'] = v

    def __init__(self, vae, unet, scheduler, ramping_coefficients=None):
        self.vae, self.unet, self.scheduler = vae, unet, scheduler
        self.ramping_coefficients = ramping_coefficients

    def encode_condition_image(self, image):
        return self.vae.encode(image).latent_dist.sample()

    @torch.no_grad()
    def __call__(self, image, prompt_embeds=None, global_embeds=None, guidance_scale=4.0, depth_image=None, output_type="pt",
                 width=640, height=960, num_inference_steps=28, timesteps=None, latents=None, generator=None,
                 callback_on_step_end=None, callback_on_step_end_tensor_inputs=("latents",),
                 use_inpaint=False, use_blending=False, latent_mask_grid=None, latent_renders_grid=None, masked_input_latents=None):
        """image: [1,3,H,W] in [-1,1] (already resized / normalised for the VAE); depth_image: [1,3,height,width] in [0,1];
        prompt_embeds [1,77,D] (and optionally global_embeds [1,1,D], added with the ramping coefficients as :802-803).
        -> images [1,3,height,width] in [0,1] (output_type 'pt') or the unscaled latents ('latent').
        ConTEXTure's inpaint / blend extension of run_sd_pipeline (src/zero123plus.py:436-440, 650-708), as the spec text has it:
        use_blending — before every step outside the inpaint range the latents are re-anchored outside the mask,
        `latents * mask + add_noise(<first argument>, randn, t) * (1 - mask)` (the spec passes `latent_mask_grid` itself as the
        sample to noise, :654-659 — mirrored literally), and after the LAST step blended with the clean `latent_renders_grid`;
        use_inpaint — steps 10 < i < 20 are predicted by `self.inpaint_unet` (SD2-inpainting layout: in_channels 9) on
        cat([latents, latent_mask_grid, masked_input_latents]) instead of the reference-only / ControlNet stack."""
        dev = image.device
        inner = self.unet.unet if hasattr(self.unet, 'controlnet') else self.unet
        cfg = inner.unet.config
        do_cfg = guidance_scale > 1
        cond_lat = self.encode_condition_image(image)
        if do_cfg:
            cond_lat = torch.cat([self.encode_condition_image(torch.zeros_like(image)), cond_lat])
        if prompt_embeds is None:
            g = torch.Generator().manual_seed(0)
            prompt_embeds = torch.randn(1, 77, cfg['cross_attention_dim'], generator=g).to(dev)
        if global_embeds is not None:
            ramp = torch.as_tensor(self.ramping_coefficients if self.ramping_coefficients is not None else [0.0] * prompt_embeds.shape[1],
                                   dtype=prompt_embeds.dtype, device=dev).unsqueeze(-1)
            prompt_embeds = prompt_embeds + global_embeds * ramp
        if do_cfg:                                               # negative prompt embeds first (encode_prompt's ordering)
            prompt_embeds = torch.cat([torch.zeros_like(prompt_embeds), prompt_embeds])
        cak = dict(cond_lat=cond_lat)
        if hasattr(self.unet, 'controlnet'):
            if depth_image is None:
                raise L.CtxError("Zero123PlusPipeline: the UNet carries a ControlNet, pass depth_image")
            if do_cfg:     # keep the CFG pair of the depth grid across calls: the ControlNet caches its embedding per tensor (identity, version)
                key = (depth_image.data_ptr(), depth_image._version, tuple(depth_image.shape))
                if getattr(self, '_depth_key', None) != key:
                    self._depth_key, self._depth_pair = key, torch.cat([depth_image] * 2)
                cak['control_depth'] = self._depth_pair
            else:
                cak['control_depth'] = depth_image
        sch = self.scheduler
        sch.set_timesteps(num_inference_steps, timesteps=timesteps) if timesteps is not None else sch.set_timesteps(num_inference_steps)
        # diffusers 0.27.2 `prepare_latents` (spec: src/zero123plus.py:612-623) multiplies by init_noise_sigma whether it drew the
        # latents or the caller supplied them (the SDS loop passes latents=latents_noisy; 1.0 under the reference's DDPMScheduler,
        # sqrt(sigma_max^2 + 1) under the pipeline's stock EulerAncestral scheduler)
        if latents is None:
            latents = torch.randn(1, cfg['in_channels'], height // 8, width // 8, generator=generator, device=dev)
        latents = latents * sch.init_noise_sigma
        if (use_inpaint or use_blending) and latent_mask_grid is None:
            raise L.CtxError("Zero123PlusPipeline: use_inpaint / use_blending need latent_mask_grid")
        if use_blending and latent_renders_grid is None:
            raise L.CtxError("Zero123PlusPipeline: use_blending needs latent_renders_grid")
        if use_inpaint and (masked_input_latents is None or getattr(self, 'inpaint_unet', None) is None):
            raise L.CtxError("Zero123PlusPipeline: use_inpaint needs masked_input_latents and pipeline.inpaint_unet (trainer.py:312)")
        n_steps = len(sch.timesteps)
        for i, t in enumerate(sch.timesteps):
            is_inpaint_range = use_inpaint and (10 < i < 20)                                    # :650
            if not is_inpaint_range and use_blending:                                           # :651-661
                noises_latent = torch.randn(latents.shape, generator=generator, device=dev, dtype=latents.dtype)
                noised = sch.add_noise(latent_mask_grid, noises_latent, t.reshape(1).cpu())
                latents = latents * latent_mask_grid + noised * (1 - latent_mask_grid)
            if not is_inpaint_range:
                x = torch.cat([latents] * 2) if do_cfg else latents
                x = sch.scale_model_input(x, t)
                noise_pred = self.unet(x, t.reshape(1), prompt_embeds, cross_attention_kwargs=cak)['sample']
            else:                                                                               # :676-690
                xin = torch.cat([latents, latent_mask_grid, masked_input_latents], dim=1)
                xin = torch.cat([xin] * 2) if do_cfg else xin
                xin = sch.scale_model_input(xin, t)
                noise_pred = self.inpaint_unet(xin, float(t), encoder_hidden_states=prompt_embeds)['sample']
            if do_cfg:
                nu, nt = noise_pred.chunk(2)
                noise_pred = nu + guidance_scale * (nt - nu)
            latents = sch.step(noise_pred, t, latents, generator=generator)['prev_sample']
            if i == n_steps - 1 and use_blending:                                               # :705-708
                latents = latents * latent_mask_grid + latent_renders_grid * (1 - latent_mask_grid)
            if callback_on_step_end is not None:
                loc = dict(latents=latents, noise_pred=noise_pred, prompt_embeds=prompt_embeds)
                out = callback_on_step_end(self, i, t, {k: loc[k] for k in callback_on_step_end_tensor_inputs})
                latents = out.pop("latents", latents) if isinstance(out, dict) else latents
        latents = unscale_latents(latents)
        if output_type == "latent":
            return types.SimpleNamespace(images=latents)
        img = unscale_image(self.vae.decode(latents / 0.18215).sample)
        return types.SimpleNamespace(images=(img / 2 + 0.5).clamp(0, 1))
