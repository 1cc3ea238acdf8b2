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
      `noise_pred`, unscale_latents -> vae.decode -> unscale_image, incl. ConTEXTure's inpaint / blend extension (:436-440, 650-708).

  ConditionEncoder  (:772-803): the condition image's CLIP path — `feature_extractor_clip` -> `vision_encoder(...).image_embeds` ->
      `global_embeds`, added with the pipeline's `ramping_coefficients` to the text encoding of the (empty) prompt — read from a
      LOCAL pipeline directory with transformers (`local_files_only`): once per painted mesh, off the per-step path.  Without such a
      directory (nothing can be fetched offline) `prompt_embeds` / `global_embeds` come in as tensors or seeded stand-ins.
"""
import json
import os
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


class ConditionEncoder:
    """The condition-image / prompt encoders of the Zero123++ pipeline (src/zero123plus.py:772-803; the pipeline's `model_index.json`
    names them `feature_extractor_clip`, `vision_encoder`, `tokenizer`, `text_encoder` and carries `ramping_coefficients`), read
    from a LOCAL directory in that layout through transformers with `local_files_only=True` — nothing is fetched by name.
    Host-side torch modules evaluated once per mesh; not part of the denoise hot path."""

    def __init__(self, model_dir, device='cpu'):
        from transformers import CLIPImageProcessor, CLIPVisionModelWithProjection, CLIPTextModel, CLIPTokenizer
        d = str(model_dir)
        for sub in ('feature_extractor_clip', 'vision_encoder'):
            if not os.path.isdir(os.path.join(d, sub)):
                raise L.CtxError(f"ConditionEncoder: {d} has no {sub}/ (expected the zero123plus pipeline layout)")
        self.device = device
        idx = os.path.join(d, 'model_index.json')
        self.ramping_coefficients = json.load(open(idx)).get('ramping_coefficients') if os.path.exists(idx) else None
        self.feature_extractor_clip = CLIPImageProcessor.from_pretrained(os.path.join(d, 'feature_extractor_clip'), local_files_only=True)
        self.vision_encoder = CLIPVisionModelWithProjection.from_pretrained(os.path.join(d, 'vision_encoder'), local_files_only=True).to(device).eval()
        self.tokenizer = self.text_encoder = None
        if os.path.isdir(os.path.join(d, 'tokenizer')) and os.path.isdir(os.path.join(d, 'text_encoder')):
            self.tokenizer = CLIPTokenizer.from_pretrained(os.path.join(d, 'tokenizer'), local_files_only=True)
            self.text_encoder = CLIPTextModel.from_pretrained(os.path.join(d, 'text_encoder'), local_files_only=True).to(device).eval()

    @torch.no_grad()
    def global_embeds(self, image01):
        """image01 [1,3,H,W] in [0,1] (the condition RGB, transparent part already greyed by to_rgb_image) -> [1,1,D]:
        `feature_extractor_clip(images=image).pixel_values` -> `vision_encoder(...).image_embeds.unsqueeze(-2)` (:773, 784-786).  The
        processor sees the same 8-bit HWC pixels a PIL image would hand it."""
        arr = (image01[0].detach().float().clamp(0, 1) * 255).round().to(torch.uint8).permute(1, 2, 0).cpu().numpy()
        pv = self.feature_extractor_clip(images=arr, return_tensors='pt').pixel_values.to(self.device)
        return self.vision_encoder(pv, output_hidden_states=False).image_embeds.unsqueeze(-2).float()

    @torch.no_grad()
    def encode_prompt(self, prompt=""):
        """`self.encode_prompt(prompt, device, 1, False)[0]` (:788-794): the text encoder's last hidden state, padded to max length."""
        if self.text_encoder is None:
            raise L.CtxError("ConditionEncoder: the directory carries no tokenizer/ + text_encoder/")
        tok = self.tokenizer
        ids = tok([prompt] if isinstance(prompt, str) else list(prompt), padding='max_length', max_length=tok.model_max_length,
                  truncation=True, return_tensors='pt').input_ids
        return self.text_encoder(ids.to(self.device))[0].float()

    def ramp(self, n_tokens, dtype=torch.float32):
        rc = self.ramping_coefficients
        if rc is None:
            raise L.CtxError("ConditionEncoder: model_index.json carries no ramping_coefficients")
        if len(rc) != n_tokens:
            raise L.CtxError(f"ConditionEncoder: {len(rc)} ramping coefficients for {n_tokens} prompt tokens")
        return torch.tensor(rc, dtype=dtype, device=self.device).unsqueeze(-1)

    def prompt_embeds(self, image01, prompt=""):
        """-> (encoder_hidden_states [1,L,D] = encode_prompt(prompt) + global_embeds * ramp  (:800-803),
               negative [1,L,D] = encode_prompt("") — what run_sd_pipeline's own encode_prompt puts in the unconditional half (:582-598))."""
        ehs = self.encode_prompt(prompt)
        ge = self.global_embeds(image01)
        return ehs + ge * self.ramp(ehs.shape[1], ehs.dtype), self.encode_prompt("")


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

    def __init__(self, vae, unet, scheduler, ramping_coefficients=None, condition_encoder=None):
        self.vae, self.unet, self.scheduler = vae, unet, scheduler
        self.condition_encoder = condition_encoder
        if ramping_coefficients is None and condition_encoder is not None:
            ramping_coefficients = condition_encoder.ramping_coefficients
        self.ramping_coefficients = ramping_coefficients

    def encode_condition_image(self, image):
        return self.vae.encode(image).latent_dist.sample()

    @torch.no_grad()
    def __call__(self, image, prompt_embeds=None, global_embeds=None, prompt="", negative_prompt_embeds=None, guidance_scale=4.0,
                 depth_image=None, output_type="pt",
                 width=640, height=960, num_inference_steps=28, timesteps=None, latents=None, generator=None,
                 callback_on_step_end=None, callback_on_step_end_tensor_inputs=("latents",),
                 use_inpaint=False, use_blending=False, latent_mask_grid=None, latent_renders_grid=None, masked_input_latents=None):
        """image: [1,3,H,W] in [-1,1] (already resized / normalised for the VAE); depth_image: [1,3,height,width] in [0,1];
        prompt_embeds [1,77,D] (and optionally global_embeds [1,1,D], added with the ramping coefficients as :802-803); with a
        `condition_encoder` (a local pipeline directory) and no prompt_embeds, both come from the CLIP vision / text encoders as in
        the reference, and the unconditional half is the encoding of "" (zeros when there is no text encoder).
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
        ce = self.condition_encoder
        if negative_prompt_embeds is None:                       # precomputed once per mesh by the SDS loop (trainer.paint_zero123plus)
            negative_prompt_embeds = getattr(self, 'negative_prompt_embeds', None)
        if prompt_embeds is None and ce is not None:             # :772-803 through the local encoders; `image` is feature_extractor_vae's
            if global_embeds is None:                            # 2x - 1 of the condition RGB, so (image + 1) / 2 is what the CLIP processor saw
                global_embeds = ce.global_embeds((image + 1) / 2).to(dev)
            prompt_embeds = ce.encode_prompt(prompt).to(dev)
            if negative_prompt_embeds is None and do_cfg:
                negative_prompt_embeds = ce.encode_prompt("").to(dev)
        if prompt_embeds is None:
            g = torch.Generator().manual_seed(0)
            prompt_embeds = torch.randn(1, 77, cfg['cross_attention_dim'], generator=g).to(dev)
        if global_embeds is not None:
            ramp = torch.as_tensor(self.ramping_coefficients if self.ramping_coefficients is not None else [0.0] * prompt_embeds.shape[1],
                                   dtype=prompt_embeds.dtype, device=dev).unsqueeze(-1)
            prompt_embeds = prompt_embeds + global_embeds * ramp
        if do_cfg:                                               # negative prompt embeds first (encode_prompt's ordering, :582-598)
            neg = negative_prompt_embeds if negative_prompt_embeds is not None else torch.zeros_like(prompt_embeds)     # no text encoder: zeros
            prompt_embeds = torch.cat([neg.to(prompt_embeds.dtype), prompt_embeds])
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
