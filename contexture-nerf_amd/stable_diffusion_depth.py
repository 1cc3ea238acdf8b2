"""StableDiffusion: mirror of the live members of src/stable_diffusion_depth.py
(`__init__` :27-106, `get_text_embeds` :222-244, `img2img_step` :284-578 incl. inner `sample`,
`encode_imgs`/`decode_latents` :971-990, `get_timesteps` :992-999) with the denoise loop on the HIP UNet engine.

Offline facts: no SD2 weights / tokenizer / diffusers => the UNet is the SD2-depth ARCHITECTURE with
seeded random-init weights (or a local safetensors state_dict passed as `unet_state_dict`), text embeddings come
from a caller-supplied encoder or are seeded random [2,77,1024] (seed = a stable SHA-256 digest of the prompt, the
same in every process); the VAE encoder and decoder are the HIP engine of vae.py (AutoencoderKL architecture,
random-init offline or a local state_dict / safetensors file).

Additions over the reference: `image_size` is a parameter (the reference hard-wires 512, :519) because
BASELINE.json's configs run 256^2 / 512^2 / 768^2.
"""
import hashlib
import os
import torch
import torch.nn.functional as F
from . import _lib as L
from .scheduler import PNDMScheduler
from .unet import UNet2DConditionModel
from .vae import AutoencoderKL
from .utils import seed_everything


class StableDiffusion:
    def __init__(self, device, model_name='stabilityai/stable-diffusion-2-depth', concept_name=None, concept_path=None,
                 latent_mode=True, min_timestep=0.02, max_timestep=0.98, no_noise=False, use_inpaint=False,
                 second_model_type=None, guess_mode=False, unet=None, unet_state_dict=None, vae=None, text_encoder=None,
                 seed=0):
        if second_model_type not in (None,):
            raise L.CtxError(f"second_model_type={second_model_type!r}: dead branch in the reference (needs src/zero123), not built")
        self.device = device
        self.latent_mode = latent_mode
        self.no_noise = no_noise
        self.use_inpaint = False                      # never activates in the reference (paint_step stays 0, trainer.py:1048)
        self.second_model_type = second_model_type
        self.num_train_timesteps = 1000
        self.min_step = int(self.num_train_timesteps * min_timestep)
        self.max_step = int(self.num_train_timesteps * max_timestep)
        # `model_name` may be a LOCAL directory in the diffusers layout (unet/diffusion_pytorch_model.safetensors,
        # vae/diffusion_pytorch_model.safetensors): the files are read by safetensors_io (nothing is fetched by name offline)
        local = model_name if isinstance(model_name, str) and os.path.isdir(model_name) else None
        self._local_dir, self._seed, self._inpaint_unet = local, seed, None
        unet_file = os.path.join(local, 'unet', 'diffusion_pytorch_model.safetensors') if local else None
        vae_file = os.path.join(local, 'vae', 'diffusion_pytorch_model.safetensors') if local else None
        from_file = unet is None and unet_state_dict is None and unet_file is not None and os.path.exists(unet_file)
        self.unet = unet if unet is not None else UNet2DConditionModel(device=device, seed=seed,
                                                                       init=unet_state_dict is None and not from_file)
        if unet_state_dict is not None:
            self.unet.load_state_dict(unet_state_dict)
        elif from_file:
            self.unet.load_file(unet_file)
        if vae is not None:
            self.vae = vae
        elif vae_file is not None and os.path.exists(vae_file):
            self.vae = AutoencoderKL.from_file(vae_file, device=device)
        else:
            self.vae = AutoencoderKL(device=device, seed=seed)      # random-init offline
        self.text_encoder = text_encoder
        self.tokenizer = None
        if text_encoder is None and local and os.path.isdir(os.path.join(local, 'text_encoder')) and os.path.isdir(os.path.join(local, 'tokenizer')):
            # the reference's own text path (stable_diffusion_depth.py:61-66, 222-244): transformers' CLIPTokenizer + CLIPTextModel read from
            # the LOCAL diffusers-layout directory (local_files_only: nothing is fetched by name).  The text encoder is a once-per-prompt
            # torch module, not part of the per-step hot path.
            from transformers import CLIPTextModel, CLIPTokenizer
            self.tokenizer = CLIPTokenizer.from_pretrained(os.path.join(local, 'tokenizer'), local_files_only=True)
            self._clip = CLIPTextModel.from_pretrained(os.path.join(local, 'text_encoder'), local_files_only=True).to(self.device).eval()
            self.text_encoder = self._clip_embeds
        self.scheduler = PNDMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                       num_train_timesteps=self.num_train_timesteps, steps_offset=1, skip_prk_steps=True)
        self.alphas = self.scheduler.alphas_cumprod.to(self.device)

    @property
    def inpaint_unet(self):
        """The SD2-inpainting UNet the reference loads next to the depth UNet (`stabilityai/stable-diffusion-2-inpainting`,
        in_channels 9, fp16; stable_diffusion_depth.py:73-88) and hands to the Zero123++ pipeline (trainer.py:312).  It never runs on
        the reference's live path (use_inpaint stays False), so it is built on first access: same architecture, seeded random
        init offline, or `<model_name>/inpaint_unet/diffusion_pytorch_model.safetensors` when model_name is a local directory."""
        if getattr(self, '_inpaint_unet', None) is None:
            cfg = dict(self.unet.config, in_channels=9)
            f = os.path.join(self._local_dir, 'inpaint_unet', 'diffusion_pytorch_model.safetensors') if self._local_dir else None
            if f and os.path.exists(f):
                self._inpaint_unet = UNet2DConditionModel.from_file(f, cfg, device=self.device)
            else:
                self._inpaint_unet = UNet2DConditionModel(cfg, device=self.device, seed=self._seed + 1)
        return self._inpaint_unet

    def _clip_embeds(self, prompt, negative_prompt=None):
        """stable_diffusion_depth.py:222-244: tokenise (pad to model_max_length), encode the prompt and '' (or the negative prompt),
        -> cat([uncond, cond])."""
        prompt = [prompt] if isinstance(prompt, str) else list(prompt)
        tok = self.tokenizer
        ti = tok(prompt, padding='max_length', max_length=tok.model_max_length, truncation=True, return_tensors='pt')
        negative_prompt = [''] * len(prompt) if negative_prompt is None else ([negative_prompt] if isinstance(negative_prompt, str) else list(negative_prompt))
        ui = tok(negative_prompt, padding='max_length', max_length=tok.model_max_length, return_tensors='pt')
        with torch.no_grad():
            te = self._clip(ti.input_ids.to(self.device))[0]
            ue = self._clip(ui.input_ids.to(self.device))[0]
        return torch.cat([ue, te]).float()

    def get_text_embeds(self, prompt, negative_prompt=None, seed=0):
        """-> cat([uncond, cond]) [2,77,1024].  With no encoder (offline) a seeded random embedding stands in."""
        if self.text_encoder is not None:
            return self.text_encoder(prompt, negative_prompt)
        # stable digest, not hash(): str hashing is randomised per interpreter, and every rank / run must condition its
        # views on the same embedding for the same prompt
        prompt = [prompt] if isinstance(prompt, str) else list(prompt)
        neg = [] if negative_prompt is None else ([negative_prompt] if isinstance(negative_prompt, str) else list(negative_prompt))
        key = '\0'.join(prompt) + '\1' + '\0'.join(neg) + '\1' + str(int(seed))
        g = torch.Generator().manual_seed(int.from_bytes(hashlib.sha256(key.encode('utf-8')).digest()[:4], 'little') & 0x7fffffff)
        return torch.randn(2, 77, self.unet.config['cross_attention_dim'], generator=g).to(self.device)

    def get_timesteps(self, num_inference_steps, strength):
        init_timestep = min(int(num_inference_steps * strength), num_inference_steps)
        t_start = max(num_inference_steps - init_timestep, 0)
        return self.scheduler.timesteps[t_start:], num_inference_steps - t_start

    def encode_imgs(self, imgs):
        imgs = 2 * imgs - 1
        return self.vae.encode(imgs).latent_dist.sample() * 0.18215

    def decode_latents(self, latents):
        latents = 1 / 0.18215 * latents
        with torch.no_grad():
            imgs = self.vae.decode(latents).sample
        return (imgs / 2 + 0.5).clamp(0, 1)

    def _new_scheduler(self):
        return PNDMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                             num_train_timesteps=self.num_train_timesteps, steps_offset=1, skip_prk_steps=True)

    class _Denoise:
        """The inner `sample` loop of img2img_step (stable_diffusion_depth.py:297-516) as a resumable job: construction does
        what precedes the loop, advance() is one loop body (CFG-batched UNet evaluation + fused CFG / PLMS update)."""

        def __init__(self, sd, unet, scheduler, text_embeddings, latents, depth_mask, strength, num_inference_steps, update_mask,
                     fixed_seed, guidance_scale):
            self.sd, self.unet, self.scheduler = sd, unet, scheduler
            self.text_embeddings, self.guidance_scale = text_embeddings, guidance_scale
            scheduler.set_timesteps(num_inference_steps)
            shape = (text_embeddings.shape[0] // 2, unet.in_channels - 1, depth_mask.shape[2], depth_mask.shape[3])
            noise = None                                                 # `noise` of the reference's sample(): stays None when latents is None
            if latents is None:
                latents = torch.randn(shape, device=sd.device)
                timesteps = scheduler.timesteps
            else:
                init_timestep = min(int(num_inference_steps * strength), num_inference_steps)
                timesteps = scheduler.timesteps[max(num_inference_steps - init_timestep, 0):]
                latent_timestep = timesteps[:1]
                if fixed_seed is not None:
                    seed_everything(fixed_seed)
                noise = torch.randn_like(latents)
                if update_mask is not None:
                    latents = torch.randn(shape, device=sd.device)        # gt_latents are never blended (blend commented out, :382)
                else:
                    latents = scheduler.add_noise(latents, noise, latent_timestep)
            self.latents, self.timesteps, self.i = latents, timesteps, 0
            self.noise = noise
            self.on_step = None                                          # intermediate_vis hook: called with (t, latents) before the step
            self.depth2 = torch.cat([depth_mask] * 2)

        def done(self):
            return self.i >= len(self.timesteps)

        def model_input(self):
            """-> (x [2,C+1,h,w] = the CFG pair of this view's latents with its depth, t)."""
            t = self.timesteps[self.i]
            latent_model_input = torch.cat([self.latents] * 2)
            latent_model_input = self.scheduler.scale_model_input(latent_model_input, t)
            return torch.cat([latent_model_input, self.depth2], dim=1), t

        def apply(self, noise_pred, t):
            """noise_pred [2,C,h,w] = (uncond, text) of this view -> fused CFG + PLMS update."""
            if self.on_step is not None:
                self.on_step(t, self.latents, self.noise)
            self.latents = self.scheduler.step_cfg(noise_pred, self.guidance_scale, int(t), self.latents)['prev_sample']
            self.i += 1

        def advance(self):
            x, t = self.model_input()
            self.apply(self.unet(x, float(t), encoder_hidden_states=self.text_embeddings)['sample'], t)

    def _prepare(self, inputs, original_depth_mask, update_mask, latent_mode, image_size):
        depth_mask = F.interpolate(original_depth_mask, size=(image_size // 8, image_size // 8), mode='bicubic', align_corners=False)
        if inputs is None:
            latents = None
        elif latent_mode:
            latents = inputs
        elif not hasattr(self.vae, 'encode'):
            # the encoded render only matters when update_mask is None (it is discarded otherwise, see _Denoise):
            # the reference's live call always passes update_mask, so a zero latent of the right shape is equivalent
            latents = torch.zeros(inputs.shape[0], self.unet.in_channels - 1, image_size // 8, image_size // 8, device=self.device)
        else:
            pred_rgb_small = F.interpolate(inputs, (image_size, image_size), mode='bilinear', align_corners=False)
            latents = self.encode_imgs(pred_rgb_small)
        if update_mask is not None:
            update_mask = F.interpolate(update_mask, (image_size // 8, image_size // 8), mode='nearest')
        depth_mask = 2.0 * (depth_mask - depth_mask.min()) / (depth_mask.max() - depth_mask.min()) - 1.0
        return latents, depth_mask, update_mask

    def img2img_step(self, text_embeddings, inputs, original_depth_mask, guidance_scale=100, strength=0.5,
                     num_inference_steps=50, update_mask=None, latent_mode=False, fixed_seed=None, intermediate_vis=False,
                     view_dir=None, front_image=None, phi=None, theta=None, condition_guidance_scales=None, image_size=512):
        intermediate_results = []
        latents, depth_mask, update_mask = self._prepare(inputs, original_depth_mask, update_mask, latent_mode, image_size)
        with torch.no_grad():
            job = StableDiffusion._Denoise(self, self.unet, self.scheduler, text_embeddings, latents, depth_mask, strength,
                                           num_inference_steps, update_mask, fixed_seed, guidance_scale)
            if intermediate_vis:
                job.on_step = lambda t, lat, noise: intermediate_results.append(self._vis_step(t, lat, noise))
            while not job.done():
                job.advance()
            target_latents = job.latents
            target_rgb = self.decode_latents(target_latents)
        if latent_mode:
            return target_rgb, target_latents
        return target_rgb, intermediate_results

    def _vis_step(self, t, latents, noise):
        """`intermediate_vis` of img2img_step (stable_diffusion_depth.py:500-511, LogConfig.vis_diffusion_steps): the x0 estimate
        the reference decodes at every step — `(latents - sigma_t * noise) / alpha_t` with the loop's INITIAL `noise` tensor (not the
        step's prediction; mirrored as written) — through the VAE, as an 8-bit PIL image."""
        from PIL import Image
        ac = self.scheduler.alphas_cumprod
        a_t, s_t = float(torch.sqrt(ac[int(t)])), float(torch.sqrt(1 - ac[int(t)]))
        vis_latents = (latents - s_t * noise) / a_t             # noise is None when latents was None: a TypeError, as in the reference
        image = self.decode_latents(vis_latents)
        image = image.cpu().permute(0, 2, 3, 1).numpy()
        return Image.fromarray((image[0] * 255).round().astype("uint8"))

    def img2img_step_multi(self, calls):
        """Several img2img_step calls (views of one mesh) with their denoise loops in flight together: one HIP stream, one
        engine (UNet2DConditionModel.clone_shared: all over ONE weight blob) and one scheduler per call.  `calls` = dicts of
        img2img_step keyword arguments.  Each result is what img2img_step(**call) returns on its own (same seeds, same
        deterministic kernels); two in flight finish ~1.25x sooner, three ~1.3x, because the deep UNet levels do not fill the chip."""
        n = len(calls)
        if n == 1:
            kw = dict(calls[0])
            return [self.img2img_step(kw.pop('text_embeddings'), kw.pop('inputs'), kw.pop('original_depth_mask'), **kw)]
        engines = getattr(self, '_engines', None)
        if engines is None:
            engines = self._engines = [self.unet]
        while len(engines) < n:
            engines.append(self.unet.clone_shared())
        streams = getattr(self, '_multi_streams', None)
        if streams is None:
            streams = self._multi_streams = []
        while len(streams) < n:
            streams.append(torch.cuda.Stream(self.device))
        main = torch.cuda.current_stream(self.device)
        jobs, metas = [], []
        with torch.no_grad():
            for k, kw in enumerate(calls):
                kw = dict(kw)
                image_size = kw.get('image_size', 512)
                latent_mode = kw.get('latent_mode', False)
                latents, depth_mask, update_mask = self._prepare(kw['inputs'], kw['original_depth_mask'], kw.get('update_mask'),
                                                                 latent_mode, image_size)
                jobs.append(StableDiffusion._Denoise(self, engines[k], self._new_scheduler(), kw['text_embeddings'], latents, depth_mask,
                                                     kw.get('strength', 0.5), kw.get('num_inference_steps', 50), update_mask,
                                                     kw.get('fixed_seed'), kw.get('guidance_scale', 100)))
                metas.append(latent_mode)
            for st in streams[:n]:
                st.wait_stream(main)
            while not all(j.done() for j in jobs):
                for k in range(n):
                    if not jobs[k].done():
                        with torch.cuda.stream(streams[k]):
                            jobs[k].advance()
            for st in streams[:n]:
                main.wait_stream(st)
            outs = []
            for k in range(n):
                jobs[k].latents.record_stream(main)
                rgb = self.decode_latents(jobs[k].latents)
                outs.append((rgb, jobs[k].latents) if metas[k] else (rgb, []))
        return outs

    def img2img_step_batched(self, calls, views_per_eval=6, groups_in_flight=2):
        """Several img2img_step calls (views, possibly of different meshes) denoised in LOCKSTEP as ONE UNet evaluation of batch
        2 x views_per_eval per step: rows [u_0, c_0, u_1, c_1, ...] (every view keeps its own text embeddings, depth, seed, scheduler
        state and fused CFG / PLMS update).  At M = views x 2 x h x w rows every layer is a large GEMM (no split-K slabs, full tiles),
        which one view at CFG batch 2 cannot offer: 155 view-steps/s against 145 for three streams of batch 2 and 116 for one (latent 96,
        plan table tuned for batch 12).
        Only FULL groups of views_per_eval views are batched; the remaining views (fewer than a group: a rank of a multi-GPU job that
        owns one or two views) go through img2img_step_multi — streams of batch 2 — because a padded batch would multiply their work.
        The executor's plan depends on the row count only and every row's arithmetic is independent of the other rows, so inside
        full groups a view's result does not depend on which views share its batch nor on its position (tested).  It is NOT
        bit-identical to the batch-2 loop: other tile / split-K plans sum in another order (same tolerance against the oracle).
        All calls must share image_size and num_inference_steps / strength (one timestep schedule); otherwise everything falls back
        to img2img_step_multi.
        groups_in_flight: with two or more full groups (a mesh batch), that many lockstep evaluations run concurrently on their own
        HIP streams and engine clones (one weight blob): 34.0 instead of 36.9 ms per batch-12 evaluation with two in flight
        (tools/bench_concurrent.py 96 6 12); a group's arithmetic does not depend on what runs beside it, so results are unchanged."""
        n, G = len(calls), int(views_per_eval)
        key = lambda kw: (kw.get('image_size', 512), kw.get('num_inference_steps', 50), kw.get('strength', 0.5), kw.get('latent_mode', False))
        if n < G or G < 2 or 2 * G > 16 or any(key(kw) != key(calls[0]) for kw in calls) or any(kw.get('intermediate_vis') for kw in calls):
            return self.img2img_step_multi(calls)
        nfull = (n // G) * G
        jobs, metas = [], []
        with torch.no_grad():
            for kw in calls[:nfull]:
                image_size, latent_mode = kw.get('image_size', 512), kw.get('latent_mode', False)
                latents, depth_mask, update_mask = self._prepare(kw['inputs'], kw['original_depth_mask'], kw.get('update_mask'),
                                                                 latent_mode, image_size)
                jobs.append(StableDiffusion._Denoise(self, self.unet, self._new_scheduler(), kw['text_embeddings'], latents, depth_mask,
                                                     kw.get('strength', 0.5), kw.get('num_inference_steps', 50), update_mask,
                                                     kw.get('fixed_seed'), kw.get('guidance_scale', 100)))
                metas.append(latent_mode)
            groups = [jobs[g0:g0 + G] for g0 in range(0, nfull, G)]
            steps = len(jobs[0].timesteps)
            if any(len(j.timesteps) != steps or not torch.equal(j.timesteps, jobs[0].timesteps) for j in jobs):
                raise L.CtxError("img2img_step_batched: the views of one evaluation must share their timestep schedule")
            F = max(1, min(int(groups_in_flight), len(groups)))

            def one_step(engine, grp, ctx):
                xs, t = [], None
                for j in grp:
                    x, t = j.model_input()
                    xs.append(x)
                noise = engine(torch.cat(xs), float(t), encoder_hidden_states=ctx)['sample']
                for v, j in enumerate(grp):
                    j.apply(noise[2 * v:2 * v + 2], t)

            if F == 1:
                for grp in groups:
                    ctx = torch.cat([j.text_embeddings for j in grp])
                    for _ in range(steps):
                        one_step(self.unet, grp, ctx)
            else:
                engines = getattr(self, '_engines', None)
                if engines is None:
                    engines = self._engines = [self.unet]
                while len(engines) < F:
                    engines.append(self.unet.clone_shared())
                streams = getattr(self, '_multi_streams', None)
                if streams is None:
                    streams = self._multi_streams = []
                while len(streams) < F:
                    streams.append(torch.cuda.Stream(self.device))
                main = torch.cuda.current_stream(self.device)
                for c0 in range(0, len(groups), F):
                    chunk = groups[c0:c0 + F]
                    ctxs = [torch.cat([j.text_embeddings for j in grp]) for grp in chunk]
                    for st in streams[:len(chunk)]:
                        st.wait_stream(main)
                    for _ in range(steps):
                        for k, grp in enumerate(chunk):
                            with torch.cuda.stream(streams[k]):
                                one_step(engines[k], grp, ctxs[k])
                    for st in streams[:len(chunk)]:
                        main.wait_stream(st)
                    for grp in chunk:
                        for j in grp:
                            j.latents.record_stream(main)
            outs = []
            for j, lm in zip(jobs, metas):
                rgb = self.decode_latents(j.latents)
                outs.append((rgb, j.latents) if lm else (rgb, []))
        if nfull < n:
            outs += self.img2img_step_multi(calls[nfull:])
        return outs

    def img2img_step_pair(self, calls):
        assert len(calls) == 2
        return self.img2img_step_multi(calls)
