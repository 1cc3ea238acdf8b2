"""Zero123++ wrapper pieces on the HIP UNet engine.  The reference keeps this file as a commented-out specification
(src/zero123plus.py; its live path loads the same code remotely as `custom_pipeline="sudo-ai/zero123plus-pipeline"`,
src/training/trainer.py:296-315); what is built here is the part that sits on the denoise hot path:

  RefOnlyNoisedUNet  (src/zero123plus.py:164-237) with ReferenceOnlyAttnProc (:127-161) as two engine passes:
      'w' over the noised condition latent parks every attn1 input, 'r' over the sample appends them to the self-attention K/V
      (the unconditional row of a CFG batch attends without them, `is_cfg_guidance`),
  scale_latents / unscale_latents / scale_image / unscale_image (:240-257; mirrors in utils.py).

  DepthControlUNet  (:260-298): a ControlNetModel engine whose residuals the UNet adds to its skip tensors / mid output.

Not built: the pipeline __call__ (:748-833) and its schedulers' sampling loop.
"""
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
        noisy_cond_lat = sched.add_noise(cond_lat, noise, t.to(torch.long).cpu())
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
