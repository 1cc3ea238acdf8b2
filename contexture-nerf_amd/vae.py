"""AutoencoderKL (decoder): the `self.vae.decode(latents).sample` seam of src/stable_diffusion_depth.py:976-990
(diffusers 0.27.2 class) on the HIP VAE engine.  Parameter names are diffusers' state_dict keys; offline the weights
are seeded random-init.  The encoder (`encode_imgs`) is not built: the reference's live paint path discards the encoded
render, so only decode sits on the hot path."""
import ctypes as C
import math
import types
import torch
from . import _lib as L


class VAEConfig(C.Structure):
    _fields_ = [("latent_channels", C.c_int32), ("out_channels", C.c_int32), ("n_levels", C.c_int32),
                ("block_out_channels", C.c_int32 * 4), ("layers_per_block", C.c_int32), ("groups", C.c_int32)]


SD_VAE = dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2, groups=32)


class AutoencoderKL:
    def __init__(self, config=None, device="cuda:0", seed=0, init=True):
        cfg = dict(SD_VAE if config is None else config)
        self.config = cfg
        self.device = torch.device(device)
        self._lib = L.load()
        ch = [int(c) for c in cfg['block_out_channels']]
        c = VAEConfig(cfg['latent_channels'], cfg['out_channels'], len(ch), (C.c_int32 * 4)(*(ch + [0] * (4 - len(ch)))),
                      cfg['layers_per_block'], cfg['groups'])
        self._h = self._lib.ctx_vae_create(C.byref(c))
        if not self._h:
            raise L.CtxError("ctx_vae_create: " + self._lib.ctx_last_error().decode())
        self._names, self._shapes = [], []
        shp = (C.c_int64 * 4)()
        for i in range(self._lib.ctx_vae_param_count(self._h)):
            nd = self._lib.ctx_vae_param_shape(self._h, i, shp)
            self._names.append(self._lib.ctx_vae_param_name(self._h, i).decode())
            self._shapes.append(tuple(int(shp[k]) for k in range(nd)))
        self._index = {n: i for i, n in enumerate(self._names)}
        self._ws_key = None
        if self.device.type == 'cuda':
            self._weights = torch.empty(self._lib.ctx_vae_weight_bytes(self._h), dtype=torch.uint8, device=self.device)
            self._ws = torch.empty(256, dtype=torch.uint8, device=self.device)
            self._bind()
            if init:
                self.init_random(seed)

    def __del__(self):
        try:
            if getattr(self, '_h', None):
                self._lib.ctx_vae_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def param_shapes(self):
        return dict(zip(self._names, self._shapes))

    def _bind(self):
        L.check(self._lib.ctx_vae_bind(self._h, L.ptr(self._weights), L.ptr(self._ws), self._ws.numel()))

    def _set(self, i, t):
        t = L.f32c(t, self.device)
        if tuple(t.shape) != self._shapes[i]:
            raise L.CtxError(f"{self._names[i]}: shape {tuple(t.shape)} != {self._shapes[i]}")
        L.check(self._lib.ctx_vae_set_param(self._h, i, L.ptr(t, torch.float32, self._names[i]), L.stream()))
        return t

    def load_state_dict(self, sd, strict=True):
        """Accepts a full AutoencoderKL state_dict; encoder / quant_conv keys are ignored (decoder-only engine)."""
        missing = [n for n in self._names if n not in sd]
        if strict and missing:
            raise L.CtxError(f"load_state_dict: missing {missing[:5]} ({len(missing)})")
        keep = [self._set(i, sd[n]) for n, i in self._index.items() if n in sd]
        torch.cuda.synchronize(self.device)
        return missing

    def init_random(self, seed=0):
        g = torch.Generator(device=self.device).manual_seed(seed)
        fan = {n[:-7]: math.prod(s[1:]) for n, s in zip(self._names, self._shapes) if n.endswith('.weight') and len(s) >= 2}
        for i, (n, s) in enumerate(zip(self._names, self._shapes)):
            base = n.rsplit('.', 1)[0]
            if len(s) == 1 and base not in fan:
                t = torch.ones(s, device=self.device) if n.endswith('.weight') else torch.zeros(s, device=self.device)
            else:
                t = (torch.rand(s, generator=g, device=self.device) * 2 - 1) / math.sqrt(fan[base])
            self._set(i, t)
        torch.cuda.synchronize(self.device)

    def decode(self, z):
        x = L.f32c(z, self.device)
        B, Lc, H, W = x.shape
        if Lc != self.config['latent_channels']:
            raise L.CtxError(f"vae.decode: expected {self.config['latent_channels']} latent channels, got {Lc}")
        key = (B, H, W)
        if self._ws_key != key:
            need = self._lib.ctx_vae_workspace_bytes(self._h, B, H, W)
            if need < 0:
                raise L.CtxError("vae.decode: latent h*w must be a multiple of 64")
            if self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
                self._bind()
            self._ws_key = key
        up = 2 ** (len(self.config['block_out_channels']) - 1)
        out = torch.empty(B, self.config['out_channels'], H * up, W * up, device=self.device)
        L.check(self._lib.ctx_vae_decode(self._h, L.ptr(x, torch.float32, "latents"), B, H, W, L.ptr(out), L.stream()))
        return types.SimpleNamespace(sample=out)

    def flops(self):
        return self._lib.ctx_vae_flops(self._h)
