"""AutoencoderKL: the `self.vae.decode(latents).sample` and `self.vae.encode(imgs).latent_dist.sample()` seams of
src/stable_diffusion_depth.py:971-990 (diffusers 0.27.2 class) on the HIP VAE engine.  Parameter names are diffusers'
state_dict keys; offline the weights are seeded random-init.  Only decode sits on the paint path (the reference's live path
discards the encoded render); encode is the forward half of SURVEY §8f n2."""
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
        self._n_dec = self._lib.ctx_vae_decoder_param_count(self._h)
        self._has_encoder = True
        self._ws_key = None
        self._tape_pending = False       # a training forward's tape lives in this engine's workspace until its backward runs
        self._sibling = None
        if self.device.type == 'cuda':
            self._weights = torch.empty(self._lib.ctx_vae_weight_bytes(self._h), dtype=torch.uint8, device=self.device)
            self._ws = torch.empty(256, dtype=torch.uint8, device=self.device)
            self._bind()
            if init:
                self.init_random(seed)

    def clone_shared(self):
        """A second engine handle over the SAME weight blob with its own workspace (as UNet2DConditionModel.clone_shared)."""
        o = AutoencoderKL.__new__(AutoencoderKL)
        o.config, o.device, o._lib = self.config, self.device, self._lib
        ch = [int(c) for c in self.config['block_out_channels']]
        c = VAEConfig(self.config['latent_channels'], self.config['out_channels'], len(ch), (C.c_int32 * 4)(*(ch + [0] * (4 - len(ch)))),
                      self.config['layers_per_block'], self.config['groups'])
        o._h = self._lib.ctx_vae_create(C.byref(c))
        if not o._h:
            raise L.CtxError("ctx_vae_create: " + self._lib.ctx_last_error().decode())
        o._names, o._shapes, o._index, o._n_dec = self._names, self._shapes, self._index, self._n_dec
        o._has_encoder, o._ws_key, o._tape_pending, o._sibling = self._has_encoder, None, False, None
        o._weights = self._weights
        o._ws = torch.empty(256, dtype=torch.uint8, device=self.device)
        o._bind()
        return o

    def _free_engine(self):
        """The engine to run a no-grad call on: this one, unless it holds the tape of a training forward whose backward is still
        to come (the SDS loop encodes the condition image between the grid's encode and loss.backward()): then a sibling handle
        over the same weights with its own workspace."""
        if not self._tape_pending:
            return self
        if self._sibling is None:
            self._sibling = self.clone_shared()
        self._sibling._has_encoder = self._has_encoder
        return self._sibling

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
        """Accepts a full AutoencoderKL state_dict, or a decoder-only one (post_quant_conv + decoder.*): `encode` is then off."""
        missing = [n for n in self._names if n not in sd]
        enc = set(self._names[self._n_dec:])
        if missing and all(m in enc for m in missing) and len(missing) == len(enc):
            self._has_encoder = False                       # decoder-only checkpoint
            missing = []
        elif not missing:
            self._has_encoder = True
        if strict and missing:
            raise L.CtxError(f"load_state_dict: missing {missing[:5]} ({len(missing)})")
        keep = [self._set(i, sd[n]) for n, i in self._index.items() if n in sd]
        torch.cuda.synchronize(self.device)
        return missing

    def load_file(self, path, strict=True):
        """Weights from a local safetensors file with diffusers' AutoencoderKL names (memory-mapped, any float dtype)."""
        from .safetensors_io import load_file
        return self.load_state_dict(load_file(path), strict=strict)

    @classmethod
    def from_file(cls, path, config=None, device="cuda:0", strict=True):
        vae = cls(config, device=device, init=False)
        vae.load_file(path, strict=strict)
        return vae

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
        eng = self._free_engine()
        if eng is not self:
            return eng.decode(z)
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

    def _encode_nograd(self, x):
        eng = self._free_engine()
        if eng is not self:
            return eng._encode_nograd(x)
        if not self._has_encoder:
            raise L.CtxError("vae.encode: a decoder-only state_dict was loaded (no encoder.* / quant_conv.* parameters)")
        x = L.f32c(x, self.device)
        B, Cc, H, W = x.shape
        if Cc != self.config['out_channels']:
            raise L.CtxError(f"vae.encode: expected {self.config['out_channels']} image channels, got {Cc}")
        need = self._lib.ctx_vae_encode_workspace_bytes(self._h, B, H, W)
        if need < 0:
            f = 2 ** (len(self.config['block_out_channels']) - 1)
            raise L.CtxError(f"vae.encode: H, W must be multiples of {f} with (H/{f})*(W/{f}) a multiple of 64 (got {H}x{W})")
        if self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._bind()
        self._ws_key = None
        f = 2 ** (len(self.config['block_out_channels']) - 1)
        Lc = self.config['latent_channels']
        mom = torch.empty(B, 2 * Lc, H // f, W // f, device=self.device)
        L.check(self._lib.ctx_vae_encode(self._h, L.ptr(x, torch.float32, "image"), B, H, W, L.ptr(mom), L.stream()))
        return types.SimpleNamespace(latent_dist=DiagonalGaussianDistribution(mom))

    def encode_moments_with_grad(self, x):
        """[B,3,H,W] in [-1,1] (may require grad) -> moments [B,2L,H/8,W/8] with autograd to x: the seam `vae.encode(...)` has in
        the reference's SDS loop, where the loss is backpropagated THROUGH the frozen encoder into the rendered views
        (src/training/trainer.py:732, 866).  Forward = ctx_vae_encode_train (keeps the tape in this engine's workspace),
        backward = ctx_vae_encode_bwd; one backward per forward, no other call on this engine in between.  A later training forward
        supersedes a tape that never saw its backward (the stale backward then raises); `drop_tape()` releases it explicitly."""
        return _VaeEncodeFn.apply(self, x)

    def drop_tape(self):
        """Forget a training forward whose backward will not be run (no-grad calls go back to this engine, not to a sibling)."""
        self._tape_pending = False
        self._tape_gen = getattr(self, '_tape_gen', 0) + 1

    def encode(self, x):
        """x [B,3,H,W] in [-1,1] -> namespace(latent_dist=DiagonalGaussianDistribution-like with .sample() / .mode() / .mean / .logvar).
        When x requires grad (and gradients are enabled) the moments carry the encoder's autograd."""
        if torch.is_grad_enabled() and isinstance(x, torch.Tensor) and x.requires_grad:
            if not self._has_encoder:
                raise L.CtxError("vae.encode: a decoder-only state_dict was loaded (no encoder.* / quant_conv.* parameters)")
            return types.SimpleNamespace(latent_dist=DiagonalGaussianDistribution(self.encode_moments_with_grad(x)))
        return self._encode_nograd(x)

    def flops(self):
        return self._lib.ctx_vae_flops(self._h)


class DiagonalGaussianDistribution:
    """diffusers.models.autoencoders.vae.DiagonalGaussianDistribution (0.27.2) over moments [B,2L,h,w]."""

    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, device=self.parameters.device, dtype=self.parameters.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class _VaeEncodeFn(torch.autograd.Function):
    """Autograd seam of AutoencoderKL.encode: forward / backward on the HIP engine (input gradient only: frozen VAE)."""

    @staticmethod
    def forward(ctx, vae, x):
        x = L.f32c(x, vae.device)
        B, Cc, H, W = x.shape
        if Cc != vae.config['out_channels']:
            raise L.CtxError(f"vae.encode: expected {vae.config['out_channels']} image channels, got {Cc}")
        # One tape per engine.  A NEW training forward supersedes a tape whose backward never came (its graph was dropped by an
        # exception, or the call was an eval on a tensor that happened to require grad): the generation counter makes the stale
        # context's backward fail loudly instead of differentiating through the wrong tape.
        vae._tape_gen = getattr(vae, '_tape_gen', 0) + 1
        need = vae._lib.ctx_vae_encode_train_workspace_bytes(vae._h, B, H, W)
        if need < 0:
            f = 2 ** (len(vae.config['block_out_channels']) - 1)
            raise L.CtxError(f"vae.encode: H, W must be multiples of {f} with (H/{f})*(W/{f}) a multiple of 64 (got {H}x{W})")
        if vae._ws.numel() < need:
            vae._ws = torch.empty(need, dtype=torch.uint8, device=vae.device)
            vae._bind()
        vae._ws_key = None
        f = 2 ** (len(vae.config['block_out_channels']) - 1)
        mom = torch.empty(B, 2 * vae.config['latent_channels'], H // f, W // f, device=vae.device)
        L.check(vae._lib.ctx_vae_encode_train(vae._h, L.ptr(x, torch.float32, "image"), B, H, W, L.ptr(mom), L.stream()))
        ctx.vae, ctx.shape, ctx.gen = vae, (B, Cc, H, W), vae._tape_gen
        vae._tape_pending = True
        return mom

    @staticmethod
    def backward(ctx, g):
        vae = ctx.vae
        if ctx.gen != vae._tape_gen or not vae._tape_pending:
            raise L.CtxError("vae.encode backward: this engine's tape was overwritten by a later training forward (or already "
                             "consumed); one backward per forward, in order")
        vae._tape_pending = False
        g = L.f32c(g, vae.device)
        gmax = float(g.abs().max())
        if gmax != gmax or gmax == float('inf'):               # NaN / Inf in the incoming gradient: propagate, as torch autograd would
            return None, torch.full(ctx.shape, float('nan'), device=vae.device)
        if not (gmax > 0.0):
            return None, torch.zeros(ctx.shape, device=vae.device)
        gscale = 2.0 ** round(__import__('math').log2(16.0 / gmax))          # fp16 gradients: max |g| scaled to ~16, a power of two
        dx = torch.empty(ctx.shape, device=vae.device)
        L.check(vae._lib.ctx_vae_encode_bwd(vae._h, L.ptr(g, torch.float32, "grad_moments"), gscale, L.ptr(dx), L.stream()))
        return None, dx
