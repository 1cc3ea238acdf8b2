"""UNet2DConditionModel: the seam `self.unet(x, t, encoder_hidden_states=ctx)['sample']` of
src/stable_diffusion_depth.py:422-423 (diffusers 0.27.2 class), executed by the HIP engine in
libctxnerf.so (fp16 MFMA conv / GEMM / attention, fp32 accumulation).

Parameters use diffusers' state_dict names, so `load_state_dict` accepts a diffusers checkpoint
(safetensors of stabilityai/stable-diffusion-2-depth/unet) unchanged; with no checkpoint (offline) the
weights are seeded random-init with torch's default Conv2d/Linear initialisers, as BASELINE.json prescribes.
"""
import ctypes as C
import math
import torch
from . import _lib as L


class UNetConfig(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("out_channels", C.c_int32), ("n_levels", C.c_int32),
                ("block_out_channels", C.c_int32 * 4), ("heads", C.c_int32 * 4),
                ("down_attn", C.c_int32 * 4), ("up_attn", C.c_int32 * 4), ("layers_per_block", C.c_int32),
                ("cross_attention_dim", C.c_int32), ("groups", C.c_int32), ("norm_eps", C.c_float)]


SD2_DEPTH = dict(in_channels=5, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 down_attn=(True, True, True, False), up_attn=(False, True, True, True),
                 cross_attention_dim=1024, heads=(5, 10, 20, 20), groups=32, norm_eps=1e-5)


def _pad4(xs):
    xs = [int(x) for x in xs]
    return (C.c_int32 * 4)(*(xs + [0] * (4 - len(xs))))


class UNet2DConditionModel:
    def __init__(self, config=None, device="cuda:0", seed=0, init=True):
        cfg = dict(SD2_DEPTH if config is None else config)
        self.config = cfg
        self.in_channels = cfg['in_channels']
        self.device = torch.device(device)
        self._lib = L.load()
        self._h = self._create_handle()
        self._names, self._shapes = [], []
        shp = (C.c_int64 * 4)()
        for i in range(self._lib.ctx_unet_param_count(self._h)):
            nd = self._lib.ctx_unet_param_shape(self._h, i, shp)
            self._names.append(self._lib.ctx_unet_param_name(self._h, i).decode())
            self._shapes.append(tuple(int(shp[k]) for k in range(nd)))
        self._index = {n: i for i, n in enumerate(self._names)}
        self._weights = None
        self._ws = None
        self._ws_key = None
        self._t = None
        if self.device.type == 'cuda':
            self._weights = torch.empty(self._lib.ctx_unet_weight_bytes(self._h), dtype=torch.uint8, device=self.device)
            self._ws = torch.empty(256, dtype=torch.uint8, device=self.device)
            self._bind()
            if init:
                self.init_random(seed)

    def _create_handle(self):
        cfg = self.config
        c = UNetConfig(cfg['in_channels'], cfg['out_channels'], len(cfg['block_out_channels']),
                       _pad4(cfg['block_out_channels']), _pad4(cfg['heads']), _pad4(cfg['down_attn']),
                       _pad4(cfg['up_attn']), cfg['layers_per_block'], cfg['cross_attention_dim'], cfg['groups'],
                       cfg['norm_eps'])
        h = self._lib.ctx_unet_create(C.byref(c))
        if not h:
            raise L.CtxError("ctx_unet_create: " + self._lib.ctx_last_error().decode())
        return h

    def clone_shared(self):
        """A second engine over the SAME weight blob with its own workspace, so two evaluations (two views of a mesh) can be
        in flight on two HIP streams at once: the kernels of the deep UNet levels do not fill the chip, and two concurrent
        evaluations finish ~1.25x sooner than back to back (tools/bench_concurrent.py)."""
        o = UNet2DConditionModel.__new__(UNet2DConditionModel)
        o.config, o.in_channels, o.device, o._lib = self.config, self.in_channels, self.device, self._lib
        o._names, o._shapes, o._index = self._names, self._shapes, self._index
        o._h = o._create_handle()
        o._weights = self._weights                      # shared, read-only during forward
        o._ws = torch.empty(256, dtype=torch.uint8, device=self.device)
        o._ws_key, o._t = None, None
        o._bind()
        return o

    def __del__(self):
        try:
            if getattr(self, '_h', None):
                self._lib.ctx_unet_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- parameter table ---------------------------------------------------------------------------------
    def param_shapes(self):
        return dict(zip(self._names, self._shapes))

    def num_parameters(self):
        return sum(math.prod(s) for s in self._shapes)

    def _bind(self):
        L.check(self._lib.ctx_unet_bind(self._h, L.ptr(self._weights), L.ptr(self._ws), self._ws.numel()))

    def _set(self, i, t):
        t = L.f32c(t, self.device)
        if tuple(t.shape) != self._shapes[i]:
            raise L.CtxError(f"{self._names[i]}: shape {tuple(t.shape)} != {self._shapes[i]}")
        L.check(self._lib.ctx_unet_set_param(self._h, i, L.ptr(t, torch.float32, self._names[i]), L.stream()))
        return t

    def load_state_dict(self, sd, strict=True):
        missing = [n for n in self._names if n not in sd]
        extra = [k for k in sd if k not in self._index]
        if strict and (missing or extra):
            raise L.CtxError(f"load_state_dict: missing {missing[:5]}... ({len(missing)}), unexpected {extra[:5]}... ({len(extra)})")
        keep = []
        for n, i in self._index.items():
            if n in sd:
                keep.append(self._set(i, sd[n]))
        torch.cuda.synchronize(self.device)   # sources must outlive the async repack kernels
        return missing, extra

    def load_file(self, path, strict=True):
        """Weights from a local safetensors file with diffusers' parameter names (what `from_pretrained` would have fetched by
        model name, src/stable_diffusion_depth.py:58-88); fp16 / bf16 / fp32 payloads are accepted and repacked to the engine's
        fp16 layout.  The file is memory-mapped: tensors go to the device one at a time."""
        from .safetensors_io import load_file
        return self.load_state_dict(load_file(path), strict=strict)

    @classmethod
    def from_file(cls, path, config=None, device="cuda:0", strict=True):
        net = cls(config, device=device, init=False)
        net.load_file(path, strict=strict)
        return net

    def init_random(self, seed=0):
        """torch default initialisers (kaiming_uniform(a=sqrt 5) => U(-1/sqrt(fan_in), +)), norms = (1, 0)."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        fan = {}
        for n, s in zip(self._names, self._shapes):
            if n.endswith('.weight') and len(s) >= 2:
                fan[n[:-7]] = math.prod(s[1:])
        for i, (n, s) in enumerate(zip(self._names, self._shapes)):
            base = n.rsplit('.', 1)[0]
            if len(s) == 1 and base not in fan:                 # norm affine
                t = torch.ones(s, device=self.device) if n.endswith('.weight') else torch.zeros(s, device=self.device)
            else:
                b = 1.0 / math.sqrt(fan[base])
                t = (torch.rand(s, generator=g, device=self.device) * 2 - 1) * b
            self._set(i, t)
            if i % 64 == 63:
                torch.cuda.synchronize(self.device)
        torch.cuda.synchronize(self.device)

    # -- forward ---------------------------------------------------------------------------------------------
    def workspace_bytes(self, B, H, W, ctx_len):
        n = self._lib.ctx_unet_workspace_bytes(self._h, B, H, W, ctx_len)
        if n < 0:
            raise L.CtxError(self._lib.ctx_last_error().decode())
        return n

    def flops(self, B, H, W, ctx_len):
        """Algorithmic FLOPs of one forward by kernel class: {'gemm_conv', 'attention', 'other'} + launches."""
        self.workspace_bytes(B, H, W, ctx_len)            # dry run fills the counters
        out = {}
        for k, name in enumerate(('gemm_conv', 'attention', 'other')):
            n, f = C.c_int64(), C.c_double()
            L.check(self._lib.ctx_unet_stats(self._h, k, C.byref(n), C.byref(f)))
            out[name] = (n.value, f.value)
        return out

    def __call__(self, sample, timestep, encoder_hidden_states=None, down_block_additional_residuals=None,
                 mid_block_additional_residual=None, **kw):
        if down_block_additional_residuals is None:
            return self.forward(sample, timestep, encoder_hidden_states)
        with self.residuals(down_block_additional_residuals):
            return self.forward(sample, timestep, encoder_hidden_states)

    def residuals(self, res):
        """Context manager: the forwards inside add a ControlNetModel's residuals (`ControlResiduals`) to the skip tensors and the
        mid-block output (diffusers' down_block_additional_residuals / mid_block_additional_residual)."""
        unet = self

        class _Scope:
            def __enter__(s):
                if not isinstance(res, ControlResiduals):
                    raise L.CtxError("unet: additional residuals must come from contexture_nerf_amd.unet.ControlNetModel")
                L.check(unet._lib.ctx_unet_set_residuals(unet._h, L.ptr(res.buffer), float(res.scale)))
                unet._ws_key = None                # the injection needs one more skip-sized buffer: re-query the workspace

            def __exit__(s, *a):
                L.check(unet._lib.ctx_unet_set_residuals(unet._h, None, 1.0))
                unet._ws_key = None
                return False
        return _Scope()

    def set_residual_fp32(self, on=True):
        """Precision experiment: keep the residual stream in fp32 (fp16 operands and weights unchanged)."""
        L.check(self._lib.ctx_unet_set_residual_fp32(self._h, int(bool(on))))
        self._ws_key = None
        return self

    def forward_ref(self, sample, timestep, encoder_hidden_states, mode, bank=None, ref_row0=0):
        """Reference-only attention passes (src/zero123plus.py:127-237): mode 'w' parks the attn1 inputs of this forward in a
        bank (returned with the output), mode 'r' appends the parked tokens of `bank` to the self-attention K/V of the batch rows
        >= ref_row0.  -> ({'sample': out}, bank)."""
        m = {'w': 1, 'r': 2}.get(mode)
        if m is None:
            raise L.CtxError(f"unet.forward_ref: mode {mode!r} (expected 'w' or 'r')")
        x = L.f32c(sample, self.device)
        ctx = L.f32c(encoder_hidden_states, self.device)
        B, Cin, H, W = x.shape
        if Cin != self.in_channels or ctx.shape[0] != B or ctx.shape[2] != self.config['cross_attention_dim']:
            raise L.CtxError(f"unet.forward_ref: sample {tuple(x.shape)} / encoder_hidden_states {tuple(ctx.shape)} do not fit")
        Lc = ctx.shape[1]
        if m == 1:
            nb = self._lib.ctx_unet_ref_bank_bytes(self._h, B, H, W)
            if nb < 0:
                raise L.CtxError(self._lib.ctx_last_error().decode())
            if bank is None or bank.numel() < nb:
                bank = torch.empty(nb, dtype=torch.uint8, device=self.device)
        elif bank is None:
            raise L.CtxError("unet.forward_ref: mode 'r' needs the bank of a 'w' pass")
        need = self._lib.ctx_unet_workspace_bytes_ref(self._h, B, H, W, Lc, m, ref_row0 if m == 2 else 0)
        if need < 0:
            raise L.CtxError(self._lib.ctx_last_error().decode())
        if self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._bind()
        self._ws_key = None
        if self._t is None:
            self._t = torch.empty(1, dtype=torch.float32, device=self.device)
        self._t.fill_(float(timestep))
        out = torch.empty(B, self.config['out_channels'], H, W, device=self.device)
        L.check(self._lib.ctx_unet_forward_ref(self._h, L.ptr(x, torch.float32, "sample"), L.ptr(self._t), L.ptr(ctx), B, H, W, Lc, m,
                                               L.ptr(bank), ref_row0, L.ptr(out), L.stream()))
        return {'sample': out}, bank

    def forward(self, sample, timestep, encoder_hidden_states):
        x = L.f32c(sample, self.device)
        ctx = L.f32c(encoder_hidden_states, self.device)
        B, Cin, H, W = x.shape
        if Cin != self.in_channels:
            raise L.CtxError(f"unet: expected {self.in_channels} input channels, got {Cin}")
        if ctx.shape[0] != B or ctx.shape[2] != self.config['cross_attention_dim']:
            raise L.CtxError(f"unet: encoder_hidden_states shape {tuple(ctx.shape)} does not match batch {B} / dim "
                             f"{self.config['cross_attention_dim']}")
        Lc = ctx.shape[1]
        key = (B, H, W, Lc)
        if self._ws_key != key:
            need = self.workspace_bytes(B, H, W, Lc)
            if self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
                self._bind()
            self._ws_key = key
        if isinstance(timestep, torch.Tensor) and timestep.is_cuda and timestep.dtype == torch.float32 and timestep.numel() == 1:
            t = timestep.reshape(1)
        else:
            if self._t is None:
                self._t = torch.empty(1, dtype=torch.float32, device=self.device)
            self._t.fill_(float(timestep))
            t = self._t
        out = torch.empty(B, self.config['out_channels'], H, W, device=self.device)
        L.check(self._lib.ctx_unet_forward(self._h, L.ptr(x, torch.float32, "sample"), L.ptr(t), L.ptr(ctx), B, H, W, Lc,
                                           L.ptr(out), L.stream()))
        return {'sample': out}


class ControlResiduals(list):
    """What ControlNetModel returns in place of diffusers' tuple of tensors: `.buffer` holds every residual (fp16, engine layout),
    `.scale` the conditioning scale the UNet applies when it adds them; the list items are [B,h,w,C] fp16 views for inspection
    (skip order, the mid-block residual last)."""
    buffer = None
    scale = 1.0


class ControlNetModel(UNet2DConditionModel):
    """diffusers.ControlNetModel (`from_unet` topology: conditioning channels 16-32-96-256) on the HIP engine; the reference loads
    "sudo-ai/controlnet-zp11-depth-v1" with conditioning_scale=2 (src/training/trainer.py:302-304).  Call shape:
    controlnet(sample, t, encoder_hidden_states=, controlnet_cond=, conditioning_scale=, return_dict=False) -> (down, mid)."""

    def __init__(self, config=None, device="cuda:0", seed=0, init=True, conditioning_channels=3):
        self.conditioning_channels = conditioning_channels
        super().__init__(config, device=device, seed=seed, init=init)
        self._res = None
        self._cond_cache, self._cond_key = None, None      # embedding of the last conditioning image (same tensor, same version)

    def _create_handle(self):
        cfg = self.config
        c = UNetConfig(cfg['in_channels'], cfg['out_channels'], len(cfg['block_out_channels']),
                       _pad4(cfg['block_out_channels']), _pad4(cfg['heads']), _pad4(cfg['down_attn']),
                       _pad4(cfg['up_attn']), cfg['layers_per_block'], cfg['cross_attention_dim'], cfg['groups'],
                       cfg['norm_eps'])
        h = self._lib.ctx_controlnet_create(C.byref(c), self.conditioning_channels)
        if not h:
            raise L.CtxError("ctx_controlnet_create: " + self._lib.ctx_last_error().decode())
        return h

    def __call__(self, sample, timestep, encoder_hidden_states=None, controlnet_cond=None, conditioning_scale=1.0,
                 return_dict=False, **kw):
        x = L.f32c(sample, self.device)
        ctx = L.f32c(encoder_hidden_states, self.device)
        cond = L.f32c(controlnet_cond, self.device)
        B, Cin, H, W = x.shape
        if Cin != self.in_channels or ctx.shape[0] != B or tuple(cond.shape) != (B, self.conditioning_channels, 8 * H, 8 * W):
            raise L.CtxError(f"controlnet: sample {tuple(x.shape)}, encoder_hidden_states {tuple(ctx.shape)}, controlnet_cond "
                             f"{tuple(cond.shape)} do not fit (the conditioning image is 8x the latent grid)")
        Lc = ctx.shape[1]
        need = self.workspace_bytes(B, H, W, Lc)
        if self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._bind()
        nb = self._lib.ctx_controlnet_residual_bytes(self._h, B, H, W)
        if self._res is None or self._res.numel() < nb:
            self._res = torch.empty(nb, dtype=torch.uint8, device=self.device)
        if self._t is None:
            self._t = torch.empty(1, dtype=torch.float32, device=self.device)
        self._t.fill_(float(timestep))
        cb = self._lib.ctx_controlnet_cond_cache_bytes(self._h, B, H, W)
        if self._cond_cache is None or self._cond_cache.numel() < cb:
            self._cond_cache, self._cond_key = torch.empty(cb, dtype=torch.uint8, device=self.device), None
        key = (controlnet_cond.data_ptr(), controlnet_cond._version, tuple(controlnet_cond.shape))
        valid = int(key == self._cond_key)                  # the depth image does not change between the steps of one denoise
        L.check(self._lib.ctx_controlnet_forward(self._h, L.ptr(x, torch.float32, "sample"), L.ptr(self._t), L.ptr(ctx),
                                                 L.ptr(cond, torch.float32, "controlnet_cond"), L.ptr(self._cond_cache), valid, B, H, W, Lc,
                                                 L.ptr(self._res), L.stream()))
        self._cond_key = key
        out = ControlResiduals()
        out.buffer, out.scale = self._res, float(conditioning_scale)
        half = self._res.view(torch.float16)
        ch, lpb, n = self.config['block_out_channels'], self.config['layers_per_block'], len(self.config['block_out_channels'])
        shapes, h, w = [(H, W, ch[0])], H, W
        for i in range(n):
            shapes += [(h, w, ch[i])] * lpb
            if i != n - 1:
                h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
                shapes.append((h, w, ch[i]))
        shapes.append((h, w, ch[-1]))                                     # mid
        off = 0
        for (hh, ww, cc) in shapes:
            out.append(half[off:off + B * hh * ww * cc].view(B, hh, ww, cc)); off += B * hh * ww * cc
        return out, out[-1]

