"""ORACLE (test infrastructure) — plain-PyTorch fp32 restatement of the SD2-depth
`UNet2DConditionModel` that the reference calls at src/stable_diffusion_depth.py:422-423
(`self.unet(x[2,5,h,w], t, encoder_hidden_states=ctx[2,77,1024])['sample']`).

The module graph and parameter names follow diffusers 0.27.2 (requirements.txt:13) and the
`stabilityai/stable-diffusion-2-depth/unet/config.json` hyper-parameters listed in SURVEY.md
Appendix A.5.  diffusers and the weights are NOT available offline, and the reference holds no
test vectors for this stage => PARITY UNPINNED against diffusers; the HIP engine is compared with
this module on seeded random-init weights (which is also what BASELINE.json prescribes).
"""
import math
import torch
import torch.nn as nn
import torch.nn.functional as F

SD2_DEPTH = dict(in_channels=5, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 down_attn=(True, True, True, False), up_attn=(False, True, True, True),
                 cross_attention_dim=1024, heads=(5, 10, 20, 20), groups=32, norm_eps=1e-5)


def tiny_config(ch=(64, 128), heads=(1, 2), ctx_dim=64, in_channels=5, groups=32):
    """A shrunken config (same topology rules) for fast CPU/GPU parity tests."""
    n = len(ch)
    return dict(in_channels=in_channels, out_channels=4, block_out_channels=tuple(ch), layers_per_block=2,
                down_attn=tuple([True] * (n - 1) + [False]), up_attn=tuple([False] + [True] * (n - 1)),
                cross_attention_dim=ctx_dim, heads=tuple(heads), groups=groups, norm_eps=1e-5)


def timestep_embedding(t, dim):
    """diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0)."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half
    emb = t.float()[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Attention(nn.Module):
    def __init__(self, qdim, kvdim, heads):
        super().__init__()
        self.heads = heads
        self.to_q = nn.Linear(qdim, qdim, bias=False)
        self.to_k = nn.Linear(kvdim, qdim, bias=False)
        self.to_v = nn.Linear(kvdim, qdim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(qdim, qdim)])

    def forward(self, x, ctx=None):
        ctx = x if ctx is None else ctx
        B, S, C = x.shape
        H = self.heads
        q = self.to_q(x).view(B, S, H, C // H).transpose(1, 2)
        k = self.to_k(ctx).view(B, -1, H, C // H).transpose(1, 2)
        v = self.to_v(ctx).view(B, -1, H, C // H).transpose(1, 2)
        a = torch.softmax((q @ k.transpose(-1, -2)) * (C // H) ** -0.5, dim=-1) @ v
        return self.to_out[0](a.transpose(1, 2).reshape(B, S, C))


class GEGLU(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.proj = nn.Linear(cin, cout * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Identity(), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, ctx_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, dim, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, ctx_dim, heads)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, ctx):
        n = self.norm1(x)
        st = getattr(self, '_ref_state', None)      # reference-only attention (src/zero123plus.py:127-161), see ref_only_forward
        if st is None:
            a = self.attn1(n)
        elif st['mode'] == 'w':
            st['bank'][id(self)] = n
            a = self.attn1(n)
        else:
            r0 = st['row0']
            outs = [self.attn1(n[:r0])] if r0 else []
            outs.append(self.attn1(n[r0:], torch.cat([n[r0:], st['bank'][id(self)]], dim=1)))
            a = torch.cat(outs)
        x = x + a
        x = x + self.attn2(self.norm2(x), ctx)
        return x + self.ff(self.norm3(x))


class Transformer2DModel(nn.Module):
    def __init__(self, dim, heads, ctx_dim, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, dim, eps=1e-6)
        self.proj_in = nn.Linear(dim, dim)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(dim, heads, ctx_dim)])
        self.proj_out = nn.Linear(dim, dim)

    def forward(self, x, ctx):
        B, C, H, W = x.shape
        h = self.norm(x).permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = self.proj_in(h)
        for blk in self.transformer_blocks:
            h = blk(h, ctx)
        h = self.proj_out(h).reshape(B, H, W, C).permute(0, 3, 1, 2)
        return h + x


class Sampler(nn.Module):
    def __init__(self, c, stride):
        super().__init__()
        self.stride = stride
        self.conv = nn.Conv2d(c, c, 3, stride=stride, padding=1)

    def forward(self, x):
        if self.stride == 1:
            x = F.interpolate(x, scale_factor=2.0, mode='nearest')
        return self.conv(x)


class Block(nn.Module):
    pass


class UNet2DConditionModelRef(nn.Module):
    def __init__(self, cfg=None):
        super().__init__()
        cfg = dict(SD2_DEPTH if cfg is None else cfg)
        self.cfg = cfg
        ch = cfg['block_out_channels']
        g, eps, cd = cfg['groups'], cfg['norm_eps'], cfg['cross_attention_dim']
        temb = ch[0] * 4
        self.in_channels = cfg['in_channels']
        self.conv_in = nn.Conv2d(cfg['in_channels'], ch[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(ch[0], temb)
        self.down_blocks = nn.ModuleList()
        out = ch[0]
        for i, c in enumerate(ch):
            blk = Block()
            cin, out = out, c
            blk.resnets = nn.ModuleList([ResnetBlock2D(cin if j == 0 else out, out, temb, g, eps)
                                         for j in range(cfg['layers_per_block'])])
            if cfg['down_attn'][i]:
                blk.attentions = nn.ModuleList([Transformer2DModel(out, cfg['heads'][i], cd, g)
                                                for _ in range(cfg['layers_per_block'])])
            if i != len(ch) - 1:
                blk.downsamplers = nn.ModuleList([Sampler(out, 2)])
            self.down_blocks.append(blk)
        self.mid_block = Block()
        self.mid_block.resnets = nn.ModuleList([ResnetBlock2D(ch[-1], ch[-1], temb, g, eps) for _ in range(2)])
        self.mid_block.attentions = nn.ModuleList([Transformer2DModel(ch[-1], cfg['heads'][-1], cd, g)])
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(ch)); rheads = list(reversed(cfg['heads']))
        out = rev[0]
        for i, c in enumerate(rev):
            blk = Block()
            prev, out = out, c
            inp = rev[min(i + 1, len(ch) - 1)]
            n = cfg['layers_per_block'] + 1
            blk.resnets = nn.ModuleList()
            for j in range(n):
                skip = inp if j == n - 1 else out
                rin = prev if j == 0 else out
                blk.resnets.append(ResnetBlock2D(rin + skip, out, temb, g, eps))
            if cfg['up_attn'][i]:
                blk.attentions = nn.ModuleList([Transformer2DModel(out, rheads[i], cd, g) for _ in range(n)])
            if i != len(ch) - 1:
                blk.upsamplers = nn.ModuleList([Sampler(out, 1)])
            self.up_blocks.append(blk)
        self.conv_norm_out = nn.GroupNorm(g, ch[0], eps=eps)
        self.conv_out = nn.Conv2d(ch[0], cfg['out_channels'], 3, padding=1)

    def forward(self, sample, timestep, encoder_hidden_states, down_block_additional_residuals=None,
                mid_block_additional_residual=None):
        """diffusers semantics of the ControlNet inputs: the residuals are added to the collected skip tensors after the down
        path, and to the mid-block output."""
        t = torch.as_tensor(timestep, device=sample.device).reshape(-1).expand(sample.shape[0])
        temb = self.time_embedding(timestep_embedding(t, self.cfg['block_out_channels'][0]))
        ctx = encoder_hidden_states
        h = self.conv_in(sample)
        skips = [h]
        for blk in self.down_blocks:
            for j, r in enumerate(blk.resnets):
                h = r(h, temb)
                if hasattr(blk, 'attentions'):
                    h = blk.attentions[j](h, ctx)
                skips.append(h)
            if hasattr(blk, 'downsamplers'):
                h = blk.downsamplers[0](h)
                skips.append(h)
        if down_block_additional_residuals is not None:
            skips = [a + b for a, b in zip(skips, down_block_additional_residuals)]
        h = self.mid_block.resnets[0](h, temb)
        h = self.mid_block.attentions[0](h, ctx)
        h = self.mid_block.resnets[1](h, temb)
        if mid_block_additional_residual is not None:
            h = h + mid_block_additional_residual
        for blk in self.up_blocks:
            for j, r in enumerate(blk.resnets):
                h = r(torch.cat([h, skips.pop()], dim=1), temb)
                if hasattr(blk, 'attentions'):
                    h = blk.attentions[j](h, ctx)
            if hasattr(blk, 'upsamplers'):
                h = blk.upsamplers[0](h)
        h = self.conv_out(F.silu(self.conv_norm_out(h)))
        return {'sample': h}


class ControlNetModelRef(nn.Module):
    """diffusers 0.27.2 ControlNetModel with the `from_unet` topology (PARITY UNPINNED vs diffusers, like the UNet): the UNet's
    conv_in / time_embedding / down_blocks / mid_block, ControlNetConditioningEmbedding (conv_in, [conv, conv stride 2] x 3 with
    SiLU after each, conv_out; channels 16-32-96-256) added to conv_in(sample), and one 1x1 convolution per skip tensor + one
    for the mid block; outputs multiplied by conditioning_scale.  Spec of its use: src/zero123plus.py:260-298."""

    def __init__(self, cfg, conditioning_channels=3, emb_channels=(16, 32, 96, 256)):
        super().__init__()
        u = UNet2DConditionModelRef(cfg)
        self.cfg = u.cfg
        self.conv_in, self.time_embedding, self.down_blocks, self.mid_block = u.conv_in, u.time_embedding, u.down_blocks, u.mid_block
        ch = cfg['block_out_channels']
        e = Block()
        e.conv_in = nn.Conv2d(conditioning_channels, emb_channels[0], 3, padding=1)
        e.blocks = nn.ModuleList()
        for i in range(len(emb_channels) - 1):
            e.blocks.append(nn.Conv2d(emb_channels[i], emb_channels[i], 3, padding=1))
            e.blocks.append(nn.Conv2d(emb_channels[i], emb_channels[i + 1], 3, padding=1, stride=2))
        e.conv_out = nn.Conv2d(emb_channels[-1], ch[0], 3, padding=1)
        self.controlnet_cond_embedding = e
        skc = [ch[0]]
        for i, c in enumerate(ch):
            skc += [c] * cfg['layers_per_block']
            if i != len(ch) - 1:
                skc.append(c)
        self.controlnet_down_blocks = nn.ModuleList([nn.Conv2d(c, c, 1) for c in skc])
        self.controlnet_mid_block = nn.Conv2d(ch[-1], ch[-1], 1)

    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale=1.0):
        t = torch.as_tensor(timestep, device=sample.device).reshape(-1).expand(sample.shape[0])
        temb = self.time_embedding(timestep_embedding(t, self.cfg['block_out_channels'][0]))
        e = self.controlnet_cond_embedding
        c = F.silu(e.conv_in(controlnet_cond))
        for blk in e.blocks:
            c = F.silu(blk(c))
        h = self.conv_in(sample) + e.conv_out(c)
        skips = [h]
        for blk in self.down_blocks:
            for j, r in enumerate(blk.resnets):
                h = r(h, temb)
                if hasattr(blk, 'attentions'):
                    h = blk.attentions[j](h, encoder_hidden_states)
                skips.append(h)
            if hasattr(blk, 'downsamplers'):
                h = blk.downsamplers[0](h)
                skips.append(h)
        h = self.mid_block.resnets[0](h, temb)
        h = self.mid_block.attentions[0](h, encoder_hidden_states)
        h = self.mid_block.resnets[1](h, temb)
        down = [z(s) * conditioning_scale for z, s in zip(self.controlnet_down_blocks, skips)]
        return down, self.controlnet_mid_block(h) * conditioning_scale


def randomize_affine(model, seed=0, scale=0.1):
    """Perturb norm affine parameters (default init is weight=1, bias=0, which hides bugs)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, (nn.GroupNorm, nn.LayerNorm)):
                m.weight.add_(scale * torch.randn(m.weight.shape, generator=g))
                m.bias.add_(scale * torch.randn(m.bias.shape, generator=g))
    return model


def count_flops(cfg, h, w, ctx_len=77):
    """2*MACs of all conv/linear/attention matmuls for ONE sample (SURVEY §8d definition)."""
    ch = cfg['block_out_channels']; cd = cfg['cross_attention_dim']; temb = ch[0] * 4
    fl = {'conv': 0, 'proj': 0, 'attn': 0, 'ff': 0}

    def conv(cin, cout, hh, ww, k=3):
        fl['conv'] += 2 * hh * ww * cin * cout * k * k

    def res(cin, cout, hh, ww):
        conv(cin, cout, hh, ww); conv(cout, cout, hh, ww)
        fl['proj'] += 2 * temb * cout
        if cin != cout:
            conv(cin, cout, hh, ww, 1)

    def tr(c, hh, ww):
        s = hh * ww
        fl['proj'] += 2 * s * c * c * 2            # proj_in/out
        fl['proj'] += 2 * s * c * c * 4            # attn1 q,k,v,out
        fl['proj'] += 2 * s * c * c * 2 + 2 * ctx_len * cd * c * 2   # attn2 q,out + k,v
        fl['attn'] += 4 * s * s * c + 4 * s * ctx_len * c
        fl['ff'] += 2 * s * c * 8 * c + 2 * s * 4 * c * c

    fl['proj'] += 2 * (ch[0] * temb + temb * temb)
    conv(cfg['in_channels'], ch[0], h, w)
    hh, ww, out = h, w, ch[0]
    for i, c in enumerate(ch):
        cin, out = out, c
        for j in range(cfg['layers_per_block']):
            res(cin if j == 0 else out, out, hh, ww)
            if cfg['down_attn'][i]:
                tr(out, hh, ww)
        if i != len(ch) - 1:
            hh, ww = (hh + 1) // 2, (ww + 1) // 2
            conv(out, out, hh, ww)
    res(ch[-1], ch[-1], hh, ww); tr(ch[-1], hh, ww); res(ch[-1], ch[-1], hh, ww)
    rev = list(reversed(ch)); out = rev[0]
    for i, c in enumerate(rev):
        prev, out = out, c
        inp = rev[min(i + 1, len(ch) - 1)]
        n = cfg['layers_per_block'] + 1
        for j in range(n):
            res((prev if j == 0 else out) + (inp if j == n - 1 else out), out, hh, ww)
            if cfg['up_attn'][i]:
                tr(out, hh, ww)
        if i != len(ch) - 1:
            hh, ww = hh * 2, ww * 2
            conv(out, out, hh, ww)
    conv(ch[0], cfg['out_channels'], hh, ww)
    fl['total'] = sum(fl.values())
    return fl


# ---- fp16-storage restatement ---------------------------------------------------------------------------------
# The reference runs this UNet under torch.autocast('cuda') (src/stable_diffusion_depth.py:330): conv / linear /
# attention operands and results are fp16, accumulation fp32.  `forward_fp16_storage` restates that precision
# contract on the CPU: fp32 arithmetic with a round-to-fp16 at every tensor an fp16 implementation stores
# (weights included).  It is what "within 1e-3 rel fp16" is measured against; the pure-fp32 forward above is kept as
# the second yardstick.  Measured on random-init configs: fp16 weights alone put ANY fp16 implementation 0.9e-3 away
# from the fp32 result, this restatement 1.5e-3 away.
def _h(x):
    return x.half().float()


class _PermLinear:
    """y = x W^T + b with the K sum taken in `chunks` pieces added in REVERSE order: the same real-number result through a
    different fp32 accumulation order (the "two correct implementations" experiment of tests / tools)."""

    def __init__(self, chunks=4):
        self.chunks = chunks

    def linear(self, m, x):
        K = x.shape[-1]
        step = -(-K // self.chunks)
        acc = None
        for k0 in reversed(range(0, K, step)):
            part = x[..., k0:k0 + step] @ m.weight[:, k0:k0 + step].T
            acc = part if acc is None else acc + part
        return acc if m.bias is None else acc + m.bias

    def conv(self, m, x):
        C = x.shape[1]
        step = -(-C // self.chunks)
        acc = None
        for c0 in reversed(range(0, C, step)):
            part = F.conv2d(x[:, c0:c0 + step], m.weight[:, c0:c0 + step], None, m.stride, m.padding)
            acc = part if acc is None else acc + part
        return acc if m.bias is None else acc + m.bias[None, :, None, None]


def forward_fp16_storage(model, sample, timestep, ctx, q=_h, q_res=_h, q_w=_h, taps=None, p16=False, temb16=False, autocast=False,
                         perm=None):
    """The precision contract restated (see above).  Knobs, all off by default (= one rounding per FUSED operation, which is what
    a fused implementation stores):
      p16      softmax probabilities rounded to fp16 before the P V product (flash / SDPA kernels do);
      temb16   SiLU(temb) and each ResBlock's time_emb_proj output rounded (they are fp16 tensors under autocast);
      autocast EVERY op-level rounding torch.autocast would make in diffusers' eager graph: each conv / linear output, each
               residual / bias-free add, GEGLU's two halves, gelu(gate) and their product, proj_out before the skip add;
               implies p16 and temb16.  This is the reference's literal contract (src/stable_diffusion_depth.py:330);
      perm     a _PermLinear: every matmul / conv accumulates its K sum in another order (identical real-number result).
    taps: a list that receives every block's output (NCHW), in the engine's tap order."""
    p16, temb16 = p16 or autocast, temb16 or autocast
    qa = q if autocast else (lambda v: v)                       # op-level roundings that a fused kernel does not make

    def lin(m, x):
        return perm.linear(m, x) if perm is not None else m(x)

    def cv(m, x):
        return perm.conv(m, x) if perm is not None else m(x)

    def tap(x):
        if taps is not None:
            taps.append(x)
        return x

    def res_fwd(m, x, temb_act):
        t = q(F.silu(m.norm1(x)))
        tp = lin(m.time_emb_proj, temb_act)
        tp = q(tp) if temb16 else tp
        hh = q(qa(cv(m.conv1, t)) + tp[:, :, None, None])
        t2 = q(F.silu(m.norm2(hh)))
        sc = x if m.conv_shortcut is None else qa(cv(m.conv_shortcut, q(x)))
        return q_res(sc + qa(cv(m.conv2, t2)))

    def attn_fwd(m, x, c=None):
        c = x if c is None else c
        B, S, C = x.shape
        H = m.heads
        qq = q(lin(m.to_q, x)).view(B, S, H, C // H).transpose(1, 2)
        k = q(lin(m.to_k, c)).view(B, -1, H, C // H).transpose(1, 2)
        v = q(lin(m.to_v, c)).view(B, -1, H, C // H).transpose(1, 2)
        sc = (qq @ k.transpose(-1, -2)) * (C // H) ** -0.5
        if p16:
            # a flash kernel rounds the UNNORMALISED probabilities exp(s - max) to fp16 for the P V product and divides by the
            # fp32 row sum afterwards
            e = torch.exp(sc - sc.amax(-1, keepdim=True))
            a = (q(e) @ v) / e.sum(-1, keepdim=True)
        else:
            a = torch.softmax(sc, -1) @ v
        return qa(lin(m.to_out[0], q(a.transpose(1, 2).reshape(B, S, C))))

    def blk_fwd(m, x, c):
        x = q_res(x + attn_fwd(m.attn1, q(m.norm1(x))))
        x = q_res(x + attn_fwd(m.attn2, q(m.norm2(x)), q(c)))
        a, g = qa(lin(m.ff.net[0].proj, q(m.norm3(x)))).chunk(2, -1)
        return q_res(x + qa(lin(m.ff.net[2], q(a * qa(F.gelu(g))))))

    def tr_fwd(m, x, c):
        B, C, H, W = x.shape
        hh = q(m.norm(x)).permute(0, 2, 3, 1).reshape(B, H * W, C)
        hh = q_res(lin(m.proj_in, hh))
        for b in m.transformer_blocks:
            hh = blk_fwd(b, hh, c)
        hh = qa(lin(m.proj_out, q(hh))).reshape(B, H, W, C).permute(0, 3, 1, 2)
        return q_res(hh + x)

    import copy
    mq = copy.deepcopy(model)
    with torch.no_grad():
        for p in mq.parameters():
            p.copy_(q_w(p))
        tt = torch.as_tensor(timestep).reshape(-1).expand(sample.shape[0])
        te = q(timestep_embedding(tt, mq.cfg['block_out_channels'][0]))
        temb = q(lin(mq.time_embedding.linear_2, q(F.silu(lin(mq.time_embedding.linear_1, te)))))
        temb = F.silu(temb)
        temb = q(temb) if temb16 else temb
        hh = tap(q_res(cv(mq.conv_in, q(sample))))
        skips = [hh]
        for blk in mq.down_blocks:
            for j, r in enumerate(blk.resnets):
                hh = tap(res_fwd(r, hh, temb))
                if hasattr(blk, 'attentions'):
                    hh = tap(tr_fwd(blk.attentions[j], hh, ctx))
                skips.append(hh)
            if hasattr(blk, 'downsamplers'):
                hh = tap(q_res(cv(blk.downsamplers[0].conv, q(hh))))
                skips.append(hh)
        hh = tap(res_fwd(mq.mid_block.resnets[0], hh, temb))
        hh = tap(tr_fwd(mq.mid_block.attentions[0], hh, ctx))
        hh = tap(res_fwd(mq.mid_block.resnets[1], hh, temb))
        for blk in mq.up_blocks:
            for j, r in enumerate(blk.resnets):
                hh = tap(res_fwd(r, torch.cat([hh, skips.pop()], 1), temb))
                if hasattr(blk, 'attentions'):
                    hh = tap(tr_fwd(blk.attentions[j], hh, ctx))
            if hasattr(blk, 'upsamplers'):
                hh = tap(q_res(cv(blk.upsamplers[0].conv, F.interpolate(q(hh), scale_factor=2.0, mode='nearest'))))
        return {'sample': cv(mq.conv_out, q(F.silu(mq.conv_norm_out(hh))))}


def forward_taps(model, sample, timestep, ctx):
    """The fp32 forward with every block's output collected in the engine's tap order -> (out, [taps])."""
    taps = []
    ident = lambda v: v
    out = forward_fp16_storage(model, sample, timestep, ctx, q=ident, q_res=ident, q_w=ident, taps=taps)
    return out, taps


def ref_only_forward(model, sample, timestep, ctx, noisy_cond_lat, is_cfg_guidance, down_res=None, mid_res=None):
    """RefOnlyNoisedUNet.forward of src/zero123plus.py:164-237 on the fp32 oracle UNet (the condition latent arrives already
    noised: the noise draw and the scheduler are the caller's): 'w' pass over the condition parks each attn1's input
    (with is_cfg_guidance only the conditional context row is used), 'r' pass appends them along the token axis to the K/V
    source of the same attn1, the unconditional batch row 0 attending without them.  down_res / mid_res: ControlNet residuals
    injected into the 'r' pass."""
    blocks = [m for m in model.modules() if isinstance(m, BasicTransformerBlock)]
    st = {'mode': 'w', 'bank': {}, 'row0': 0}
    for b in blocks:
        b._ref_state = st
    try:
        model(noisy_cond_lat, timestep, ctx[1:] if is_cfg_guidance else ctx)
        st['mode'] = 'r'; st['row0'] = 1 if is_cfg_guidance else 0
        # ControlNet residuals go to the main ('r') pass only (DepthControlUNet -> RefOnlyNoisedUNet.forward, spec :205-237, 260-298)
        if down_res is not None:
            out = model(sample, timestep, ctx, down_block_additional_residuals=down_res, mid_block_additional_residual=mid_res)
        else:
            out = model(sample, timestep, ctx)
    finally:
        for b in blocks:
            del b._ref_state
    return out
