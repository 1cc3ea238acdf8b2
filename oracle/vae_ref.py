"""ORACLE (test infrastructure) — plain-PyTorch fp32 restatement of diffusers 0.27.2 `AutoencoderKL.decode`
(post_quant_conv + Decoder: conv_in, UNetMidBlock2D with one single-head attention, 4 UpDecoderBlock2D, GroupNorm-SiLU-conv_out)
and `AutoencoderKL.encode` (Encoder: conv_in, 4 DownEncoderBlock2D whose Downsample2D is F.pad(x,(0,1,0,1)) + stride-2 conv with
padding 0, the same mid block, GroupNorm-SiLU-conv_out to 2L channels; quant_conv; moments = mean | logvar)
as called by src/stable_diffusion_depth.py:971-990.  diffusers / weights are absent offline => PARITY UNPINNED vs diffusers;
state_dict keys follow diffusers so a real checkpoint would load."""
import torch
import torch.nn as nn
import torch.nn.functional as F

SD_VAE = dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2, groups=32)


class Res(nn.Module):
    def __init__(self, cin, cout, g):
        super().__init__()
        self.norm1 = nn.GroupNorm(g, cin, eps=1e-6); self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(g, cout, eps=1e-6); self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv2(F.silu(self.norm2(self.conv1(F.silu(self.norm1(x))))))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class Attn(nn.Module):
    def __init__(self, c, g):
        super().__init__()
        self.group_norm = nn.GroupNorm(g, c, eps=1e-6)
        self.to_q = nn.Linear(c, c); self.to_k = nn.Linear(c, c); self.to_v = nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c)])

    def forward(self, x):
        B, C, H, W = x.shape
        h = self.group_norm(x).reshape(B, C, H * W).transpose(1, 2)
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        a = torch.softmax(q @ k.transpose(1, 2) * C ** -0.5, -1) @ v
        return x + self.to_out[0](a).transpose(1, 2).reshape(B, C, H, W)


class _B(nn.Module):
    pass


class DecoderRef(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        ch, g, n = cfg['block_out_channels'], cfg['groups'], len(cfg['block_out_channels'])
        top = ch[-1]
        self.conv_in = nn.Conv2d(cfg['latent_channels'], top, 3, padding=1)
        self.mid_block = _B()
        self.mid_block.resnets = nn.ModuleList([Res(top, top, g), Res(top, top, g)])
        self.mid_block.attentions = nn.ModuleList([Attn(top, g)])
        self.up_blocks = nn.ModuleList()
        out = top
        for i in range(n):
            prev, out = out, ch[n - 1 - i]
            b = _B()
            b.resnets = nn.ModuleList([Res(prev if j == 0 else out, out, g) for j in range(cfg['layers_per_block'] + 1)])
            if i != n - 1:
                up = _B(); up.conv = nn.Conv2d(out, out, 3, padding=1)
                b.upsamplers = nn.ModuleList([up])
            self.up_blocks.append(b)
        self.conv_norm_out = nn.GroupNorm(g, ch[0], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[0], cfg['out_channels'], 3, padding=1)

    def forward(self, z):
        h = self.conv_in(z)
        h = self.mid_block.resnets[1](self.mid_block.attentions[0](self.mid_block.resnets[0](h)))
        for b in self.up_blocks:
            for r in b.resnets:
                h = r(h)
            if hasattr(b, 'upsamplers'):
                h = b.upsamplers[0].conv(F.interpolate(h, scale_factor=2.0, mode='nearest'))
        return self.conv_out(F.silu(self.conv_norm_out(h)))


class EncoderRef(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        ch, g, n = cfg['block_out_channels'], cfg['groups'], len(cfg['block_out_channels'])
        top = ch[-1]
        self.conv_in = nn.Conv2d(cfg['out_channels'], ch[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        cur = ch[0]
        for i in range(n):
            b = _B()
            b.resnets = nn.ModuleList([Res(cur if j == 0 else ch[i], ch[i], g) for j in range(cfg['layers_per_block'])])
            cur = ch[i]
            if i != n - 1:
                dn = _B(); dn.conv = nn.Conv2d(cur, cur, 3, stride=2, padding=0)
                b.downsamplers = nn.ModuleList([dn])
            self.down_blocks.append(b)
        self.mid_block = _B()
        self.mid_block.resnets = nn.ModuleList([Res(top, top, g), Res(top, top, g)])
        self.mid_block.attentions = nn.ModuleList([Attn(top, g)])
        self.conv_norm_out = nn.GroupNorm(g, top, eps=1e-6)
        self.conv_out = nn.Conv2d(top, 2 * cfg['latent_channels'], 3, padding=1)

    def forward(self, x):
        h = self.conv_in(x)
        for b in self.down_blocks:
            for r in b.resnets:
                h = r(h)
            if hasattr(b, 'downsamplers'):
                h = b.downsamplers[0].conv(F.pad(h, (0, 1, 0, 1)))
        h = self.mid_block.resnets[1](self.mid_block.attentions[0](self.mid_block.resnets[0](h)))
        return self.conv_out(F.silu(self.conv_norm_out(h)))


class AutoencoderKLRef(nn.Module):
    """decode and encode (moments) with diffusers' key names."""

    def __init__(self, cfg=None):
        super().__init__()
        cfg = dict(SD_VAE if cfg is None else cfg)
        self.cfg = cfg
        self.post_quant_conv = nn.Conv2d(cfg['latent_channels'], cfg['latent_channels'], 1)
        self.decoder = DecoderRef(cfg)
        self.encoder = EncoderRef(cfg)
        self.quant_conv = nn.Conv2d(2 * cfg['latent_channels'], 2 * cfg['latent_channels'], 1)

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))

    def encode_moments(self, x):
        return self.quant_conv(self.encoder(x))


class AutoencoderKLDecodeRef(nn.Module):
    def __init__(self, cfg=None):
        super().__init__()
        cfg = dict(SD_VAE if cfg is None else cfg)
        self.cfg = cfg
        self.post_quant_conv = nn.Conv2d(cfg['latent_channels'], cfg['latent_channels'], 1)
        self.decoder = DecoderRef(cfg)

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))
