"""AutoencoderKL on the HIP kernels: `vae.encode(img)['latent_dist'].mean`, `vae.decode(z)['sample']`.

Call sites: `/root/reference/p2p/inversion/ddim.py:35-41` (image2latent) and
`/root/reference/p2p/model/sd_utils.py:82-88` (latent2image).  The architecture restates diffusers'
`AutoencoderKL` [ext] for the SD1.x VAE (`vae/config.json`: block_out_channels (128,256,512,512),
layers_per_block 2, 4 latent channels, GroupNorm 32 / eps 1e-6, one single-head attention in each mid
block) with diffusers' parameter names, so a real `vae/diffusion_pytorch_model.safetensors` loads.

Layout and kernels are the UNet's: NHWC fp16 activations, implicit-GEMM 3x3 convs (`ief_conv3x3_f16`; the
encoder's stride-2 downsample uses pad (0,1,0,1) = `pad_hi_only`), fused GroupNorm+SiLU, nearest-2x
upsample fused into the following conv's gather, boundary convs (`ief_conv_in_f32` 3->C from the fp32
NCHW image, `ief_conv_out_f32` C->3 back to fp32 NCHW).  The mid-block attention has head dim 512
(outside the flash kernel's range): scores by GEMM, `ief_softmax_rows_f16`, P.V by GEMM on V^T.
`quant_conv` / `post_quant_conv` (1x1 on 8 / 4 channels) run in `ief_pointwise_f32`.
"""
from collections import OrderedDict
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Tuple

import torch

from . import hip


@dataclass(frozen=True)
class VAEConfig:
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    latent_channels: int = 4
    in_channels: int = 3
    norm_num_groups: int = 32
    eps: float = 1e-6
    scaling_factor: float = 0.18215


SD_VAE = VAEConfig()
TINY_VAE = VAEConfig(block_out_channels=(64, 64, 128, 128), layers_per_block=1)


# ------------------------------------------------------------------------------------- parameter inventory
def vae_param_shapes(cfg: VAEConfig):
    s = OrderedDict()
    ch, L = cfg.block_out_channels, cfg.layers_per_block

    def resnet(p, cin, cout):
        s[p + ".norm1.weight"] = (cin,); s[p + ".norm1.bias"] = (cin,)
        s[p + ".conv1.weight"] = (cout, cin, 3, 3); s[p + ".conv1.bias"] = (cout,)
        s[p + ".norm2.weight"] = (cout,); s[p + ".norm2.bias"] = (cout,)
        s[p + ".conv2.weight"] = (cout, cout, 3, 3); s[p + ".conv2.bias"] = (cout,)
        if cin != cout:
            s[p + ".conv_shortcut.weight"] = (cout, cin, 1, 1); s[p + ".conv_shortcut.bias"] = (cout,)

    def mid(p, c):
        resnet(p + ".resnets.0", c, c)
        a = p + ".attentions.0"
        s[a + ".group_norm.weight"] = (c,); s[a + ".group_norm.bias"] = (c,)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            s[f"{a}.{n}.weight"] = (c, c); s[f"{a}.{n}.bias"] = (c,)
        resnet(p + ".resnets.1", c, c)

    s["encoder.conv_in.weight"] = (ch[0], cfg.in_channels, 3, 3); s["encoder.conv_in.bias"] = (ch[0],)
    cout = ch[0]
    for i, c in enumerate(ch):
        cin, cout = cout, c
        for j in range(L):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i < len(ch) - 1:
            s[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            s[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (cout,)
    mid("encoder.mid_block", ch[-1])
    s["encoder.conv_norm_out.weight"] = (ch[-1],); s["encoder.conv_norm_out.bias"] = (ch[-1],)
    s["encoder.conv_out.weight"] = (2 * cfg.latent_channels, ch[-1], 3, 3); s["encoder.conv_out.bias"] = (2 * cfg.latent_channels,)
    s["quant_conv.weight"] = (2 * cfg.latent_channels, 2 * cfg.latent_channels, 1, 1); s["quant_conv.bias"] = (2 * cfg.latent_channels,)
    s["post_quant_conv.weight"] = (cfg.latent_channels, cfg.latent_channels, 1, 1); s["post_quant_conv.bias"] = (cfg.latent_channels,)
    rev = tuple(reversed(ch))
    s["decoder.conv_in.weight"] = (rev[0], cfg.latent_channels, 3, 3); s["decoder.conv_in.bias"] = (rev[0],)
    mid("decoder.mid_block", rev[0])
    cout = rev[0]
    for i, c in enumerate(rev):
        cin, cout = cout, c
        for j in range(L + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i < len(ch) - 1:
            s[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            s[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (cout,)
    s["decoder.conv_norm_out.weight"] = (ch[0],); s["decoder.conv_norm_out.bias"] = (ch[0],)
    s["decoder.conv_out.weight"] = (cfg.in_channels, ch[0], 3, 3); s["decoder.conv_out.bias"] = (cfg.in_channels,)
    return s


def synthetic_vae_state_dict(cfg: VAEConfig, seed: int = 2):
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, shape in vae_param_shapes(cfg).items():
        leaf = name.rsplit(".", 2)[-2]
        if "norm" in leaf:
            t = torch.randn(shape, generator=g) * 0.1 + (1.0 if name.endswith(".weight") else 0.0)
        elif name.endswith(".bias"):
            t = torch.randn(shape, generator=g) * 0.02
        else:
            fan = 1
            for d in shape[1:]:
                fan *= d
            t = torch.randn(shape, generator=g) / fan ** 0.5
        sd[name] = t
    return sd


# ------------------------------------------------------------------------------------- device modules
_PACK = torch.float16      # storage dtype while an AutoencoderKL is being packed (fp32 for precision="f32", as in unet.py)


def _f16(t, dev):
    return t.detach().to(device=dev, dtype=_PACK).contiguous()


def _f32(t, dev):
    return t.detach().to(device=dev, dtype=torch.float32).contiguous()


class _Resnet:
    def __init__(self, sd, p, cin, cout, G, eps, dev):
        self.G, self.eps = G, eps
        self.n1 = (_f32(sd[p + ".norm1.weight"], dev), _f32(sd[p + ".norm1.bias"], dev))
        self.n2 = (_f32(sd[p + ".norm2.weight"], dev), _f32(sd[p + ".norm2.bias"], dev))
        self.w1 = _f16(sd[p + ".conv1.weight"].permute(0, 2, 3, 1), dev)
        self.b1 = _f32(sd[p + ".conv1.bias"], dev)
        w2 = sd[p + ".conv2.weight"].permute(0, 2, 3, 1).reshape(cout, 9 * cout)
        if cin != cout:   # 1x1 shortcut folded into conv2 as an extra K range over the block input
            ws = sd[p + ".conv_shortcut.weight"].reshape(cout, cin)
            self.w2 = _f16(torch.cat([w2, ws], 1), dev)
            self.b2 = _f32(sd[p + ".conv2.bias"] + sd[p + ".conv_shortcut.bias"], dev)
            self.short = True
        else:
            self.w2 = _f16(w2.reshape(cout, 3, 3, cout), dev)
            self.b2 = _f32(sd[p + ".conv2.bias"], dev)
            self.short = False

    def __call__(self, x, cstat=None):
        """-> (out, column statistics of out for the next GroupNorm); cstat: those of x from its producer"""
        h = hip.groupnorm(x, self.n1[0], self.n1[1], self.G, self.eps, silu=True, cstat=cstat)
        h, hs = hip.conv3x3(h, self.w1, self.b1, col_stats=True)
        h = hip.groupnorm(h, self.n2[0], self.n2[1], self.G, self.eps, silu=True, cstat=hs)
        if self.short:
            return hip.conv3x3_shortcut(h, self.w2, self.b2, x, col_stats=True)
        return hip.conv3x3(h, self.w2, self.b2, residual=x, col_stats=True)


class _MidAttention:
    """single head, d = C (512): GroupNorm -> q,k,v -> softmax(q k^T / sqrt(C)) v -> out-proj + residual"""

    def __init__(self, sd, p, c, G, eps, dev):
        self.G, self.eps, self.c = G, eps, c
        self.gn = (_f32(sd[p + ".group_norm.weight"], dev), _f32(sd[p + ".group_norm.bias"], dev))
        self.wqkv = _f16(torch.cat([sd[f"{p}.to_{n}.weight"] for n in "qkv"], 0), dev)
        self.bqkv = _f32(torch.cat([sd[f"{p}.to_{n}.bias"] for n in "qkv"], 0), dev)
        self.wo, self.bo = _f16(sd[p + ".to_out.0.weight"], dev), _f32(sd[p + ".to_out.0.bias"], dev)

    def __call__(self, x, cstat=None):
        B, H, W, C = x.shape
        N = H * W
        h = hip.groupnorm(x, self.gn[0], self.gn[1], self.G, self.eps, silu=False, cstat=cstat).reshape(B, N, C)
        qkv = hip.gemm(h, self.wqkv, bias=self.bqkv)
        out = torch.empty(B, N, C, dtype=x.dtype, device=x.device)
        for b in range(B):
            q, k, v = qkv[b, :, :C], qkv[b, :, C:2 * C], qkv[b, :, 2 * C:]
            probs = hip.gemm(q, k, out_scale=C ** -0.5)                 # [N, N] scores
            hip.softmax_rows_(probs)
            hip.gemm_nt(probs, v, out=out[b])                            # P @ V
        y = hip.gemm(out, self.wo, bias=self.bo, residual=x.reshape(B, N, C))
        return y.reshape(B, H, W, C), None


class AutoencoderKL:
    def __init__(self, cfg: VAEConfig = SD_VAE, state_dict=None, device="cuda:0", seed: int = 2, precision: str = "f16"):
        global _PACK
        if precision not in ("f16", "f32", "f16x3"):
            raise ValueError('precision must be "f16", "f32" or "f16x3"')
        saved, _PACK = _PACK, (torch.float16 if precision == "f16" else torch.float32)
        try:
            self._build(cfg, state_dict, device, seed)
        finally:
            _PACK = saved
        self.precision = precision
        self.contract = "x3" if precision == "f16x3" else "f32"       # hip.f32_contraction mode (as unet.py)

    def _build(self, cfg, state_dict, device, seed):
        hip.load()
        dev = torch.device(device)
        self.cfg = cfg
        self.config = SimpleNamespace(scaling_factor=cfg.scaling_factor, latent_channels=cfg.latent_channels)
        self.dtype = torch.float32
        self.device = dev
        sd = state_dict if state_dict is not None else synthetic_vae_state_dict(cfg, seed)
        missing = [k for k in vae_param_shapes(cfg) if k not in sd]
        if missing:
            raise KeyError(f"VAE checkpoint lacks {len(missing)} tensors, e.g. {missing[:3]}")
        ch, L, G, eps = cfg.block_out_channels, cfg.layers_per_block, cfg.norm_num_groups, cfg.eps
        p3 = lambda w: _f16(w.permute(0, 2, 3, 1), dev)
        # ---- encoder
        self.e_in = (_f16(sd["encoder.conv_in.weight"].permute(2, 3, 1, 0), dev), _f32(sd["encoder.conv_in.bias"], dev))
        self.e_down = []
        cout = ch[0]
        for i, c in enumerate(ch):
            cin, cout = cout, c
            res = [_Resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout, G, eps, dev) for j in range(L)]
            ds = None
            if i < len(ch) - 1:
                q = f"encoder.down_blocks.{i}.downsamplers.0.conv"
                ds = (p3(sd[q + ".weight"]), _f32(sd[q + ".bias"], dev))
            self.e_down.append((res, ds))
        self.e_mid = (_Resnet(sd, "encoder.mid_block.resnets.0", ch[-1], ch[-1], G, eps, dev),
                      _MidAttention(sd, "encoder.mid_block.attentions.0", ch[-1], G, eps, dev),
                      _Resnet(sd, "encoder.mid_block.resnets.1", ch[-1], ch[-1], G, eps, dev))
        self.e_norm = (_f32(sd["encoder.conv_norm_out.weight"], dev), _f32(sd["encoder.conv_norm_out.bias"], dev))
        self.e_out = (p3(sd["encoder.conv_out.weight"]), _f32(sd["encoder.conv_out.bias"], dev))
        self.quant = (_f32(sd["quant_conv.weight"].reshape(2 * cfg.latent_channels, -1), dev), _f32(sd["quant_conv.bias"], dev))
        self.post_quant = (_f32(sd["post_quant_conv.weight"].reshape(cfg.latent_channels, -1), dev), _f32(sd["post_quant_conv.bias"], dev))
        # ---- decoder
        rev = tuple(reversed(ch))
        self.d_in = (_f16(sd["decoder.conv_in.weight"].permute(2, 3, 1, 0), dev), _f32(sd["decoder.conv_in.bias"], dev))
        self.d_mid = (_Resnet(sd, "decoder.mid_block.resnets.0", rev[0], rev[0], G, eps, dev),
                      _MidAttention(sd, "decoder.mid_block.attentions.0", rev[0], G, eps, dev),
                      _Resnet(sd, "decoder.mid_block.resnets.1", rev[0], rev[0], G, eps, dev))
        self.d_up = []
        cout = rev[0]
        for i, c in enumerate(rev):
            cin, cout = cout, c
            res = [_Resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout, G, eps, dev) for j in range(L + 1)]
            us = None
            if i < len(ch) - 1:
                q = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                us = (p3(sd[q + ".weight"]), _f32(sd[q + ".bias"], dev))
            self.d_up.append((res, us))
        self.d_norm = (_f32(sd["decoder.conv_norm_out.weight"], dev), _f32(sd["decoder.conv_norm_out.bias"], dev))
        self.d_out = (p3(sd["decoder.conv_out.weight"]), _f32(sd["decoder.conv_out.bias"], dev))

    def to(self, device):
        return self

    # ------------------------------------------------------------------ encode / decode
    @torch.no_grad()
    def encode(self, image):
        with hip.f32_contraction(self.contract):
            return self._encode(image)

    @torch.no_grad()
    def decode(self, z):
        with hip.f32_contraction(self.contract):
            return self._decode(z)

    def _encode(self, image):
        """image fp32 NCHW [B,3,H,W] in [-1,1] -> {'latent_dist': dist}, dist.mean [B,4,H/8,W/8] fp32"""
        x = image.to(self.device, torch.float32).contiguous()
        G, eps = self.cfg.norm_num_groups, self.cfg.eps
        h, hs = hip.conv_in(x, self.e_in[0], self.e_in[1]), None
        for res, ds in self.e_down:
            for r in res:
                h, hs = r(h, hs)
            if ds is not None:
                h, hs = hip.conv3x3(h, ds[0], ds[1], stride=2, pad_hi_only=True, col_stats=True)
        for m in self.e_mid:
            h, hs = m(h, hs)
        h = hip.groupnorm(h, self.e_norm[0], self.e_norm[1], G, eps, silu=True, cstat=hs)
        moments = hip.conv_out(h, self.e_out[0], self.e_out[1])
        moments = hip.pointwise_f32(moments, self.quant[0], self.quant[1])
        return {"latent_dist": _Dist(moments)}

    def _decode(self, z):
        """z fp32 NCHW [B,4,h,w] -> {'sample': fp32 NCHW [B,3,8h,8w]}"""
        z = z.to(self.device, torch.float32).contiguous()
        G, eps = self.cfg.norm_num_groups, self.cfg.eps
        z = hip.pointwise_f32(z, self.post_quant[0], self.post_quant[1])
        h, hs = hip.conv_in(z, self.d_in[0], self.d_in[1]), None
        for m in self.d_mid:
            h, hs = m(h, hs)
        for res, us in self.d_up:
            for r in res:
                h, hs = r(h, hs)
            if us is not None:
                h, hs = hip.conv3x3(h, us[0], us[1], upsample=True, col_stats=True)
        h = hip.groupnorm(h, self.d_norm[0], self.d_norm[1], G, eps, silu=True, cstat=hs)
        return {"sample": hip.conv_out(h, self.d_out[0], self.d_out[1])}


class _Dist:
    def __init__(self, moments):
        self.mean, self.logvar = moments.chunk(2, dim=1)

    def sample(self, generator=None):
        return self.mean

    def mode(self):
        return self.mean
