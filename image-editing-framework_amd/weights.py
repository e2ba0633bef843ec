"""UNet parameter inventory in diffusers' key naming, and seeded synthetic weights.

A real SD checkpoint (``unet/diffusion_pytorch_model.safetensors`` under a local dir named in
``sd_mapping.py``) uses exactly these keys, so the same loader packs either.  There are no
checkpoints in this environment, so tests / smoke / bench draw the tensors
from a CPU generator: std = 1/sqrt(fan_in) for conv / linear weights, small biases, norm
affines near (1, 0) (SURVEY.md §8d "Synthetic inputs").
"""
from collections import OrderedDict
import math

import torch

from .config import UNetConfig


def _resnet(shapes, p, cin, cout, temb):
    shapes[p + ".norm1.weight"] = (cin,)
    shapes[p + ".norm1.bias"] = (cin,)
    shapes[p + ".conv1.weight"] = (cout, cin, 3, 3)
    shapes[p + ".conv1.bias"] = (cout,)
    shapes[p + ".time_emb_proj.weight"] = (cout, temb)
    shapes[p + ".time_emb_proj.bias"] = (cout,)
    shapes[p + ".norm2.weight"] = (cout,)
    shapes[p + ".norm2.bias"] = (cout,)
    shapes[p + ".conv2.weight"] = (cout, cout, 3, 3)
    shapes[p + ".conv2.bias"] = (cout,)
    if cin != cout:
        shapes[p + ".conv_shortcut.weight"] = (cout, cin, 1, 1)
        shapes[p + ".conv_shortcut.bias"] = (cout,)


def _transformer(shapes, p, c, ctx, linear=False, depth=1):
    shapes[p + ".norm.weight"] = (c,)
    shapes[p + ".norm.bias"] = (c,)
    shapes[p + ".proj_in.weight"] = (c, c) if linear else (c, c, 1, 1)
    shapes[p + ".proj_in.bias"] = (c,)
    for k in range(depth):
        b = f"{p}.transformer_blocks.{k}"
        for n in ("norm1", "norm2", "norm3"):
            shapes[f"{b}.{n}.weight"] = (c,)
            shapes[f"{b}.{n}.bias"] = (c,)
        for a, kdim in (("attn1", c), ("attn2", ctx)):
            shapes[f"{b}.{a}.to_q.weight"] = (c, c)
            shapes[f"{b}.{a}.to_k.weight"] = (c, kdim)
            shapes[f"{b}.{a}.to_v.weight"] = (c, kdim)
            shapes[f"{b}.{a}.to_out.0.weight"] = (c, c)
            shapes[f"{b}.{a}.to_out.0.bias"] = (c,)
        shapes[f"{b}.ff.net.0.proj.weight"] = (8 * c, c)
        shapes[f"{b}.ff.net.0.proj.bias"] = (8 * c,)
        shapes[f"{b}.ff.net.2.weight"] = (c, 4 * c)
        shapes[f"{b}.ff.net.2.bias"] = (c,)
    shapes[p + ".proj_out.weight"] = (c, c) if linear else (c, c, 1, 1)
    shapes[p + ".proj_out.bias"] = (c,)


def unet_param_shapes(cfg: UNetConfig) -> "OrderedDict[str, tuple]":
    """name -> shape, torch layouts (conv OIHW, linear [out, in])."""
    s = OrderedDict()
    ch = cfg.block_out_channels
    temb = cfg.time_embed_dim
    nlev = len(ch)
    s["conv_in.weight"] = (ch[0], cfg.in_channels, 3, 3)
    s["conv_in.bias"] = (ch[0],)
    s["time_embedding.linear_1.weight"] = (temb, ch[0])
    s["time_embedding.linear_1.bias"] = (temb,)
    s["time_embedding.linear_2.weight"] = (temb, temb)
    s["time_embedding.linear_2.bias"] = (temb,)
    if cfg.addition_embed:
        s["add_embedding.linear_1.weight"] = (temb, cfg.addition_input_dim)
        s["add_embedding.linear_1.bias"] = (temb,)
        s["add_embedding.linear_2.weight"] = (temb, temb)
        s["add_embedding.linear_2.bias"] = (temb,)
    cout = ch[0]
    for i in range(nlev):
        cin, cout = cout, ch[i]
        for j in range(cfg.layers_per_block):
            _resnet(s, f"down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout, temb)
            if cfg.down_has_attn[i]:
                _transformer(s, f"down_blocks.{i}.attentions.{j}", cout, cfg.cross_attention_dim, cfg.use_linear_projection,
                             cfg.depth(i))
        if i < nlev - 1:
            s[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            s[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (cout,)
    cm = ch[-1]
    _resnet(s, "mid_block.resnets.0", cm, cm, temb)
    _transformer(s, "mid_block.attentions.0", cm, cfg.cross_attention_dim, cfg.use_linear_projection, cfg.depth(nlev - 1))
    _resnet(s, "mid_block.resnets.1", cm, cm, temb)
    rev = tuple(reversed(ch))
    rev_attn = tuple(reversed(cfg.down_has_attn))
    out_c = rev[0]
    for i in range(nlev):
        prev, out_c = out_c, rev[i]
        in_c = rev[min(i + 1, nlev - 1)]
        for j in range(cfg.layers_per_block + 1):
            skip = in_c if j == cfg.layers_per_block else out_c
            rin = prev if j == 0 else out_c
            _resnet(s, f"up_blocks.{i}.resnets.{j}", rin + skip, out_c, temb)
            if rev_attn[i]:
                _transformer(s, f"up_blocks.{i}.attentions.{j}", out_c, cfg.cross_attention_dim, cfg.use_linear_projection,
                             cfg.depth(nlev - 1 - i))
        if i < nlev - 1:
            s[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (out_c, out_c, 3, 3)
            s[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (out_c,)
    s["conv_norm_out.weight"] = (ch[0],)
    s["conv_norm_out.bias"] = (ch[0],)
    s["conv_out.weight"] = (cfg.out_channels, ch[0], 3, 3)
    s["conv_out.bias"] = (cfg.out_channels,)
    return s


def _is_norm(name: str) -> bool:
    leaf = name.rsplit(".", 2)[-2]
    return leaf.startswith("norm") or leaf == "conv_norm_out"


def synthetic_state_dict(cfg: UNetConfig, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """fp32 CPU tensors, deterministic in (cfg, seed); independent of thread count: tensor i draws from its own
    generator seeded with (seed, i), so the 0.86 - 2.6 G normals of a full-size UNet are drawn on all cores at once."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    shapes = unet_param_shapes(cfg)
    names = list(shapes)

    def make(i):
        name, shape = names[i], shapes[names[i]]
        g = torch.Generator(device="cpu")
        g.manual_seed((int(seed) << 24) + i)
        if _is_norm(name):
            t = torch.randn(shape, generator=g) * 0.1
            if name.endswith(".weight"):
                t = t + 1.0
        elif name.endswith(".bias"):
            t = torch.randn(shape, generator=g) * 0.02
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        return t

    with ThreadPoolExecutor(max_workers=max(1, min(32, os.cpu_count() or 1))) as ex:
        tensors = list(ex.map(make, range(len(names))))
    return OrderedDict(zip(names, tensors))


def num_params(cfg: UNetConfig) -> int:
    n = 0
    for shape in unet_param_shapes(cfg).values():
        k = 1
        for d in shape:
            k *= d
        n += k
    return n
