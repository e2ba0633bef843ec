"""ORACLE (test infrastructure only — never imported by the product path).

Eager-PyTorch fp32 CPU restatement of the per-step compute the reference delegates to
diffusers 0.27 (`UNet2DConditionModel.forward`, un-vendored third-party code pinned at
`/root/reference/requirements.txt:2`), with the reference's own execution semantics for the
hooked attention: probabilities are MATERIALISED and handed to a Python controller once per
Attention module, exactly as `/root/reference/p2p/model/register.py:11-64` does.

Parity status: the attention dataflow follows register.py:33-62 line by line; the ResNet
dataflow follows the copy of `ResnetBlock2D.forward` spelled out inside the reference at
`/root/reference/pnp/model/register.py:102-175`; the block topology restates the public SD1.5
`unet/config.json`.  The reference holds NO tests or golden vectors for
UNet numerics and diffusers is not installed here, so this part of the oracle is
"parity unpinned" against diffusers itself; what IS pinned (controllers, aligner, DDIM
reverse, NTI loop) is pinned by `tests/golden/` fixtures generated from the reference's own
modules (`tests/golden/make_golden.py`).

Layout here is the reference's: NCHW activations, fp32, weights as a diffusers-keyed
state dict (conv OIHW, linear [out, in]).
"""
import math
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- embeddings
def timestep_embedding(timesteps: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers `Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)` [ext]."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = timesteps.float()[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


# --------------------------------------------------------------------------- blocks
def resnet_block(sd: Dict[str, torch.Tensor], p: str, x, temb, groups: int, eps: float,
                 inject: Optional[Callable] = None):
    """`/root/reference/pnp/model/register.py:102-175` (the non-injected dataflow)."""
    h = F.group_norm(x, groups, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps)
    h = F.silu(h)
    h = F.conv2d(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    t = F.linear(F.silu(temb), sd[p + ".time_emb_proj.weight"], sd[p + ".time_emb_proj.bias"])
    h = h + t[:, :, None, None]
    h = F.group_norm(h, groups, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps)
    h = F.silu(h)
    h = F.conv2d(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if inject is not None:
        h = inject(h)
    if (p + ".conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return x + h


def attention(sd, p: str, x, ctx, heads: int, hook, place: str, qkv_hook=None, qkv_path_hook=None):
    """`/root/reference/p2p/model/register.py:33-62`; x [B,N,C], ctx [B,L,Cc] or None."""
    is_cross = ctx is not None
    B, N, C = x.shape
    q = F.linear(x, sd[p + ".to_q.weight"])
    src = ctx if is_cross else x
    k = F.linear(src, sd[p + ".to_k.weight"])
    v = F.linear(src, sd[p + ".to_v.weight"])
    d = C // heads

    def h2b(t):  # head_to_batch_dim [ext]: [B,L,h*d] -> [B*h, L, d], batch-major then head
        L = t.shape[1]
        return t.reshape(B, L, heads, d).permute(0, 2, 1, 3).reshape(B * heads, L, d)

    q, k, v = h2b(q), h2b(k), h2b(v)
    if qkv_hook is not None:  # MasaCtrl style editors act on q,k,v
        q, k, v = qkv_hook(q, k, v, is_cross, place, heads)
    if qkv_path_hook is not None:  # PnP patches individual modules: the hook is told WHICH one (its parameter prefix)
        q, k, v = qkv_path_hook(q, k, v, is_cross, p, heads)
    scale = d ** -0.5
    probs = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * scale, dim=-1)
    if hook is not None:
        probs = hook(probs, is_cross, place)
    o = torch.bmm(probs, v)
    o = o.reshape(B, heads, N, d).permute(0, 2, 1, 3).reshape(B, N, C)
    return F.linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])


def transformer(sd, p: str, x, ctx, heads: int, groups: int, hook, place: str, qkv_hook=None, qkv_path_hook=None):
    B, C, H, W = x.shape
    res = x
    h = F.group_norm(x, groups, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)
    if sd[p + ".proj_in.weight"].dim() == 2:   # SD2.x: use_linear_projection (reshape to tokens, then nn.Linear)
        h = F.linear(h.permute(0, 2, 3, 1).reshape(B, H * W, C), sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
    else:
        h = F.conv2d(h, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    k = 0
    while f"{p}.transformer_blocks.{k}.norm1.weight" in sd:       # SDXL: several BasicTransformerBlocks per Transformer2DModel
        b = f"{p}.transformer_blocks.{k}"
        n = F.layer_norm(h, (C,), sd[b + ".norm1.weight"], sd[b + ".norm1.bias"], 1e-5)
        h = h + attention(sd, b + ".attn1", n, None, heads, hook, place, qkv_hook, qkv_path_hook)
        n = F.layer_norm(h, (C,), sd[b + ".norm2.weight"], sd[b + ".norm2.bias"], 1e-5)
        h = h + attention(sd, b + ".attn2", n, ctx, heads, hook, place, qkv_hook, qkv_path_hook)
        n = F.layer_norm(h, (C,), sd[b + ".norm3.weight"], sd[b + ".norm3.bias"], 1e-5)
        g = F.linear(n, sd[b + ".ff.net.0.proj.weight"], sd[b + ".ff.net.0.proj.bias"])
        hid, gate = g.chunk(2, dim=-1)
        g = hid * F.gelu(gate)
        h = h + F.linear(g, sd[b + ".ff.net.2.weight"], sd[b + ".ff.net.2.bias"])
        k += 1
    if sd[p + ".proj_out.weight"].dim() == 2:
        h = F.linear(h, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"]).reshape(B, H, W, C).permute(0, 3, 1, 2)
    else:
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        h = F.conv2d(h, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return h + res


# --------------------------------------------------------------------------- whole net
def unet_forward(sd, cfg, sample, timestep, ctx, hook=None, qkv_hook=None, taps=None, qkv_path_hook=None, res_inject=None,
                 added_cond_kwargs=None):
    """sample [B,4,H,W] fp32, timestep scalar/int tensor, ctx [B,77,Cc] -> eps [B,4,H,W].

    `hook(probs, is_cross, place)` is called once per Attention module in module-tree
    order down -> mid -> up (`/root/reference/p2p/model/register.py:88-96`; diffusers registers
    down_blocks before up_blocks before mid_block as attributes, but the reference iterates
    `named_children()` and the controller only counts calls, and the forward itself always
    executes down, mid, up).
    `qkv_path_hook(q, k, v, is_cross, prefix, heads)` / `res_inject = {resnet prefix: fn(conv2 output)}` are the
    per-module patches of Plug-and-Play (`/root/reference/pnp/model/register.py:27-90,100-182`).
    """
    res_inject = res_inject or {}
    ch = cfg.block_out_channels
    nlev = len(ch)
    G, eps = cfg.norm_num_groups, cfg.norm_eps
    B = sample.shape[0]
    t = torch.as_tensor(timestep).reshape(-1).float().expand(B)
    temb = timestep_embedding(t, ch[0])
    temb = F.linear(temb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    temb = F.linear(F.silu(temb), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
    if "add_embedding.linear_1.weight" in sd:
        # SDXL `addition_embed_type == "text_time"` (diffusers UNet2DConditionModel.get_aug_embed [ext]): the 6 time ids are
        # embedded like timesteps (dim 256 each), flattened, appended to the pooled text embedding, and run through a
        # TimestepEmbedding MLP; the result is added to the time embedding
        # (`/root/reference/pix2pix-zero/model/sd_utils.py:408-421` builds the kwargs)
        text_embeds, time_ids = added_cond_kwargs["text_embeds"].float(), added_cond_kwargs["time_ids"].float()
        dim = (sd["add_embedding.linear_1.weight"].shape[1] - text_embeds.shape[-1]) // time_ids.shape[-1]
        time_embeds = timestep_embedding(time_ids.flatten(), dim).reshape(B, -1)
        aug = torch.cat([text_embeds, time_embeds], dim=-1)
        aug = F.linear(aug, sd["add_embedding.linear_1.weight"], sd["add_embedding.linear_1.bias"])
        aug = F.linear(F.silu(aug), sd["add_embedding.linear_2.weight"], sd["add_embedding.linear_2.bias"])
        temb = temb + aug

    def tap(name, v):
        if taps is not None:
            taps[name] = v

    x = F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    tap("conv_in", x)
    skips = [x]
    for i in range(nlev):
        for j in range(cfg.layers_per_block):
            x = resnet_block(sd, f"down_blocks.{i}.resnets.{j}", x, temb, G, eps)
            if cfg.down_has_attn[i]:
                x = transformer(sd, f"down_blocks.{i}.attentions.{j}", x, ctx, cfg.num_heads[i],
                                G, hook, "down", qkv_hook, qkv_path_hook)
            skips.append(x)
        if i < nlev - 1:
            p = f"down_blocks.{i}.downsamplers.0.conv"
            x = F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=2, padding=1)
            skips.append(x)
        tap(f"down{i}", x)
    x = resnet_block(sd, "mid_block.resnets.0", x, temb, G, eps)
    x = transformer(sd, "mid_block.attentions.0", x, ctx, cfg.num_heads[-1], G, hook, "mid", qkv_hook, qkv_path_hook)
    x = resnet_block(sd, "mid_block.resnets.1", x, temb, G, eps)
    tap("mid", x)
    rev_attn = tuple(reversed(cfg.down_has_attn))
    rev_heads = tuple(reversed(cfg.num_heads))
    for i in range(nlev):
        for j in range(cfg.layers_per_block + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet_block(sd, f"up_blocks.{i}.resnets.{j}", x, temb, G, eps, res_inject.get(f"up_blocks.{i}.resnets.{j}"))
            if rev_attn[i]:
                x = transformer(sd, f"up_blocks.{i}.attentions.{j}", x, ctx, rev_heads[i],
                                G, hook, "up", qkv_hook, qkv_path_hook)
        if i < nlev - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            p = f"up_blocks.{i}.upsamplers.0.conv"
            x = F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=1)
        tap(f"up{i}", x)
    x = F.group_norm(x, G, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], eps)
    x = F.silu(x)
    return F.conv2d(x, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


def count_attention_layers(cfg) -> int:
    nlev = len(cfg.block_out_channels)
    n_tr = sum(cfg.layers_per_block * cfg.depth(i) for i, a in enumerate(cfg.down_has_attn) if a) + cfg.depth(nlev - 1)
    n_tr += sum((cfg.layers_per_block + 1) * cfg.depth(i) for i, a in enumerate(cfg.down_has_attn) if a)
    return 2 * n_tr
