"""ORACLE (test infrastructure only — never imported by the product path): Plug-and-Play on the fp32 CPU UNet.

Restates `/root/reference/pnp/model/register.py` and the loop of `/root/reference/pnp/model/sd_utils.py:93-113`:
  - self-attention of `up_blocks[res].attentions[block]` for res, block in {1: [1, 2], 2: [0, 1, 2], 3: [0, 1, 2]}
    (:84): when the timestep is in the schedule, rows of batch blocks 1 and 3 take the Q and K of block 2 (:45-52);
  - `up_blocks[1].resnets[1]`: conv2's output rows of blocks 1 and 3 are overwritten with block 2's (:161-166);
  - batch = [uncond_src, uncond_tgt, cond_src, cond_tgt], CFG, DDIM step (`sd_utils.py:96-110`).
  - SDXL family (`register_*_xl`, :188-364; `PnP_XL`, `sd_utils.py:130-258`): self-attention of EVERY transformer block of
    `up_blocks[1].attentions[0..2]`, conv2 output of `up_blocks[1].resnets[0]`, and `added_cond_kwargs` on every UNet call.
Parity unpinned: the reference module imports diffusers (not installed), so no fixture could be generated from it;
the rules above are read off its source.
"""
import torch

from . import unet_ref
from .p2p_ref import DDIMRef

QK_BLOCKS = {1: [1, 2], 2: [0, 1, 2], 3: [0, 1, 2]}


def _inject_rows(t, heads=1):
    """rows of blocks 1 and 3 <- block 2, batch-major layout with `heads` sub-rows per batch row"""
    B = t.shape[0] // heads
    s = B // 4
    if s == 0 or B != 4 * s:
        return t
    t = t.clone()
    v = t.reshape(B, heads, *t.shape[1:])
    v[s:2 * s] = v[2 * s:3 * s]
    v[3 * s:4 * s] = v[2 * s:3 * s]
    return v.reshape(t.shape)


def pnp_forward(sd, cfg, sample, t, ctx, inject_qk: bool, inject_conv: bool, added_cond_kwargs=None):
    xl = bool(getattr(cfg, "addition_embed", False))
    if xl:      # the `_xl` hooks (register.py:246-251,339): every transformer block of up_blocks[1]; up_blocks[1].resnets[0]
        nlev = len(cfg.block_out_channels)
        depth = cfg.depth(nlev - 1 - 1)
        layers = {f"up_blocks.1.attentions.{b}.transformer_blocks.{k}.attn1" for b in range(3) for k in range(depth)}
    else:
        layers = {f"up_blocks.{r}.attentions.{b}.transformer_blocks.0.attn1" for r, bs in QK_BLOCKS.items() for b in bs}

    def qkv_path_hook(q, k, v, is_cross, prefix, heads):
        if inject_qk and not is_cross and prefix in layers:
            return _inject_rows(q, heads), _inject_rows(k, heads), v
        return q, k, v

    res = {("up_blocks.1.resnets.0" if xl else "up_blocks.1.resnets.1"): _inject_rows} if inject_conv else {}
    return unet_ref.unet_forward(sd, cfg, sample, t, ctx, qkv_path_hook=qkv_path_hook, res_inject=res,
                                 added_cond_kwargs=added_cond_kwargs)


@torch.no_grad()
def pnp_loop(sd, cfg, context, x_T, sched: DDIMRef, guidance_scale=7.5, pnp_attn_t=0.5, pnp_f_t=0.8, num_steps=None,
             uncond_list=None, added_cond_kwargs=None):
    """context [4,77,C] = [uncond_src, uncond_tgt, cond_src, cond_tgt]; x_T [1,4,h,w] -> latents [2,4,h,w].
    uncond_list: `PnP_NTI` — both unconditional rows take the null-text embedding of step i
    (`/root/reference/pnp/model/sd_utils.py:340`)."""
    n = sched.num_inference_steps
    qk_n, conv_n = int(n * pnp_attn_t), int(n * pnp_f_t)
    lat = x_T.expand(2, *x_T.shape[1:]).clone()
    context = context.clone()
    for i, t in enumerate(sched.timesteps[: (num_steps or n)]):
        if uncond_list is not None:
            context[0:2] = uncond_list[i].expand(2, -1, -1)
        eps = pnp_forward(sd, cfg, torch.cat([lat] * 2), t, context, i < qk_n, i < conv_n, added_cond_kwargs)
        e_u, e_c = eps.chunk(2)
        lat = sched.step(e_u + guidance_scale * (e_c - e_u), int(t), lat)
    return lat
