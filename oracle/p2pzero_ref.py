"""ORACLE (test infrastructure only — never imported by the product path): Pix2Pix-zero on the fp32 CPU UNet.

Restates `/root/reference/pix2pix-zero/model/sd_utils.py:6-197` (`P2P_Zero`, SD1.x / 2.x family) with torch autograd:
  - reference pass (:92-122): the plain CFG sampler under the SOURCE prompt; after every UNet call the softmax maps of
    every cross-attention module (`attn2`, stored by `MyAttnProcessor`, `attention_control.py:44-47`) are kept per timestep;
  - edit pass (:152-188), per timestep: `x_in = cat([latents] * 2)` requires grad; UNet under the TARGET prompt;
    `loss = sum_layers ((curr - ref) ** 2).sum((1, 2)).mean(0)` (:166-172); one plain SGD step on x_in with
    lr = guidance_amount (:160,174); the noise is RECOMPUTED on the updated x_in without grad (:177-178);
    `latents = x_in.chunk(2)[0]` (:180); CFG; DDIM step (:183-188).
  - `P2P_Zero_NTI` (:426-617): the same with row 0 of the context replaced by the null-text embedding of step i in
    both passes (:518,582).
  - `P2P_Zero_XL` (:212-423): the same on the SDXL family; every UNet call takes the prompt's `added_cond_kwargs`.
Parity unpinned: the reference module imports diffusers (not installed here), so no fixture could be generated from
it; the rules above are read off its source.
"""
from typing import List, Optional

import torch

from . import unet_ref
from .p2p_ref import DDIMRef


def forward_with_maps(sd, cfg, x, t, ctx, added=None):
    """-> (eps, [cross-attention maps [B*heads, N, 77] in execution order]); differentiable"""
    maps: List[torch.Tensor] = []

    def hook(probs, is_cross, place):
        if is_cross:
            maps.append(probs)
        return probs

    eps = unet_ref.unet_forward(sd, cfg, x, t, ctx, hook=hook, added_cond_kwargs=added)
    return eps, maps


def map_loss(maps, ref_maps):
    loss = 0.0
    for c, r in zip(maps, ref_maps):
        loss = loss + ((c - r) ** 2).sum((1, 2)).mean(0)
    return loss


def input_gradient(sd, cfg, x, t, ctx, ref_maps, added=None):
    """-> (loss, d loss / d x) of the cross-attention-map objective"""
    x_in = x.detach().clone().requires_grad_(True)
    _, maps = forward_with_maps(sd, cfg, x_in, t, ctx, added)
    loss = map_loss(maps, ref_maps)
    grad, = torch.autograd.grad(loss, x_in)
    return float(loss.detach()), grad


def reference_pass(sd, cfg, ctx, x_T, sched: DDIMRef, guidance_scale: float, uncond_list: Optional[list] = None,
                   num_steps: Optional[int] = None, added=None):
    """ctx [2,77,C] = (uncond, cond of the source prompt).  -> (x_0, maps[step][layer])"""
    lat = x_T.clone()
    ctx = ctx.clone()
    all_maps = []
    ts = sched.timesteps if num_steps is None else sched.timesteps[:num_steps]
    with torch.no_grad():
        for i, t in enumerate(ts):
            if uncond_list is not None:
                ctx[0] = uncond_list[i][0]
            eps, maps = forward_with_maps(sd, cfg, torch.cat([lat] * 2), int(t), ctx, added)
            all_maps.append([m.clone() for m in maps])
            eu, ec = eps.chunk(2)
            lat = sched.step(eu + guidance_scale * (ec - eu), int(t), lat)
    return lat, all_maps


def edit_pass(sd, cfg, ctx_edit, x_T, all_maps, sched: DDIMRef, guidance_scale: float, guidance_amount: float = 0.1,
              uncond_list: Optional[list] = None, num_steps: Optional[int] = None, added=None):
    lat = x_T.clone()
    ctx = ctx_edit.clone()
    losses = []
    ts = sched.timesteps if num_steps is None else sched.timesteps[:num_steps]
    for i, t in enumerate(ts):
        if uncond_list is not None:
            ctx[0] = uncond_list[i][0]
        x_in = torch.cat([lat] * 2)
        loss, grad = input_gradient(sd, cfg, x_in, int(t), ctx, all_maps[i], added)
        losses.append(loss)
        x_new = x_in - guidance_amount * grad
        with torch.no_grad():
            eps = unet_ref.unet_forward(sd, cfg, x_new, int(t), ctx, added_cond_kwargs=added)
            lat = x_new.chunk(2)[0]
            eu, ec = eps.chunk(2)
            lat = sched.step(eu + guidance_scale * (ec - eu), int(t), lat)
    return lat, losses


def p2pzero(sd, cfg, ctx_src, ctx_tgt, x_T, sched: DDIMRef, guidance_scale: float = 7.5, guidance_amount: float = 0.1,
            uncond_list: Optional[list] = None, num_steps: Optional[int] = None, added_src=None, added_tgt=None):
    """-> (reconstruction latent, edited latent, per-step losses).  added_src / added_tgt: `P2P_Zero_XL` — the
    `added_cond_kwargs` of the source / target prompt (`sd_utils.py:256,317`)"""
    rec, maps = reference_pass(sd, cfg, ctx_src, x_T, sched, guidance_scale, uncond_list, num_steps, added_src)
    edit, losses = edit_pass(sd, cfg, ctx_tgt, x_T, maps, sched, guidance_scale, guidance_amount, uncond_list, num_steps,
                             added_tgt)
    return rec, edit, losses
