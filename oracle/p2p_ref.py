"""ORACLE (test infrastructure only — never imported by the product path).

CPU fp32 restatement of the reference's Prompt-to-Prompt control flow around the UNet:
controller call protocol, cross/self map editing, the DDIM step and its inverse, the
edit / inversion / null-text loops.  Each function cites the reference lines it follows.
Pinned by `tests/golden/p2p_*.npz` (made by `tests/golden/make_golden.py`, which imports the
reference's own `model/attention_base.py`, `attention_control.py`, `ptp_utils.py`,
`inversion/ddim.py`, `inversion/nti.py` in the build container).

The DDIM scheduler constants restate diffusers' `DDIMScheduler` [ext] under the config dict at
`/root/reference/p2p/edit_syn.py:46-57` — unpinned against diffusers
itself (not installed), pinned against the reference's `ddim_reverse` through G6.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from . import unet_ref


# --------------------------------------------------------------------------- scheduler
class DDIMRef:
    def __init__(self, num_inference_steps: int = 50, num_train_timesteps: int = 1000,
                 beta_start: float = 0.00085, beta_end: float = 0.012, steps_offset: int = 1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                               dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]  # set_alpha_to_one = False
        self.num_train_timesteps = num_train_timesteps
        self.num_inference_steps = num_inference_steps
        ratio = num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts + steps_offset)
        self.init_noise_sigma = 1.0

    def step(self, eps, t: int, x):
        """eta = 0, epsilon prediction, no clipping (`sd_utils.py:76`)."""
        t = int(t)
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        return a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * eps

    def reverse(self, eps, t: int, x):
        """`/root/reference/p2p/inversion/ddim.py:9-18`."""
        nxt = int(t)
        cur = min(self.num_train_timesteps - 1,
                  nxt - self.num_train_timesteps // self.num_inference_steps)
        a_t = self.alphas_cumprod[cur] if cur >= 0 else self.final_alpha_cumprod
        a_n = self.alphas_cumprod[nxt]
        x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        return a_n ** 0.5 * x0 + (1 - a_n) ** 0.5 * eps


# --------------------------------------------------------------------------- controller
@dataclass
class P2PControlRef:
    """`AttentionControl.__call__` (`attention_base.py:16-28`) + `AttentionControlEdit.forward`
    (:113-125) + the three `replace_cross_attention` bodies (`attention_control.py:15-16,
    28-31, 42-46`), LOW_RESOURCE = False.

    mode: "empty" | "replace" | "refine" | "reweight"
    cross_alpha: [num_steps+1, Bp-1, 1, 1, 77]   (ptp_utils.get_time_words_attention_alpha)
    mapper:  replace -> float [Bp-1,77,77];  refine -> int64 [Bp-1,77]
    alphas:  refine  -> float [Bp-1,1,1,77]
    equalizer: reweight -> float [Bp-1,77]
    prev: reweight -> the controller whose edit is re-weighted (`prev_controller`, attention_control.py:43-44), or None
    """
    mode: str = "empty"
    num_prompts: int = 2
    cross_alpha: Optional[torch.Tensor] = None
    num_self_replace: tuple = (0, 0)
    mapper: Optional[torch.Tensor] = None
    alphas: Optional[torch.Tensor] = None
    equalizer: Optional[torch.Tensor] = None
    num_att_layers: int = -1
    cur_step: int = 0
    cur_att_layer: int = 0
    store: Optional[dict] = None  # AttentionStore semantics when not None
    prev: Optional["P2PControlRef"] = None

    def _edit_cross(self, base, repl):
        if self.mode == "replace":
            return torch.einsum("hpw,bwn->bhpn", base, self.mapper)
        if self.mode == "refine":
            gathered = base[:, :, self.mapper].permute(2, 0, 1, 3)
            return gathered * self.alphas + repl * (1 - self.alphas)
        if self.mode == "reweight":
            if self.prev is not None:      # attention_control.py:43-45: the chained controller's edit first, [Bp-1,h,p,n]
                return self.prev._edit_cross(base, repl) * self.equalizer[:, None, None, :]
            return base[None] * self.equalizer[:, None, None, :]
        raise ValueError(self.mode)

    def _forward(self, attn, is_cross: bool, place: str):
        if self.store is not None and attn.shape[1] <= 32 ** 2:  # attention_base.py:64-68
            self.store.setdefault(f"{place}_{'cross' if is_cross else 'self'}", []).append(attn.clone())
        if self.mode == "empty":
            return attn
        lo, hi = self.num_self_replace
        if is_cross or (lo <= self.cur_step < hi):
            h = attn.shape[0] // self.num_prompts
            a = attn.reshape(self.num_prompts, h, *attn.shape[1:])
            base, repl = a[0], a[1:]
            if is_cross:
                aw = self.cross_alpha[self.cur_step]
                a[1:] = self._edit_cross(base, repl) * aw + (1 - aw) * repl
            elif repl.shape[2] <= 16 ** 2:  # attention_base.py:132-136
                a[1:] = base.unsqueeze(0).expand(repl.shape[0], *base.shape)
            attn = a.reshape(self.num_prompts * h, *a.shape[2:])
        return attn

    def __call__(self, attn, is_cross: bool, place: str):
        h = attn.shape[0]
        attn[h // 2:] = self._forward(attn[h // 2:], is_cross, place)  # cond half only
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1
        return attn


# --------------------------------------------------------------------------- loops
@torch.no_grad()
def edit_loop(sd, cfg, context, x_T, controller, sched: DDIMRef, guidance_scale: float = 7.5,
              num_steps: Optional[int] = None, uncond_list: Optional[List[torch.Tensor]] = None,
              trace: Optional[list] = None, added_cond_kwargs=None):
    """`P2P.text2image_ldm_stable` hot loop + `diffusion_step`
    (`/root/reference/p2p/model/sd_utils.py:58-79`; NTI context swap :133-138; `P2P_XL`, :168-186, passes
    `added_cond_kwargs` to every UNet call).

    context [2*Bp,77,C] = cat(uncond, cond); x_T [1,4,h,w] shared by all prompts (:13-21).
    Returns final latents [Bp,4,h,w].
    """
    bp = context.shape[0] // 2
    lat = x_T.expand(bp, *x_T.shape[1:]).clone()
    if controller is not None:
        controller.num_att_layers = unet_ref.count_attention_layers(cfg)
    hook = controller
    ts = sched.timesteps if num_steps is None else sched.timesteps[:num_steps]
    for i, t in enumerate(ts):
        ctx = context
        if uncond_list is not None:
            ctx = torch.cat([uncond_list[i].expand(bp, *context.shape[1:]), context[bp:]])
        eps = unet_ref.unet_forward(sd, cfg, torch.cat([lat] * 2), t, ctx, hook=hook, added_cond_kwargs=added_cond_kwargs)
        e_u, e_c = eps.chunk(2)
        eps = e_u + guidance_scale * (e_c - e_u)
        lat = sched.step(eps, int(t), lat)
        if trace is not None:
            trace.append(lat.clone())
    return lat


@torch.no_grad()
def ddim_inversion_loop(sd, cfg, cond_emb, latent, sched: DDIMRef, num_steps: Optional[int] = None, added_cond_kwargs=None):
    """`/root/reference/p2p/inversion/ddim.py:21-32`: cond-only UNet, ascending timesteps (`ddim_inversion_xl`, :60-86:
    the same with the prompt's `added_cond_kwargs`)."""
    all_lat = [latent]
    lat = latent.clone()
    n = sched.num_inference_steps if num_steps is None else num_steps
    for i in range(n):
        t = sched.timesteps[len(sched.timesteps) - i - 1]
        eps = unet_ref.unet_forward(sd, cfg, lat, t, cond_emb, added_cond_kwargs=added_cond_kwargs)
        lat = sched.reverse(eps, int(t), lat)
        all_lat.append(lat)
    return all_lat


def null_optimization(sd, cfg, latents, context, sched: DDIMRef, num_inner_steps: int = 10,
                      epsilon: float = 1e-5, guidance_scale: float = 7.5,
                      num_outer: Optional[int] = None, added_cond=None, added_uncond=None, lr: float = 1e-2,
                      restart: bool = False, lr_decay: float = 100.0, grad_trace: Optional[list] = None,
                      start: int = 0, cur0: Optional[torch.Tensor] = None):
    """`/root/reference/p2p/inversion/nti.py:9-45` with the oracle UNet as `model.unet`.
    `NTI_XL` (`/root/reference/pix2pix-zero/inversion/nti.py:47-96`; the masactrl and pnp folders hold the same file, the
    p2p folder's copy uses lr = 0.5 (1 - i / 500), `p2p/inversion/nti.py:50,69`) is the same loop with lr = 5e-2 (:69), the embedding
    RESTARTED from the negative prompt embedding at every timestep (:67, `restart`), and the conditional / unconditional
    UNet calls taking their own `added_cond_kwargs` (:58-61,74,76,90-92).
    start / cur0 (tests): run timesteps start .. start + num_outer - 1 only, from the latent `cur0` and the embedding in
    `context` -- the reference's loop body entered with a given state, so that a product run can be checked timestep by
    timestep from identical starting points."""
    from torch.optim.adam import Adam
    import torch.nn.functional as F

    uncond, cond = context.chunk(2)
    uncond0 = uncond.clone()
    both = None
    if added_cond is not None:
        both = {k: torch.cat([added_uncond[k], added_cond[k]]) for k in added_cond}
    out = []
    cur = latents[-1] if cur0 is None else cur0
    n = sched.num_inference_steps if num_outer is None else num_outer
    for i in range(start, start + n):
        uncond = (uncond0 if restart else uncond).clone().detach()
        uncond.requires_grad = True
        opt = Adam([uncond], lr=lr * (1.0 - i / lr_decay))
        prev = latents[len(latents) - i - 2]
        t = sched.timesteps[i]
        with torch.no_grad():
            e_c = unet_ref.unet_forward(sd, cfg, cur, t, cond, added_cond_kwargs=added_cond)
        for j in range(num_inner_steps):
            e_u = unet_ref.unet_forward(sd, cfg, cur, t, uncond, added_cond_kwargs=added_uncond)
            eps = e_u + guidance_scale * (e_c - e_u)
            rec = sched.step(eps, int(t), cur)
            loss = F.mse_loss(rec, prev)
            opt.zero_grad()
            loss.backward()
            if grad_trace is not None:      # (timestep index, inner step, d loss / d embedding) for tests that weigh elements
                grad_trace.append((i, j, uncond.grad.detach().clone()))
            opt.step()
            if loss.item() < epsilon + i * 2e-5:
                break
        out.append(uncond[:1].detach())
        with torch.no_grad():
            eps = unet_ref.unet_forward(sd, cfg, torch.cat([cur] * 2), t, torch.cat([uncond, cond]), added_cond_kwargs=both)
            e_u, e_c2 = eps.chunk(2)
            cur = sched.step(e_u + guidance_scale * (e_c2 - e_u), int(t), cur)
    return out


def latent_to_uint8(image: torch.Tensor) -> np.ndarray:
    """tail of `latent2image` (`sd_utils.py:85-88`): [-1,1] NCHW -> uint8 NHWC, truncating."""
    image = (image / 2 + 0.5).clamp(0, 1)
    image = image.cpu().permute(0, 2, 3, 1).numpy()
    return (image * 255).astype(np.uint8)
