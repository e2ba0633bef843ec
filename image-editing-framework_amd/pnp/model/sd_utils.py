"""`PnP` sampler — call signature of `/root/reference/pnp/model/sd_utils.py:11-128` (the arguments its CLIs use).

    editor = PnP(pipe, num_inference_steps)
    images = editor(prompt=source_prompt + target_prompt, num_inference_steps=50, guidance_scale=7.5,
                    pnp_attn_t=1.0, pnp_f_t=1.0, latents=None)          # uint8 [2, H, W, 3]

Batch = [uncond_src, uncond_tgt, cond_src, cond_tgt] (`prompt_embeds = cat([negative, positive])`, :72-73), one x_T
shared by both prompts (:75-83).  The 50 steps of the loop (:93-113) are one captured hipGraph replayed per step; the
two injection schedules are device tables (`model/register.py`).
"""
from typing import List, Optional, Union

import numpy as np
import torch

from ... import hip
from ...denoise import acquire, run_interleaved
from ...p2p.model.sd_utils import _encode_prompts, encode_prompt_xl
from .register import (register_attention_control_efficient, register_attention_control_efficient_xl,
                       register_conv_control_efficient, register_conv_control_efficient_xl, register_time,
                       unregister_attention_control_efficient, unregister_conv_control_efficient)


class PnP:
    xl = False

    def __init__(self, pipeline, num_inference_steps) -> None:
        self.model = pipeline
        self.model.scheduler.set_timesteps(num_inference_steps)

    def init_pnp(self, conv_injection_t, qk_injection_t):
        ts = self.model.scheduler.timesteps
        self.qk_injection_timesteps = ts[:qk_injection_t] if qk_injection_t >= 0 else []
        self.conv_injection_timesteps = ts[:conv_injection_t] if conv_injection_t >= 0 else []
        if self.xl:       # `PnP_XL.init_pnp` (:131-136)
            register_attention_control_efficient_xl(self.model, self.qk_injection_timesteps)
            register_conv_control_efficient_xl(self.model, self.conv_injection_timesteps)
        else:
            register_attention_control_efficient(self.model, self.qk_injection_timesteps)
            register_conv_control_efficient(self.model, self.conv_injection_timesteps)

    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str]] = None, height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 50, guidance_scale: float = 7.5, latents: Optional[torch.Tensor] = None,
                 pnp_attn_t: float = 0.5, pnp_f_t: float = 0.8, use_graph: bool = True, return_latents: bool = False,
                 uncond_embeddings_list=None, **unused):
        model = self.model
        dev = model.unet.device
        model.scheduler.set_timesteps(num_inference_steps)
        height = height or model.unet.config.sample_size * model.vae_scale_factor
        width = width or model.unet.config.sample_size * model.vae_scale_factor
        prompt = [prompt] if isinstance(prompt, str) else list(prompt)
        batch_size = len(prompt)
        added_cond_kwargs = None
        if self.xl:       # `PnP_XL.__call__` (:175): both text encoders, pooled embedding and time ids
            emb, added_cond_kwargs = encode_prompt_xl(model, prompt, dev, True, height, width, batch_size)
            uncond_embeddings, text_embeddings = emb[:batch_size], emb[batch_size:]
        else:
            uncond_embeddings, text_embeddings = _encode_prompts(model, prompt)
        C = model.unet.config.in_channels
        if latents is None:
            latents = torch.randn((1, C, height // 8, width // 8), dtype=torch.float32)     # CPU generator, see p2p
        latents = latents.to(dev).float() * model.scheduler.init_noise_sigma
        if latents.shape[0] == 1:
            latents = latents.expand(batch_size, C, height // 8, width // 8)
        self.init_pnp(conv_injection_t=int(num_inference_steps * pnp_f_t), qk_injection_t=int(num_inference_steps * pnp_attn_t))
        try:
            context = torch.cat([uncond_embeddings, text_embeddings])
            g = guidance_scale if guidance_scale > 1.0 else None
            if g is None:
                context = text_embeddings
            # `PnP_NTI`: `prompt_embeds[0:2] = uncond_embeddings_list[i]` every step (:340) = per-step unconditional rows
            loop = acquire(model, context, batch_size, (height // 8, width // 8), g, use_graph=use_graph,
                           uncond_list=uncond_embeddings_list,
                           added_cond_kwargs=added_cond_kwargs if g is not None else None)
            try:
                if use_graph:
                    latents = loop.run(latents)
                else:                                         # the reference's own loop shape (:93-113), step by step
                    loop.start(latents)
                    for t in model.scheduler.timesteps:
                        register_time(model, int(t))
                        loop.step_once()
                    latents = loop.result()
            finally:
                loop.release()
        finally:
            unregister_attention_control_efficient(model)
            unregister_conv_control_efficient(model)
        if return_latents:
            return latents
        return self.latent2image(latents)

    @torch.no_grad()
    def edit_many(self, jobs, num_inference_steps: int = 50, guidance_scale: float = 7.5, pnp_attn_t: float = 0.5,
                  pnp_f_t: float = 0.8, height: Optional[int] = None, width: Optional[int] = None):
        """Several independent Plug-and-Play edits IN FLIGHT on one GPU (a throughput schedule the reference does not have:
        its drivers call the sampler once per image).  jobs: [(prompts [source, target], latents [2,4,h,w] or [1,4,h,w])]
        or [(prompts, latents, uncond_embeddings_list)] (`PnP_NTI`).  Each job's loop is captured while the injection
        schedules are registered, then all loops are stepped in turn on their own streams (`denoise.run_interleaved`).
        Returns one uint8 [2,H,W,3] per job — the values `__call__` gives job by job."""
        model = self.model
        dev = model.unet.device
        if not guidance_scale > 1.0:
            raise NotImplementedError("PnP.edit_many: classifier-free guidance is what the reference's drivers run")
        model.scheduler.set_timesteps(num_inference_steps)
        height = height or model.unet.config.sample_size * model.vae_scale_factor
        width = width or model.unet.config.sample_size * model.vae_scale_factor
        C = model.unet.config.in_channels
        loops = []
        try:
            for job in jobs:
                prompt, latents = job[0], job[1]
                uncond_list = job[2] if len(job) > 2 else None
                added_cond_kwargs = None
                if self.xl:       # as `__call__`: both text encoders, pooled embedding and time ids
                    emb, added_cond_kwargs = encode_prompt_xl(model, list(prompt), dev, True, height, width, len(prompt))
                    uncond_embeddings, text_embeddings = emb[:len(prompt)], emb[len(prompt):]
                else:
                    uncond_embeddings, text_embeddings = _encode_prompts(model, list(prompt))
                latents = latents.to(dev).float() * model.scheduler.init_noise_sigma
                if latents.shape[0] == 1:
                    latents = latents.expand(len(prompt), C, height // 8, width // 8)
                self.init_pnp(conv_injection_t=int(num_inference_steps * pnp_f_t),
                              qk_injection_t=int(num_inference_steps * pnp_attn_t))
                try:
                    loop = acquire(model, torch.cat([uncond_embeddings, text_embeddings]), len(prompt),
                                   (height // 8, width // 8), guidance_scale, uncond_list=uncond_list,
                                   added_cond_kwargs=added_cond_kwargs)
                    loops.append(loop)
                    loop.start(latents)
                finally:                                    # the captured graph carries the plan's tables
                    unregister_attention_control_efficient(model)
                    unregister_conv_control_efficient(model)
            run_interleaved(loops)
            torch.cuda.synchronize()
            return [self.latent2image(loop.result()) for loop in loops]
        finally:
            for loop in loops:
                loop.release()

    @torch.no_grad()
    def latent2image(self, latents, return_type="np"):
        latents = 1 / self.model.vae.config.scaling_factor * latents.detach()
        image = self.model.vae.decode(latents)["sample"]
        if return_type == "np":      # clamp -> NHWC -> uint8 (truncating) as one kernel on the device (hip.image_u8)
            return hip.image_u8(image.float().contiguous()).cpu().numpy()
        return (image / 2 + 0.5).clamp(0, 1)


class PnP_XL(PnP):
    """`PnP_XL` (`/root/reference/pnp/model/sd_utils.py:130-258`): the sampler on an SDXL-family pipeline — the `_xl`
    injection sites of `model/register.py` and the `added_cond_kwargs` of `encode_prompt_xl`."""
    xl = True

    def encode_prompt_xl(self, prompt, device, do_classifier_free_guidance, height, width, batch_size):
        return encode_prompt_xl(self.model, prompt, device, do_classifier_free_guidance, height, width, batch_size)


class PnP_XL_NTI(PnP_XL):
    """`PnP_XL_NTI` (:360-): `PnP_XL` with the per-step unconditional embeddings of `NTI_XL` on both unconditional rows."""


class PnP_NTI(PnP):
    """`PnP_NTI` of the reference (`pnp/model/sd_utils.py:262-`): the sampler with the per-step unconditional embeddings
    of null-text inversion on both unconditional rows; `PnP.__call__` already takes `uncond_embeddings_list`."""

    def __call__(self, *args, uncond_embeddings_list=None, **kw):
        if uncond_embeddings_list is None:
            raise ValueError("PnP_NTI: uncond_embeddings_list is required")
        return super().__call__(*args, uncond_embeddings_list=uncond_embeddings_list, **kw)
