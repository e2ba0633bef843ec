"""Prompt-to-Prompt editors: the denoising loops around the UNet.

Same classes and call signatures as `/root/reference/p2p/model/sd_utils.py` (`P2P` :9-88,
`P2P_NTI` :90-140):

    editor = P2P(model=pipe, num_inference_steps=50)
    images, x_T = editor.text2image_ldm_stable(pipe, prompts, controller, latent=..., num_inference_steps=50,
                                               guidance_scale=7.5, low_resource=False)

What differs is only HOW a step executes.  When the controller was lowered to a device plan
(`register.py`), the 50-step loop is a captured hipGraph replayed per step (`denoise.FusedDenoiser`);
otherwise (`low_resource`, an arbitrary Python controller, a custom `step_callback`) the loop runs
step by step with the reference's dataflow (:67-79) on the same kernels.

RNG: the reference draws x_T on the CUDA device from the global generator (:15-18).  CUDA, HIP
and CPU generators produce different streams, so x_T is drawn on the CPU generator
(`torch.manual_seed(seed)` / an explicit CPU `generator`) and uploaded — the oracle does the same,
which is what makes seeds comparable (SURVEY.md §7 "RNG").
"""
from typing import List, Optional

import numpy as np
import torch
from tqdm import tqdm

from ... import hip
from ...denoise import acquire, run_interleaved
from .register import register_attention_control, unregister_attention_control


def _encode_prompts(model, prompt: List[str]):
    """cond and uncond text embeddings, as `sd_utils.py:42-55` (also `inversion/ddim.py:43-58`)."""
    tok = model.tokenizer
    text_input = tok(prompt, padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt")
    text_embeddings = model.text_encoder(text_input.input_ids.to(model.device))[0]
    max_length = text_input.input_ids.shape[-1]
    uncond_input = tok([""] * len(prompt), padding="max_length", max_length=max_length, return_tensors="pt")
    uncond_embeddings = model.text_encoder(uncond_input.input_ids.to(model.device))[0]
    return uncond_embeddings, text_embeddings


def encode_prompt_xl(model, prompt, device, do_classifier_free_guidance, height, width, batch_size):
    """The `encode_prompt_xl` every `*_XL` sampler of the reference carries (`p2p/model/sd_utils.py:186-224`,
    `masactrl/model/sd_utils.py:191-226`, `pix2pix-zero/model/sd_utils.py:388-423`): -> (prompt_embeds [2B,77,C] =
    (negative, prompt), added_cond_kwargs = {"text_embeds" [2B, pooled], "time_ids" [2B, 6]})."""
    prompt_embeds, negative_prompt_embeds, pooled, negative_pooled = model.encode_prompt(
        prompt=prompt, prompt_2=None, device=device, num_images_per_prompt=1,
        do_classifier_free_guidance=do_classifier_free_guidance, negative_prompt=None, negative_prompt_2=None)
    original_size = target_size = (height, width)
    add_time_ids = model._get_add_time_ids(original_size, (0, 0), target_size, dtype=prompt_embeds.dtype)
    add_text_embeds = pooled
    if do_classifier_free_guidance:
        prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0)
        add_text_embeds = torch.cat([negative_pooled, pooled], dim=0)
        add_time_ids = torch.cat([add_time_ids, add_time_ids], dim=0)
    add_time_ids = add_time_ids.to(device).repeat(batch_size, 1)
    return prompt_embeds.to(device), {"text_embeds": add_text_embeds.to(device), "time_ids": add_time_ids}


def _fusable(model, controller, low_resource) -> bool:
    if low_resource:
        return False
    plan = getattr(model.unet, "_plan", None)
    if controller is None:
        return all(m.is_native() for m in model.unet.attention_modules())
    # a plan exists only for controller classes whose step_callback is the identity (register.lower_controller)
    return plan is not None and plan.controller is controller


class P2P:
    def __init__(self, model, num_inference_steps) -> None:
        model.scheduler.set_timesteps(num_inference_steps)

    def init_latent(self, latent, model, height, width, generator, batch_size):
        C = model.unet.config.in_channels
        if latent is None:
            latent = torch.randn((1, C, height // 8, width // 8), generator=generator, dtype=torch.float32).to(model.device)
        latent = latent * model.scheduler.init_noise_sigma
        latents = latent.expand(batch_size, C, height // 8, width // 8)
        return latent, latents

    @torch.no_grad()
    def text2image_ldm_stable(self, model, prompt: List[str], controller, num_inference_steps: int = 50,
                              guidance_scale: float = 7.5, generator: Optional[torch.Generator] = None,
                              latent: Optional[torch.FloatTensor] = None, low_resource: bool = False,
                              uncond_embeddings_list=None, height: Optional[int] = None, width: Optional[int] = None,
                              return_latents: bool = False, cfg_split_group=None):
        """cfg_split_group (not a reference argument): a torch.distributed group of TWO ranks that run this ONE edit
        together — rank 0 the unconditional rows of the CFG batch, rank 1 the conditional rows, one eps exchange per
        step (`denoise.CfgSplitDenoiser`); both ranks return the same latents / images."""
        if cfg_split_group is not None:
            return self._text2image_cfg_split(model, prompt, controller, num_inference_steps, guidance_scale, latent,
                                              uncond_embeddings_list, height, width, return_latents, cfg_split_group)
        if controller is not None:
            register_attention_control(model, controller)
        if height is None:
            height = width = model.unet.config.sample_size * model.vae_scale_factor
        batch_size = len(prompt)
        uncond_embeddings, text_embeddings, added_cond_kwargs = self._encode(model, prompt, height, width)
        latent, latents = self.init_latent(latent, model, height, width, generator, batch_size)
        model.scheduler.set_timesteps(num_inference_steps)
        context = torch.cat([uncond_embeddings, text_embeddings])
        if _fusable(model, controller, low_resource):
            loop = acquire(model, context, batch_size, (height // 8, width // 8), guidance_scale,
                                 uncond_list=uncond_embeddings_list, added_cond_kwargs=added_cond_kwargs)
            try:
                latents = loop.run(latents)
            finally:
                loop.release()
        else:
            for i, t in enumerate(tqdm(model.scheduler.timesteps, desc="Now doing P2P editing")):
                if uncond_embeddings_list is not None:
                    context = torch.cat([uncond_embeddings_list[i].expand(*text_embeddings.shape), text_embeddings])
                latents = self.diffusion_step(model, controller, latents, context, t, guidance_scale, low_resource,
                                              added_cond_kwargs=added_cond_kwargs)
        if return_latents:
            return latents, latent
        image = self.latent2image(model.vae, latents)
        return image, latent

    @torch.no_grad()
    def _text2image_cfg_split(self, model, prompt, controller, num_inference_steps, guidance_scale, latent,
                              uncond_embeddings_list, height, width, return_latents, group):
        import torch.distributed as dist
        from ...denoise import CfgSplitDenoiser
        half = dist.get_rank(group)
        register_attention_control(model, controller, rows="cond" if half == 1 else "uncond")
        if height is None:
            height = width = model.unet.config.sample_size * model.vae_scale_factor
        batch_size = len(prompt)
        uncond_embeddings, text_embeddings, added = self._encode(model, prompt, height, width)
        if added is not None:
            raise NotImplementedError("CFG split is built for the SD1.x / SD2.x pipelines")
        latent, latents = self.init_latent(latent, model, height, width, None, batch_size)
        # both ranks must start from the SAME x_T: take the first rank's (seeded draws agree anyway; a passed-in latent might not)
        # (a host tensor only on a gloo group: an RCCL group has no backend for CPU tensors -- as `denoise.exchange_eps`)
        src = dist.get_global_rank(group, 0) if group is not None else 0
        if dist.get_backend(group) == "gloo":
            x0 = latents.float().contiguous().cpu()
            dist.broadcast(x0, src=src, group=group)
            latents = x0.to(model.device)
        else:
            latents = latents.float().contiguous().clone()
            dist.broadcast(latents, src=src, group=group)
        model.scheduler.set_timesteps(num_inference_steps)
        loop = CfgSplitDenoiser(model, torch.cat([uncond_embeddings, text_embeddings]), batch_size,
                                (height // 8, width // 8), guidance_scale, group=group, uncond_list=uncond_embeddings_list)
        try:
            latents = loop.run(latents)
        finally:
            loop.release()
        if return_latents:
            return latents, latent
        return self.latent2image(model.vae, latents), latent

    def _encode(self, model, prompt, height, width):
        """-> (uncond [B,77,C], cond [B,77,C], added_cond_kwargs or None)"""
        uncond_embeddings, text_embeddings = _encode_prompts(model, prompt)
        return uncond_embeddings, text_embeddings, None

    @torch.no_grad()
    def edit_many(self, model, jobs, num_inference_steps: int = 50, guidance_scale: float = 7.5,
                  height: Optional[int] = None, width: Optional[int] = None):
        """Several independent edits IN FLIGHT on one GPU (a throughput schedule the reference does not have: its
        drivers call `text2image_ldm_stable` once per image).  jobs: [(prompts, controller, latent x_T [1,4,h,w])] or
        [(prompts, controller, x_T, uncond_embeddings_list)] (null-text embeddings, as `P2P_NTI`);
        every controller must be one of the lowered classes.  Each job's loop is captured while its controller is
        registered, then all loops are stepped in turn on their own streams (`denoise.run_interleaved`).  Returns
        [(images uint8 [len(prompts),H,W,3], x_T)] — the same values as one `text2image_ldm_stable` call per job."""
        if height is None:
            height = width = model.unet.config.sample_size * model.vae_scale_factor
        model.scheduler.set_timesteps(num_inference_steps)
        loops, firsts = [], []
        try:
            for job in jobs:
                prompt, controller, latent = job[:3]
                uncond_list = job[3] if len(job) > 3 else None
                register_attention_control(model, controller)
                if not _fusable(model, controller, False):
                    raise RuntimeError("edit_many: only controllers lowered to a device plan can run concurrently")
                # `_encode`: the SD1.x / 2.x text encoder, or (P2P_XL) both SDXL encoders + pooled embedding / time ids
                uncond_embeddings, text_embeddings, added_cond_kwargs = self._encode(model, prompt, height, width)
                latent, latents = self.init_latent(latent, model, height, width, None, len(prompt))
                loop = acquire(model, torch.cat([uncond_embeddings, text_embeddings]), len(prompt),
                                     (height // 8, width // 8), guidance_scale, uncond_list=uncond_list,
                                     added_cond_kwargs=added_cond_kwargs)
                loop.start(latents)
                loops.append(loop)
                firsts.append(latent)
                unregister_attention_control(model, None)      # the captured graph carries the plan's tables
            run_interleaved(loops)
            torch.cuda.synchronize()
            return [(self.latent2image(model.vae, loop.result()), first) for loop, first in zip(loops, firsts)]
        finally:
            for loop in loops:
                loop.release()

    def diffusion_step(self, model, controller, latents, context, t, guidance_scale, low_resource=False,
                       added_cond_kwargs=None):
        """one eager step (`sd_utils.py:67-79`, XL: :175-186); CFG + DDIM update fused in one kernel."""
        latents = latents.float().contiguous()
        if low_resource:
            bp = latents.shape[0]
            half = lambda i: None if added_cond_kwargs is None else {k: v[i * bp:(i + 1) * bp] for k, v in added_cond_kwargs.items()}
            eps_u = model.unet(latents, t, encoder_hidden_states=context[:bp], added_cond_kwargs=half(0))["sample"]
            eps_c = model.unet(latents, t, encoder_hidden_states=context[bp:], added_cond_kwargs=half(1))["sample"]
        else:
            eps = model.unet(torch.cat([latents] * 2), t, encoder_hidden_states=context,
                             added_cond_kwargs=added_cond_kwargs)["sample"]
            eps_u, eps_c = eps.chunk(2)
        a_t, a_p = model.scheduler.step_coeffs(int(t))
        coef = torch.tensor([a_t, a_p, float(guidance_scale)], dtype=torch.float32, device=latents.device)
        latents = hip.cfg_ddim_step(eps_u.contiguous(), eps_c.contiguous(), latents, coef)
        if controller is not None:
            latents = controller.step_callback(latents)
        return latents

    @torch.no_grad()
    def latent2image(self, vae, latents):
        latents = 1 / vae.config.scaling_factor * latents
        image = vae.decode(latents)["sample"]
        # (image / 2 + 0.5).clamp(0, 1) -> NHWC -> * 255 -> uint8 (truncating), `sd_utils.py:85-88`, as one kernel on the
        # device: uint8 NHWC crosses PCIe instead of fp32 NCHW
        return hip.image_u8(image.float().contiguous()).cpu().numpy()


class P2P_XL(P2P):
    """`P2P_XL` (`/root/reference/p2p/model/sd_utils.py:142-224`): the same sampler on an SDXL-family pipeline.  What
    differs is the text side — two encoders, pooled embeddings and the six time ids handed to the UNet as
    `added_cond_kwargs` — and the latent size (the reference hard-codes 1024x1024, :160; here the pipeline's own sample
    size, which is 1024 for `xl-base`).  The additional embedding is constant over the steps, so the captured step graph
    reads it folded into its per-step time-embedding rows (`denoise.FusedDenoiser`)."""

    def _encode(self, model, prompt, height, width):
        emb, added = self.encode_prompt_xl(model, prompt, model._execution_device, True, height, width, len(prompt))
        return emb[:len(prompt)], emb[len(prompt):], added

    def encode_prompt_xl(self, model, prompt, device, do_classifier_free_guidance, height, width, batch_size):
        return encode_prompt_xl(model, prompt, device, do_classifier_free_guidance, height, width, batch_size)


class P2P_NTI(P2P):
    """`text2image_ldm_stable(..., uncond_embeddings_list=[50 x [1,77,C]])` (:92-140): the base class
    already takes the per-step null-text embeddings, selected per step on the device."""


class P2P_XL_NTI(P2P_XL):
    """`P2P_XL_NTI` (`/root/reference/p2p/model/sd_utils.py:226-`): `P2P_XL` with the per-step null-text embeddings of
    `NTI_XL` on the unconditional rows; the base sampler already takes `uncond_embeddings_list`."""
