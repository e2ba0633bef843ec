"""DDIM inversion: image -> latent -> x_T trajectory.

Same class and methods as `/root/reference/p2p/inversion/ddim.py` (`ddim_inversion` :7-58):
`image2latent` (:35-41), `ddim_inversion_loop` (:21-32, returns all 51 latents + the [2,77,C]
context), `ddim_reverse` (:9-18), `get_context` (:43-58).  The 50 cond-only UNet steps run as a
captured hipGraph (`denoise.FusedDenoiser(mode="invert")`) when no hook owns the attention modules.
"""
from typing import Union

import numpy as np
import torch
from tqdm import tqdm

from ... import hip
from ...denoise import acquire
from ..model.sd_utils import _encode_prompts


class ddim_inversion:
    def ddim_reverse(self, model, model_output, timestep, sample):
        """x_t -> x_{t+1} on device tensors (fp32 elementwise kernel, same operation order as :14-17)."""
        a_cur, a_next = model.scheduler.reverse_coeffs(int(timestep))
        coef = torch.tensor([a_cur, a_next, 1.0], dtype=torch.float32, device=sample.device)
        return hip.cfg_ddim_step(None, model_output.float().contiguous(), sample.float().contiguous(), coef)

    @torch.no_grad()
    def ddim_inversion_loop(self, model, latent, prompt, cross_attention_kwargs=None):
        context = self.get_context(model, prompt)
        uncond_embeddings, cond_embeddings = context.chunk(2)
        unet = model.unet
        native = all(m.is_native() for m in unet.attention_modules()) and getattr(unet, "_plan", None) is None
        if native:
            loop = acquire(model, cond_embeddings, latent.shape[0], tuple(latent.shape[-2:]), None, mode="invert")
            try:
                _, all_latent = loop.run(latent, keep_all=True)
            finally:
                loop.release()
            return all_latent, context
        all_latent = [latent]
        latent = latent.clone().detach()
        n = model.scheduler.num_inference_steps
        for i in tqdm(range(n), desc="Now doing inversion"):
            t = model.scheduler.timesteps[len(model.scheduler.timesteps) - i - 1]
            noise_pred = unet(latent, t, encoder_hidden_states=cond_embeddings,
                              cross_attention_kwargs=cross_attention_kwargs).sample
            latent = self.ddim_reverse(model, noise_pred, t, latent)
            all_latent.append(latent)
        return all_latent, context

    @torch.no_grad()
    def image2latent(self, model, image, device, dtype):
        image = np.array(image)
        image = torch.from_numpy(image).to(dtype) / 127.5 - 1
        image = image.permute(2, 0, 1).unsqueeze(0).to(device)
        latents = model.vae.encode(image)["latent_dist"].mean
        return latents * model.vae.config.scaling_factor

    def get_context(self, model, prompt):
        uncond_embeddings, text_embeddings = _encode_prompts(model, prompt)
        return torch.cat([uncond_embeddings, text_embeddings])


class ddim_inversion_xl(ddim_inversion):
    """`ddim_inversion_xl` (`/root/reference/pix2pix-zero/inversion/ddim.py:60-109`, same class in every method folder):
    the inversion on an SDXL-family pipeline.  `get_context` returns the encoder's 4-tuple; the loop conditions on the
    prompt embeddings and the prompt's `added_cond_kwargs` (pooled embedding + time ids of the image size; the reference
    defaults to 1024x1024, here the latent's own size)."""

    @torch.no_grad()
    def ddim_inversion_loop(self, model, latent, prompt, cross_attention_kwargs=None, height=None, width=None):
        context = self.get_context(model, prompt)
        prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds = context
        height = height or latent.shape[-2] * model.vae_scale_factor
        width = width or latent.shape[-1] * model.vae_scale_factor
        device = model._execution_device
        add_time_ids = model._get_add_time_ids((height, width), (0, 0), (height, width), dtype=prompt_embeds.dtype)
        B = latent.shape[0]
        added_cond_kwargs = {"text_embeds": pooled_prompt_embeds.to(device), "time_ids": add_time_ids.to(device).repeat(B, 1)}
        unet = model.unet
        if not (all(m.is_native() for m in unet.attention_modules()) and getattr(unet, "_plan", None) is None):
            raise RuntimeError("ddim_inversion_xl: an attention hook is installed; invert before registering controllers")
        loop = acquire(model, prompt_embeds.to(device), B, tuple(latent.shape[-2:]), None, mode="invert",
                       added_cond_kwargs=added_cond_kwargs)
        try:
            _, all_latent = loop.run(latent, keep_all=True)
        finally:
            loop.release()
        return all_latent, context

    def get_context(self, model, prompt):
        return model.encode_prompt(prompt=prompt, prompt_2=None, device=model.unet.device, num_images_per_prompt=1,
                                   do_classifier_free_guidance=True, negative_prompt=None, negative_prompt_2=None)
