"""`MasaCtrl` sampler (`/root/reference/masactrl/model/sd_utils.py:7-124`): same call signature; the DDIM loop
(:94-115) runs as a captured hipGraph with the editor lowered to a device plan."""
import numpy as np
import torch

from ... import hip
from ...denoise import acquire
from ...p2p.model.sd_utils import encode_prompt_xl


class MasaCtrl:
    xl = False

    def __init__(self, pipeline, num_inference_steps) -> None:
        self.model = pipeline
        self.model.scheduler.set_timesteps(num_inference_steps)

    @torch.no_grad()
    def latent2image(self, latents, return_type="np"):
        latents = 1 / self.model.vae.config.scaling_factor * latents.detach()
        image = self.model.vae.decode(latents)["sample"]
        if return_type == "np":      # clamp -> NHWC -> uint8 (truncating) as one kernel on the device (hip.image_u8)
            return hip.image_u8(image.float().contiguous()).cpu().numpy()
        return (image / 2 + 0.5).clamp(0, 1)

    @torch.no_grad()
    def __call__(self, prompt, batch_size=1, height=512, width=512, num_inference_steps=50, guidance_scale=7.5,
                 latents=None, unconditioning=None, neg_prompt=None, ref_intermediate_latents=None,
                 return_intermediates=False, return_latents=False, uncond_embeddings_list=None, **kwds):
        model = self.model
        dev = model.unet.device
        if isinstance(prompt, list):
            batch_size = len(prompt)
        elif isinstance(prompt, str):
            prompt = [prompt] * batch_size
        if ref_intermediate_latents is not None or kwds.get("dir"):
            raise NotImplementedError("ref_intermediate_latents / dir are not used by the reference CLIs and not built")
        tok = model.tokenizer
        text_input = tok(prompt, padding="max_length", max_length=tok.model_max_length, return_tensors="pt")
        text_embeddings = model.text_encoder(text_input.input_ids.to(dev))[0]
        C = model.unet.config.in_channels
        shape = (batch_size, C, height // 8, width // 8)
        if latents is None:
            latents = torch.randn(shape, dtype=torch.float32).to(dev)     # CPU generator, see p2p/model/sd_utils.py
        else:
            assert tuple(latents.shape) == shape, \
                f"The shape of input latent tensor {latents.shape} should equal to predefined one."
        init_latent = latents.clone()
        # per-step null-text embeddings: `MasaCtrl_NTI(..., uncond_embeddings_list=...)` in the reference
        # (/root/reference/masactrl/model/sd_utils.py:231-245); the same loop serves both classes here
        uncond_list = uncond_embeddings_list if uncond_embeddings_list is not None else (
            unconditioning if isinstance(unconditioning, list) else None)
        if guidance_scale > 1.0:
            uc = tok([neg_prompt or ""] * batch_size, padding="max_length", max_length=tok.model_max_length,
                     return_tensors="pt")
            context = torch.cat([model.text_encoder(uc.input_ids.to(dev))[0], text_embeddings])
            g = guidance_scale
        else:
            context, g = text_embeddings, None
        added_cond_kwargs = None
        if self.xl:       # `MasaCtrl_XL.__call__` (:148-149): both text encoders + the pooled / time-id conditioning
            context, added_cond_kwargs = encode_prompt_xl(model, prompt, dev, guidance_scale > 1.0, height, width, batch_size)
        model.scheduler.set_timesteps(num_inference_steps)
        # a user editor on the generic path runs its own Python inside every forward: step eagerly then (a captured graph
        # would replay the kernels of the capture pass without calling it)
        native = all(m.is_native() for m in model.unet.attention_modules())
        loop = acquire(model, context, batch_size, (height // 8, width // 8), g, uncond_list=uncond_list,
                       added_cond_kwargs=added_cond_kwargs, use_graph=native)
        try:
            latents = loop.run(latents)
        finally:
            loop.release()
        if return_latents:
            return latents, init_latent
        return self.latent2image(latents, return_type="np"), init_latent


class MasaCtrl_XL(MasaCtrl):
    """`MasaCtrl_XL` (`/root/reference/masactrl/model/sd_utils.py:127-226`): the same sampler on an SDXL-family pipeline
    (the reference defaults height = width = 1024; pass the pipeline's own size for the small test family)."""
    xl = True

    def __call__(self, prompt, batch_size=1, height=None, width=None, **kw):
        size = self.model.unet.config.sample_size * self.model.vae_scale_factor
        return super().__call__(prompt, batch_size=batch_size, height=height or size, width=width or size, **kw)

    def encode_prompt_xl(self, prompt, device, do_classifier_free_guidance, height, width, batch_size):
        return encode_prompt_xl(self.model, prompt, device, do_classifier_free_guidance, height, width, batch_size)


class MasaCtrl_NTI(MasaCtrl):
    """`MasaCtrl_NTI` of the reference (`masactrl/model/sd_utils.py:229-`): the sampler with per-step unconditional
    embeddings from null-text inversion; `MasaCtrl.__call__` already takes `uncond_embeddings_list`."""


class MasaCtrl_XL_NTI(MasaCtrl_XL):
    """`MasaCtrl_XL_NTI` (`/root/reference/masactrl/model/sd_utils.py:316-`): `MasaCtrl_XL` with per-step unconditional
    embeddings from `NTI_XL`; `MasaCtrl.__call__` already takes `uncond_embeddings_list`."""
