"""Null-text inversion (`/root/reference/p2p/inversion/nti.py:9-45`), same class name and signature.

Per timestep the reference Adam-optimises the unconditional embedding through the UNet with torch
autograd.  Here the loop runs on the HIP kernels: `ief_amd.nti.NullTextOptimizer` replays one captured
hipGraph per inner iteration (UNet forward keeping the adjoint's inputs -> objective -> hand-written
activation-gradient pass -> Adam) and reads the loss back for the reference's early-stop rule.
The numerics of the loop are pinned in the oracle against the reference (`oracle/p2p_ref.py:
null_optimization`, fixture G8); `tests/test_gpu_grad.py` holds this class to that oracle.
"""
from ...nti import NullTextOptimizer, run_many
from .ddim import ddim_inversion, ddim_inversion_xl


class NTI(ddim_inversion):
    def null_optimization(self, model, latents, context, num_inner_steps, epsilon, guidance_scale):
        """latents: the 51 latents of `ddim_inversion_loop`; context [2,77,C] = (uncond, cond).
        Returns a list of `num_inference_steps` tensors [1,77,C] (detached), as the reference does (:36,45)."""
        uncond_embeddings, cond_embeddings = context.chunk(2)
        opt = NullTextOptimizer(model, cond_embeddings, guidance_scale, tuple(latents[-1].shape[-2:]))
        try:
            out = opt.run(latents, uncond_embeddings, num_inner_steps, epsilon)
        finally:
            opt.release()
        self.inner_steps_run = opt.inner_steps_run      # diagnostics: Adam steps taken per timestep
        return out

    def null_optimization_many(self, model, latents_list, contexts, num_inner_steps, epsilon, guidance_scale):
        """`null_optimization` for several independent images IN FLIGHT on one GPU (`ief_amd.nti.run_many`): a
        throughput schedule the reference does not have; per image the values are those of `null_optimization`.
        latents_list[k]: the 51 inversion latents of image k; contexts[k]: its [2,77,C] (uncond, cond)."""
        opts = []
        try:
            for latents, context in zip(latents_list, contexts):
                opts.append(NullTextOptimizer(model, context.chunk(2)[1], guidance_scale, tuple(latents[-1].shape[-2:])))
            outs = run_many(opts, latents_list, [c.chunk(2)[0] for c in contexts], num_inner_steps, epsilon)
        finally:
            for o in opts:
                o.release()
        self.inner_steps_run = [o.inner_steps_run for o in opts]
        return outs


class NTI_XL(ddim_inversion_xl):
    """`NTI_XL` of the P2P folder (`/root/reference/p2p/inversion/nti.py:47-96`): null-text optimisation on an SDXL-family
    pipeline.  context = the 4-tuple of `ddim_inversion_xl.get_context`; the embedding restarts from the negative prompt
    embedding at every timestep, conditional and unconditional UNet calls take their own `added_cond_kwargs`
    (`ief_amd.nti.NullTextOptimizer(added_cond=, added_uncond=, lr=, lr_decay=, restart=)`).  Learning rate: this folder's
    copy takes `lr=0.5` and decays it as lr (1 - i / 500) (:50,69); the copies in the masactrl, pnp and pix2pix-zero
    folders hard-code 5e-2 (1 - i / 100) (`pix2pix-zero/inversion/nti.py:69`) — `NTI_XL_5e2` below."""
    LR, LR_DECAY = 0.5, 500.0

    def null_optimization(self, model, latents, context, num_inner_steps, epsilon, guidance_scale, height=None, width=None,
                          lr=None):
        prompt_embeds, negative_prompt_embeds, pooled, negative_pooled = context
        height = height or latents[-1].shape[-2] * model.vae_scale_factor
        width = width or latents[-1].shape[-1] * model.vae_scale_factor
        dev = model.unet.device
        ids = model._get_add_time_ids((height, width), (0, 0), (height, width), dtype=prompt_embeds.dtype).to(dev)
        added_cond = {"text_embeds": pooled.to(dev), "time_ids": ids}
        added_uncond = {"text_embeds": negative_pooled.to(dev), "time_ids": ids}
        opt = NullTextOptimizer(model, prompt_embeds, guidance_scale, tuple(latents[-1].shape[-2:]), added_cond=added_cond,
                                added_uncond=added_uncond, lr=self.LR if lr is None else lr, lr_decay=self.LR_DECAY,
                                restart=True)
        try:
            out = opt.run(latents, negative_prompt_embeds, num_inner_steps, epsilon)
        finally:
            opt.release()
        self.inner_steps_run = opt.inner_steps_run
        return out

    def _optimizer(self, model, latents, context, guidance_scale, lr=None):
        prompt_embeds, negative_prompt_embeds, pooled, negative_pooled = context
        height, width = latents[-1].shape[-2] * model.vae_scale_factor, latents[-1].shape[-1] * model.vae_scale_factor
        dev = model.unet.device
        ids = model._get_add_time_ids((height, width), (0, 0), (height, width), dtype=prompt_embeds.dtype).to(dev)
        return NullTextOptimizer(model, prompt_embeds, guidance_scale, tuple(latents[-1].shape[-2:]),
                                 added_cond={"text_embeds": pooled.to(dev), "time_ids": ids},
                                 added_uncond={"text_embeds": negative_pooled.to(dev), "time_ids": ids},
                                 lr=self.LR if lr is None else lr, lr_decay=self.LR_DECAY, restart=True)

    def null_optimization_many(self, model, latents_list, contexts, num_inner_steps, epsilon, guidance_scale):
        """`null_optimization` for several independent images IN FLIGHT on one GPU (`ief_amd.nti.run_many`), as
        `NTI.null_optimization_many`; contexts[k]: image k's 4-tuple of `get_context`"""
        opts = []
        try:
            for latents, context in zip(latents_list, contexts):
                opts.append(self._optimizer(model, latents, context, guidance_scale))
            outs = run_many(opts, latents_list, [c[1] for c in contexts], num_inner_steps, epsilon)
        finally:
            for o in opts:
                o.release()
        self.inner_steps_run = [o.inner_steps_run for o in opts]
        return outs


class NTI_XL_5e2(NTI_XL):
    """the `NTI_XL` of the masactrl, pnp and pix2pix-zero folders (identical files): lr = 5e-2 (1 - i / 100)"""
    LR, LR_DECAY = 5e-2, 100.0
