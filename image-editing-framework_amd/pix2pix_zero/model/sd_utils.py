"""`P2P_Zero` / `P2P_Zero_NTI` samplers — call signature of `/root/reference/pix2pix-zero/model/sd_utils.py:6-197,426-617`
(SD1.x / 2.x family), `P2P_Zero_XL` (:212-423) and `P2P_Zero_XL_NTI` (:619-784) on the SDXL family.

    editor = P2P_Zero(pipe, num_inference_steps)
    image_rec, image_edit = editor(prompt=source_prompt + target_prompt, num_inference_steps=50, guidance_scale=7.5,
                                   only_sample=False, edit_dir=None, latents=x_T)     # uint8 [1, H, W, 3] each

Two loops over the same x_T:
  reference pass (:92-122)  plain CFG sampler (UNet batch 2) under prompt[0]; every cross-attention module's softmax maps
                            of every step are KEPT ON THE DEVICE (the reference moves them to the CPU and back, :110,171):
                            one fp16 tensor [steps, B*heads, N, 77] per module, 3.3 GB for SD1.5 at 512x512.
  edit pass (:152-188)      per step: x_in = cat([latents] * 2); UNet under prompt[1]; objective = sum over modules of
                            ((maps - reference maps) ** 2).sum((1, 2)).mean(0); ONE plain SGD step on x_in with
                            lr = guidance_amount; noise recomputed on the updated x_in; latents = x_in[0]; CFG; DDIM step.
The reference differentiates with torch autograd.  Here `grad.UNetAdjoint(mode="input")` chains the hand-written
activation-gradient kernels from the queries of the 16 cross-attention modules down to the latent; each loop body is one
captured hipGraph replayed per step (per-step rows — time embedding, DDIM coefficients, reference maps, null-text row —
are copied into the static buffers the graph reads); buffers and graphs are kept by the editor across images (`_Engine`).
"""
from typing import List, Optional, Union

import numpy as np
import torch

from ... import hip
from ...grad import UNetAdjoint
from ...p2p.model.sd_utils import _encode_prompts, encode_prompt_xl
from .attention_control import prep_unet, restore_original_processors

GRAD_SCALE = 1024.0     # the fp16 gradients of the map objective are carried times this (undone in the SGD step)


class _Engine:
    """static buffers + the two captured loop bodies for one (latent size, step count, guidance_amount): built on first
    use and kept across images, since everything an image changes is a buffer the graphs READ"""

    def __init__(self, model, h, w, nsteps, guidance_amount, use_graph, temb_rows=1):
        unet = model.unet
        dev = unet.device
        C = unet.config.in_channels
        f32 = dict(dtype=torch.float32, device=dev)
        self.unet, self.use_graph, self.guidance_amount = unet, use_graph, float(guidance_amount)
        self.coef = torch.zeros(4, **f32)
        self.temb = torch.zeros(temb_rows, unet._temb_width, **f32)    # one row, or one per batch row (SDXL's embedding)
        self.lat = torch.zeros(1, C, h, w, **f32)
        self.x_in = torch.zeros(2, C, h, w, **f32)
        self.zero_eps = torch.zeros(2, C, h, w, **f32)
        self.step_loss = torch.zeros(1, **f32)
        adt = unet.dtype                # activation dtype: fp16, or fp32 in the fp32-storage modes (maps then fp32 as well)
        self.ctx16 = torch.zeros(2, 77, unet.config.cross_attention_dim, dtype=adt, device=dev)
        self.cross = [m for m in unet.attention_modules() if m.is_cross]
        with torch.no_grad():           # one dry forward: allocator warm-up, and every module notes its query count
            unet(self.x_in, encoder_hidden_states=self.ctx16, temb_row=self.temb)
        self.stage = [torch.zeros(2 * m.heads, m.last_tokens, 77, dtype=adt, device=dev) for m in self.cross]
        self.maps = [torch.zeros(nsteps, *st.shape, dtype=adt, device=dev) for st in self.stage]
        self.adj = UNetAdjoint(unet, GRAD_SCALE, mode="input")
        self.adj.prepack()
        self.adj.set_reference_maps(self.stage)      # the edit graph reads the staged maps of the current step
        self._ref = self._edit = None

    def set_ctx(self, emb):
        """fp32 [2,77,C] embeddings -> the context buffer the graphs read"""
        if self.ctx16.dtype == torch.float32:
            self.ctx16.copy_(emb)
        else:
            hip.to_f16(emb.contiguous(), out=self.ctx16)

    def _ref_body(self):
        self.x_in.copy_(self.lat.expand_as(self.x_in))
        eps = self.unet(self.x_in, encoder_hidden_states=self.ctx16, temb_row=self.temb)["sample"]
        hip.cfg_ddim_step(eps[0:1], eps[1:2], self.lat, self.coef, out=self.lat)

    def _edit_body(self):
        adj, x_in = self.adj, self.x_in
        x_in.copy_(self.lat.expand_as(x_in))
        adj.forward(x_in, self.temb, self.ctx16)
        d_x = adj.backward(self.zero_eps)
        torch.sum(adj.loss_parts, dim=0, keepdim=True, out=self.step_loss)
        hip.axpy(x_in, d_x, -self.guidance_amount / GRAD_SCALE)              # SGD, lr = guidance_amount (:160,174)
        eps = self.unet(x_in, encoder_hidden_states=self.ctx16, temb_row=self.temb)["sample"]
        hip.cfg_ddim_step(eps[0:1], eps[1:2], x_in[0:1], self.coef, out=self.lat)   # latents = x_in.chunk(2)[0] (:180)

    def _capture(self, body):
        if not self.use_graph:
            return body
        used0 = hip.counters_used()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            body()                                  # warm-up: allocator pools, packed adjoint weights
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        # split-K launches combine inside the launch: the graph owns its arrival counters (kept alive with the loop)
        arena = hip.counter_arena(hip.counters_used() - used0, self.lat.device)
        self._arenas = getattr(self, "_arenas", []) + [arena]
        g = torch.cuda.CUDAGraph()
        with arena, torch.cuda.graph(g):
            body()
        return g.replay

    def ref_step(self):
        if self._ref is None:           # captured with the map buffers installed (the caller sets `map_out`)
            saved = self.lat.clone()
            self._ref = self._capture(self._ref_body)
            self.lat.copy_(saved)
        self._ref()

    def edit_step(self):
        if self._edit is None:
            saved = self.lat.clone()
            self._edit = self._capture(self._edit_body)
            self.lat.copy_(saved)
        self._edit()


class P2P_Zero:
    def __init__(self, pipeline, num_inference_steps):
        self.model = pipeline
        self.model.scheduler.set_timesteps(num_inference_steps)
        self.last_losses: List[float] = []
        self._engines = {}

    def release(self):
        """drop the cached graphs and map buffers (3.3 GB for SD1.5 at 512x512)"""
        self._engines.clear()

    def __call__(self, prompt: Union[str, List[str]] = None, height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 50, guidance_scale: float = 7.5, latents: Optional[torch.Tensor] = None,
                 guidance_amount: float = 0.1, edit_dir=None, only_sample: bool = False, uncond_embeddings_list=None,
                 use_graph: bool = True, return_latents: bool = False, num_steps: Optional[int] = None, **unused):
        model = self.model
        unet, sched = model.unet, model.scheduler
        dev = unet.device
        if not guidance_scale > 1.0:
            raise NotImplementedError("P2P_Zero: the reference's CLIs run with classifier-free guidance (7.5)")
        sched.set_timesteps(num_inference_steps)
        height = height or unet.config.sample_size * model.vae_scale_factor
        width = width or unet.config.sample_size * model.vae_scale_factor
        h, w = height // 8, width // 8
        prompt = [prompt] if isinstance(prompt, str) else list(prompt)
        C = unet.config.in_channels
        if latents is None:
            latents = torch.randn((1, C, h, w), dtype=torch.float32)        # CPU generator, see p2p
        latents_init = (latents.to(dev).float() * sched.init_noise_sigma).contiguous()
        ts = sched.timesteps.tolist()
        if num_steps is not None:
            ts = ts[:num_steps]
        f32 = dict(dtype=torch.float32, device=dev)
        coef_table = torch.tensor([[*sched.step_coeffs(t), float(guidance_scale), 0.0] for t in ts], **f32)
        ts_dev = torch.tensor(ts, **f32)
        null_rows = None
        if uncond_embeddings_list is not None:       # P2P_Zero_NTI: `prompt_embeds[0] = uncond_embeddings_list[i]` (:518,582)
            null_rows = [unet._act(u.to(dev))[0] for u in uncond_embeddings_list]

        def context_of(p):
            """-> ([2,77,C] = (negative, prompt) embeddings, per-step time-embedding rows [steps, 1 or 2, width])"""
            emb, added = self._encode(model, p, height, width)
            aug = unet.aug_embedding(added)
            rows = unet.time_rows(ts_dev, aug)
            return emb.to(dev).float(), rows.reshape(len(ts), 1 if aug is None else 2, -1).contiguous()

        unet, self.original_processors = prep_unet(unet)
        saved_cache = [(m, m.cache_kv) for m in unet.attention_modules()]
        for m in unet.attention_modules():
            m.cache_kv = False          # the context buffer is rewritten in place (prompt switch, null-text rows)
            m._kv_key, m._kv = None, None
        nb = 2 if unet.cfg.addition_embed else 1
        key = (h, w, len(ts), float(guidance_amount), bool(use_graph))
        try:
            E = self._engines.get(key)
            if E is None:
                self._engines.clear()                       # one shape at a time: the map buffers are large
                E = self._engines[key] = _Engine(model, h, w, len(ts), guidance_amount, use_graph, temb_rows=nb)

            def set_step(i):
                E.temb.copy_(temb_table[i]), E.coef.copy_(coef_table[i])
                if null_rows is not None:
                    E.ctx16[0].copy_(null_rows[i])

            # ---------------- reference pass: record the maps
            emb, temb_table = context_of(prompt[0])
            E.set_ctx(emb)
            for m, st in zip(E.cross, E.stage):
                m.map_out = st
            with torch.no_grad():
                E.lat.copy_(latents_init)
                for i in range(len(ts)):
                    set_step(i)
                    E.ref_step()
                    for st, mp in zip(E.stage, E.maps):
                        mp[i].copy_(st)
            for m in E.cross:
                m.map_out = None
            rec_latents = E.lat.clone()
            if only_sample:
                return rec_latents if return_latents else self.latent2image(rec_latents)

            # ---------------- edit pass
            emb, temb_table = context_of(prompt[1])
            if edit_dir is not None:            # `prompt_embeds_edit += edit_dir` (:145-146)
                emb = emb + edit_dir.to(dev).float()
            E.set_ctx(emb)
            loss_log = torch.zeros(len(ts), **f32)
            with torch.no_grad():
                E.lat.copy_(latents_init)
                for i in range(len(ts)):
                    set_step(i)
                    for st, mp in zip(E.stage, E.maps):
                        st.copy_(mp[i])
                    E.edit_step()
                    loss_log[i:i + 1].copy_(E.step_loss)
            self.last_losses = loss_log.tolist()
            edit_latents = E.lat.clone()
        finally:
            for m in unet.attention_modules():
                m.map_out = None
            for m, c in saved_cache:
                m.cache_kv = c
                m._kv_key, m._kv = None, None
            restore_original_processors(unet, self.original_processors)
        if return_latents:
            return rec_latents, edit_latents
        return self.latent2image(rec_latents), self.latent2image(edit_latents)

    def _encode(self, model, p, height, width):
        """-> ([2,77,C] = (negative, prompt) embeddings, added_cond_kwargs or None)"""
        with torch.no_grad():
            u, c = _encode_prompts(model, [p])
        return torch.cat([u, c]), None

    @torch.no_grad()
    def latent2image(self, latents, return_type="np"):
        latents = 1 / self.model.vae.config.scaling_factor * latents.detach()
        image = self.model.vae.decode(latents)["sample"]
        if return_type == "np":      # clamp -> NHWC -> uint8 (truncating) as one kernel on the device (hip.image_u8)
            return hip.image_u8(image.float().contiguous()).cpu().numpy()
        return (image / 2 + 0.5).clamp(0, 1)


class P2P_Zero_NTI(P2P_Zero):
    """`P2P_Zero` with row 0 of the context replaced per step by the null-text embedding (`uncond_embeddings_list`)."""

    def __call__(self, *args, uncond_embeddings_list=None, **kw):
        if uncond_embeddings_list is None:
            raise ValueError("P2P_Zero_NTI: uncond_embeddings_list is required")
        return super().__call__(*args, uncond_embeddings_list=uncond_embeddings_list, **kw)


class P2P_Zero_XL(P2P_Zero):
    """`P2P_Zero_XL` (`/root/reference/pix2pix-zero/model/sd_utils.py:212-423`): the same two passes on an SDXL-family
    pipeline.  Each prompt comes with its `added_cond_kwargs` (pooled embedding + six time ids, :388-423); they are
    constant over the steps, so each pass reads them folded into its per-step time-embedding rows (one row per batch
    row).  The transformers are several blocks deep there: the reverse pass walks every block (`grad.UNetAdjoint`) and
    the objective has one term per cross-attention module, 70 of them on SDXL."""

    def _encode(self, model, p, height, width):
        return self.encode_prompt_xl(p, model._execution_device, True, height, width, 1)

    def encode_prompt_xl(self, prompt, device, do_classifier_free_guidance, height, width, batch_size):
        return encode_prompt_xl(self.model, prompt, device, do_classifier_free_guidance, height, width, batch_size)


class P2P_Zero_XL_NTI(P2P_Zero_XL):
    """`P2P_Zero_XL_NTI` (`/root/reference/pix2pix-zero/model/sd_utils.py:619-784`): `P2P_Zero_XL` with row 0 of the context
    replaced per step by the null-text embedding of `NTI_XL` (:696,749)."""

    def __call__(self, *args, uncond_embeddings_list=None, **kw):
        if uncond_embeddings_list is None:
            raise ValueError("P2P_Zero_XL_NTI: uncond_embeddings_list is required")
        return super().__call__(*args, uncond_embeddings_list=uncond_embeddings_list, **kw)
