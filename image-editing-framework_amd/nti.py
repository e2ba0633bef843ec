"""Null-text optimisation engine: the loop of `/root/reference/p2p/inversion/nti.py:9-45` on the HIP kernels.

Per DDIM timestep i the reference
  1. computes eps_cond once (no grad),
  2. runs <= num_inner_steps of {UNet(uncond) -> CFG -> scheduler.step -> mse against the inversion latent ->
     backward -> Adam}, stopping early when the loss (of the PRE-update embedding) falls under eps + 2e-5 i,
  3. advances latent_cur with the optimised embedding.
Here each of the three is ONE captured hipGraph over static buffers (a fresh Adam per timestep = zeroed
moments and step counter, lr in a device scalar), so an inner iteration costs one graph replay plus the
host read of the loss that the early-stop rule needs — exactly the reference's `loss.item()`.

The gradient is not torch autograd: `grad.UNetAdjoint` chains activation-gradient kernels (weights are
frozen on this path).  Everything that varies with the timestep is a row of a device table.
"""
from typing import List, Optional

import torch

from . import hip
from .grad import UNetAdjoint


class NullTextOptimizer:
    def __init__(self, model, cond: torch.Tensor, guidance_scale: float, latent_hw, grad_scale: float = 1.0,
                 use_graph: bool = True, added_cond=None, added_uncond=None, lr: float = 1e-2, restart: bool = False,
                 lr_decay: float = 100.0):
        """added_cond / added_uncond, lr, restart: `NTI_XL` (`/root/reference/pix2pix-zero/inversion/nti.py:47-96`) — the
        conditional and unconditional UNet calls take their own SDXL `added_cond_kwargs` (folded into two tables of
        per-step time-embedding rows), lr = 5e-2, and the embedding restarts from its initial value every timestep."""
        self.model, self.unet, self.sched = model, model.unet, model.scheduler
        dev = self.unet.device
        self.dev = dev
        self.adj = UNetAdjoint(self.unet, grad_scale)
        self.adj.prepack()
        h, w = latent_hw
        C = self.unet.config.in_channels
        f32 = dict(dtype=torch.float32, device=dev)
        self.lat = torch.zeros(1, C, h, w, **f32)
        self.target = torch.zeros(1, C, h, w, **f32)
        self.eps_c = torch.zeros(1, C, h, w, **f32)
        self.d_eps = torch.zeros(1, C, h, w, **f32)
        self.stats = torch.zeros(2, **f32)
        self.hyper = torch.tensor([1e-2, 0.9, 0.999, 1e-8], **f32)     # torch.optim.Adam defaults (nti.py:17)
        self.adam_step = torch.zeros(1, dtype=torch.int32, device=dev)
        ts = self.sched.timesteps.tolist()
        self.num_steps = len(ts)
        g = float(guidance_scale)
        self.coef_table = torch.tensor([[*self.sched.step_coeffs(t), g, 0.0] for t in ts], **f32)
        self.coef = torch.zeros(4, **f32)
        tsd = torch.tensor(ts, **f32)
        self.temb_table = self.unet.time_rows(tsd, self.unet.aug_embedding(added_uncond)).contiguous()     # uncond calls
        self.temb = torch.zeros(1, self.temb_table.shape[1], **f32)
        if added_cond is not None:
            self.temb_table_c = self.unet.time_rows(tsd, self.unet.aug_embedding(added_cond)).contiguous()
            self.temb_c = torch.zeros_like(self.temb)
        else:
            self.temb_table_c, self.temb_c = self.temb_table, self.temb
        self.lr, self.restart, self.lr_decay = float(lr), bool(restart), float(lr_decay)      # lr_i = lr (1 - i / lr_decay)
        self.f32 = self.unet.dtype == torch.float32          # fp32-storage modes: the context the UNet reads IS the parameter
        self.cond16 = self.unet._act(cond.to(dev))
        L, Cc = self.cond16.shape[1:]
        self.param = torch.zeros(1, L, Cc, **f32)
        self.m = torch.zeros_like(self.param)
        self.v = torch.zeros_like(self.param)
        self.p16 = self.param if self.f32 else torch.zeros(1, L, Cc, dtype=torch.float16, device=dev)
        self.use_graph = use_graph
        self._graphs = None
        self.inner_steps_run: List[int] = []     # per timestep, how many Adam steps the early-stop rule allowed
        self.last_losses: List[float] = []

    # ------------------------------------------------------------------ the three stream-ordered bodies
    def _body_cond(self):
        eps = self.unet(self.lat, encoder_hidden_states=self.cond16, temb_row=self.temb_c)["sample"]
        self.eps_c.copy_(eps)

    def _body_inner(self):
        eps_u = self.adj.forward(self.lat, self.temb, self.p16)
        hip.nti_loss_grad(eps_u, self.eps_c, self.lat, self.target, self.coef, self.d_eps, self.stats, self.adj.grad_scale)
        g16 = self.adj.backward(self.d_eps)
        hip.nti_adam(self.param, self.m, self.v, g16, self.stats, self.hyper, self.adam_step, self.p16)

    def _body_tail(self):
        eps_u = self.unet(self.lat, encoder_hidden_states=self.p16, temb_row=self.temb)["sample"]
        hip.cfg_ddim_step(eps_u, self.eps_c, self.lat, self.coef, out=self.lat)

    def _sync_p16(self):
        if not self.f32:
            hip.to_f16(self.param, out=self.p16)

    def _capture(self):
        for m in self.unet.attention_modules():
            m.cache_kv = False          # the uncond context changes under the same buffer: never cache its K/V
        state = [self.lat, self.param, self.m, self.v, self.adam_step, self.eps_c] + ([] if self.f32 else [self.p16])
        saved = [t.clone() for t in state]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        need = []
        with torch.cuda.stream(s):          # warm-up: allocator pools, packed adjoint weights
            for body in (self._body_cond, self._body_inner, self._body_tail):
                used0 = hip.counters_used()
                body()
                need.append(hip.counters_used() - used0)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        graphs, self._arenas = [], []
        for body, n in zip((self._body_cond, self._body_inner, self._body_tail), need):
            arena = hip.counter_arena(n, self.dev)      # this graph's own split-K arrival counters
            g = torch.cuda.CUDAGraph()
            with arena, torch.cuda.graph(g):
                body()
            graphs.append(g)
            self._arenas.append(arena)
        for t, sv in zip(state, saved):
            t.copy_(sv)
        self._graphs = graphs

    def _run(self, which: int):
        if self._graphs is not None:
            self._graphs[which].replay()
        else:
            (self._body_cond, self._body_inner, self._body_tail)[which]()

    # ------------------------------------------------------------------ public: one image = begin, then per timestep
    # outer_begin -> inner_step / inner_loss ... -> outer_end  (so several images can be interleaved: `run_many`)
    def begin(self, latents: List[torch.Tensor], uncond: torch.Tensor):
        dev = self.dev
        self._latents = latents
        self.lat.copy_(latents[-1].to(dev).float())
        self.param.copy_(uncond.to(dev).float()[:1])
        self._param0 = self.param.clone()
        self._sync_p16()
        if self.use_graph and self._graphs is None:
            self._capture()
        elif not self.use_graph:
            for m in self.unet.attention_modules():
                m.cache_kv = False
        self.out: List[torch.Tensor] = []
        self.inner_steps_run, self.last_losses = [], []

    def outer_begin(self, i: int):
        latents = self._latents
        self.temb.copy_(self.temb_table[i:i + 1])
        if self.temb_c is not self.temb:
            self.temb_c.copy_(self.temb_table_c[i:i + 1])
        if self.restart:
            self.param.copy_(self._param0)
            self._sync_p16()
        self.coef.copy_(self.coef_table[i])
        self.target.copy_(latents[len(latents) - i - 2].to(self.dev).float())
        self.m.zero_(), self.v.zero_(), self.adam_step.zero_()          # `Adam([uncond], lr=...)` anew (nti.py:17)
        self.hyper[0:1].fill_(self.lr * (1.0 - i / self.lr_decay))
        self._run(0)
        self._done, self._loss = 0, float("nan")

    def inner_step(self):
        self._run(1)
        self._done += 1

    def inner_loss(self) -> float:
        """loss of the embedding BEFORE the Adam step just taken (what the reference's `loss.item()` reads, :31)"""
        self._loss = float(self.stats[0].item())
        return self._loss

    def outer_end(self):
        self.inner_steps_run.append(self._done)
        self.last_losses.append(self._loss)
        self.out.append(self.param.clone())
        self._run(2)

    def run(self, latents: List[torch.Tensor], uncond: torch.Tensor, num_inner_steps: int, epsilon: float,
            num_outer: Optional[int] = None) -> List[torch.Tensor]:
        """latents: the 51 inversion latents (x_0 .. x_T); uncond [1,77,C].  Returns one [1,77,C] fp32 per timestep."""
        self.begin(latents, uncond)
        n = self.num_steps if num_outer is None else num_outer
        for i in range(n):
            self.outer_begin(i)
            for j in range(num_inner_steps):
                self.inner_step()
                if self.inner_loss() < epsilon + i * 2e-5:
                    break
            self.outer_end()
        return self.out

    def release(self):
        self._graphs = None
        for m in self.unet.attention_modules():
            m.cache_kv = True
            m._kv_key, m._kv = None, None


def run_many(opts: List[NullTextOptimizer], latents_list, uncond_list, num_inner_steps: int, epsilon: float,
             num_outer: Optional[int] = None) -> List[List[torch.Tensor]]:
    """Null-text optimisation of several independent images in flight, each on its own stream.

    An inner iteration is ~900 dependent launches at UNet batch 1 — almost pure dispatch latency — so E images
    stepped in turn fill each other's gaps.  Every image keeps its own early stop; per round the host reads one loss
    per still-active image (the reference's `loss.item()`, one image at a time there).  Same values as `run` per image."""
    cur = torch.cuda.current_stream()
    streams = [torch.cuda.Stream() for _ in opts]
    for o, lat, unc in zip(opts, latents_list, uncond_list):
        o.begin(lat, unc)                       # captures on first use (sequential: graphs share the UNet modules)
    for s in streams:
        s.wait_stream(cur)
    n = min(o.num_steps for o in opts) if num_outer is None else num_outer
    for i in range(n):
        for o, s in zip(opts, streams):
            with torch.cuda.stream(s):
                o.outer_begin(i)
        active = list(range(len(opts)))
        for j in range(num_inner_steps):
            for k in active:
                with torch.cuda.stream(streams[k]):
                    opts[k].inner_step()
            still = []
            for k in active:
                with torch.cuda.stream(streams[k]):
                    if opts[k].inner_loss() >= epsilon + i * 2e-5:
                        still.append(k)
            active = still
            if not active:
                break
        for o, s in zip(opts, streams):
            with torch.cuda.stream(s):
                o.outer_end()
    for s in streams:
        cur.wait_stream(s)
    torch.cuda.synchronize()
    return [o.out for o in opts]
