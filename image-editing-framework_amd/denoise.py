"""The denoising / inversion hot loop as ONE captured hipGraph replayed once per step.

One step of the reference's loop (`/root/reference/p2p/model/sd_utils.py:67-79`) is
    cat([latents]*2) -> UNet -> chunk -> CFG -> scheduler.step -> controller.step_callback
and one inversion step (`/root/reference/p2p/inversion/ddim.py:27-31`) is UNet (cond only) ->
`ddim_reverse`.  Eagerly that is ~430 kernel launches per step driven from Python; the launch
overhead would exceed the GPU time.  Here the whole step is captured once:

    select_step(time-embedding row)  select_step(DDIM coefficients)     <- read a DEVICE step counter
    [P2P plan: select_step(gate coefficients), select_step(self-replace sources)]
    latents -> CFG batch copy -> UNet (fused attention control) -> fused CFG + DDIM update (in place)
    advance_step

so a 50-step edit is 50 graph replays with no host work in between; everything that varies
with the step is a per-step table row selected on the device.  Static buffers: latents (fp32
NCHW), the CFG batch, the fp16 context (per-step rows when null-text embeddings are supplied).
"""
from typing import List, Optional

import torch

from . import hip


class FusedDenoiser:
    def __init__(self, model, context: torch.Tensor, num_latents: int, latent_hw, guidance_scale: Optional[float],
                 mode: str = "denoise", uncond_list: Optional[List[torch.Tensor]] = None, use_graph: bool = True):
        """context: [2*Bp,77,C] (uncond, cond) for mode 'denoise' with CFG, [Bp,77,C] for 'invert' / no-CFG."""
        self.model, self.unet, self.sched = model, model.unet, model.scheduler
        dev = self.unet.device
        self.mode = mode
        self.cfg = guidance_scale is not None
        self.Bp = num_latents
        self.B = 2 * num_latents if self.cfg else num_latents
        if context.shape[0] != self.B:
            raise ValueError(f"context batch {context.shape[0]} != UNet batch {self.B}")
        h, w = latent_hw
        C = self.unet.config.in_channels
        self.lat = torch.zeros(self.Bp, C, h, w, dtype=torch.float32, device=dev)
        self.lat_in = torch.zeros(self.B, C, h, w, dtype=torch.float32, device=dev) if self.cfg else self.lat
        ts = self.sched.timesteps.tolist()
        if mode == "invert":
            ts = ts[::-1]
            coef = [self.sched.reverse_coeffs(t) for t in ts]
        else:
            coef = [self.sched.step_coeffs(t) for t in ts]
        g = float(guidance_scale) if self.cfg else 1.0
        self.num_steps = len(ts)
        self.coef_table = torch.tensor([[a, b, g, 0.0] for a, b in coef], dtype=torch.float32, device=dev)
        self.coef_cur = torch.zeros(4, dtype=torch.float32, device=dev)
        self.temb_table = self.unet.time_rows(torch.tensor(ts, dtype=torch.float32, device=dev)).contiguous()
        self.temb_cur = torch.zeros(1, self.temb_table.shape[1], dtype=torch.float32, device=dev)
        self.step = torch.zeros(1, dtype=torch.int32, device=dev)
        ctx16 = hip.to_f16(context.to(dev).float().contiguous())
        self.ctx_table = None
        if uncond_list is not None:  # null-text embeddings: the uncond half changes every step
            rows = []
            for u in uncond_list:
                u16 = hip.to_f16(u.to(dev).float().contiguous()).expand(self.Bp, *ctx16.shape[1:])
                rows.append(torch.cat([u16, ctx16[self.Bp:]], 0))
            self.ctx_table = torch.stack(rows).contiguous()
            self.ctx = torch.zeros_like(ctx16)
        else:
            self.ctx = ctx16
        self.use_graph = use_graph
        self.graph = None
        self.plan = self.unet._plan

    # ------------------------------------------------------------------ one step, stream-ordered
    def _step_body(self):
        hip.select_step(self.temb_table, self.temb_cur, self.step)
        hip.select_step(self.coef_table, self.coef_cur, self.step)
        if self.ctx_table is not None:
            hip.select_step(self.ctx_table, self.ctx, self.step)
        if self.cfg:
            self.lat_in[: self.Bp].copy_(self.lat)
            self.lat_in[self.Bp:].copy_(self.lat)
        eps = self.unet(self.lat_in, encoder_hidden_states=self.ctx, temb_row=self.temb_cur)["sample"]
        if self.cfg:
            hip.cfg_ddim_step(eps[: self.Bp], eps[self.Bp:], self.lat, self.coef_cur, out=self.lat)
        else:
            hip.cfg_ddim_step(None, eps, self.lat, self.coef_cur, out=self.lat)
        hip.advance_step(self.step)

    def _set_kv_cache(self, on: bool):
        for m in self.unet.attention_modules():
            m.cache_kv = on

    def _capture(self):
        plan = self.plan
        # the context K/V are step-invariant unless null-text rows are swapped in per step
        self._set_kv_cache(self.ctx_table is None)
        # warm-up outside the graph with the control plan muted (fills the K/V cache and the
        # allocator pools without moving any counter), on a side stream as capture requires
        saved = self.lat.clone()
        if plan is not None:
            plan.prepare(self.B)
            plan.muted = True
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._step_body()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if plan is not None:
            plan.muted = False
        self.lat.copy_(saved)
        self.step.zero_()
        if plan is not None:
            plan.sync_step()
            plan.captured = True
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._step_body()
        self.lat.copy_(saved)
        self.step.zero_()
        if plan is not None:
            plan.sync_step()

    # ------------------------------------------------------------------ public
    def start(self, latents: torch.Tensor):
        """load the initial latents [Bp,C,h,w] (or [1,...] broadcast), rewind, capture the step graph if needed.
        Must run while this loop's controller (if any) is registered on the UNet; stepping afterwards does not
        need the registration, so several started loops can be stepped in turn (`run_interleaved`)."""
        self.lat.copy_(latents.to(self.lat.device).float().expand_as(self.lat))
        self.step.zero_()
        if self.use_graph and self.graph is None:
            self._capture()
        elif not self.use_graph:
            self._set_kv_cache(self.ctx_table is None)
        # tensors the graph READS but does not own (cross-attention K/V projected during the warm-up): keep them
        # alive for as long as this loop may be replayed, whatever another loop's warm-up caches afterwards
        self._keep = [m._kv for m in self.unet.attention_modules()]

    def step_once(self):
        if self.graph is not None:
            self.graph.replay()
            if self.plan is not None:
                self.plan.replay_done()
        else:
            self._step_body()

    def result(self) -> torch.Tensor:
        return self.lat.clone()

    def run(self, latents: torch.Tensor, num_steps: Optional[int] = None, keep_all: bool = False):
        """latents [Bp,C,h,w] (or [1,...] broadcast) -> final latents (and the trajectory if keep_all)."""
        self.start(latents)
        n = self.num_steps if num_steps is None else num_steps
        traj = [self.lat.clone()] if keep_all else None
        for _ in range(n):
            self.step_once()
            if keep_all:
                traj.append(self.lat.clone())
        out = self.lat.clone()
        return (out, traj) if keep_all else out

    def release(self):
        if self.plan is not None:
            self.plan.captured = False
        self.graph = None


def run_interleaved(loops: List[FusedDenoiser], num_steps: Optional[int] = None):
    """Step several STARTED loops (independent images) in turn, each on its own stream.

    One edit step is ~380 dependent launches of which each pays ~4 us of dispatch latency, so a single chain leaves
    the GPU idle a fifth of the time; with E chains in flight those gaps are filled by the other chains' kernels
    (measured on SD1.5 shapes, UNet batch 4: 7.7 ms per step alone, 6.4 / 5.8 / 5.6 ms per step with 2 / 3 / 4 in
    flight).  Results are those of running the loops one after the other."""
    if not loops:
        return
    cur = torch.cuda.current_stream()
    streams = [torch.cuda.Stream() for _ in loops]
    for s in streams:
        s.wait_stream(cur)
    n = min(l.num_steps for l in loops) if num_steps is None else num_steps
    for _ in range(n):
        for loop, s in zip(loops, streams):
            with torch.cuda.stream(s):
                loop.step_once()
    for s in streams:
        cur.wait_stream(s)
