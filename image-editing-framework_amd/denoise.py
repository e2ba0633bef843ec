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
import os
from typing import Dict, List, Optional

import torch

from . import hip

# Captured step graphs are kept and RE-USED for the next image of the same shape (`acquire` / `release`): capturing
# costs an eager warm-up step plus the capture pass (tens of ms, ~10 % of a 50-step edit) and, with several images
# in flight, is the serial part of the schedule.  IEF_REUSE_GRAPHS=0 captures afresh for every loop.
REUSE_GRAPHS = os.environ.get("IEF_REUSE_GRAPHS", "1") == "1"
_POOL: Dict[tuple, list] = {}
_POOL_CAP = 8


def _check_finite(lat, unet):
    """one host sync per finished loop: the fp16-operand modes ("f16", "f16x3") represent operand elements up to 65504; an
    activation beyond that becomes inf in its hi half and the output NaN -- say so instead of handing NaN images on"""
    if not bool(torch.isfinite(lat).all()):
        mode = getattr(unet, "precision", "f16")
        hint = ("an operand left the fp16 range (|x| > 65504): run this model with precision=\"f32\" (IEF_PRECISION=f32 / --precision f32), "
                "the fp32-input MFMA has the fp32 range" if mode != "f32" else "check the inputs / weights")
        raise FloatingPointError(f"non-finite latents after the denoising loop (precision={mode}): {hint}")


class FusedDenoiser:
    def __init__(self, model, context: torch.Tensor, num_latents: int, latent_hw, guidance_scale: Optional[float],
                 mode: str = "denoise", uncond_list: Optional[List[torch.Tensor]] = None, use_graph: bool = True,
                 added_cond_kwargs=None):
        """context: [2*Bp,77,C] (uncond, cond) for mode 'denoise' with CFG, [Bp,77,C] for 'invert' / no-CFG.
        added_cond_kwargs: SDXL's {"text_embeds" [B, 1280], "time_ids" [B, 6]} (one row per UNet batch row); they are
        constant over the steps, so they fold into the per-step time-embedding rows (then one row PER BATCH ROW)."""
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
        self.step = torch.zeros(1, dtype=torch.int32, device=dev)
        self.coef_cur = torch.zeros(4, dtype=torch.float32, device=dev)
        self.coef_table = self.temb_table = self.temb_cur = self.ctx = self.ctx_table = None
        self._fill(context, guidance_scale, uncond_list, added_cond_kwargs)
        self.use_graph = use_graph
        self.graph = None
        self.plan = self.unet._plan
        self._keep, self._pooled_started, self._pool_key = None, False, None
        self._done = 0          # steps taken since the last rewind (host mirror of the device step counter)

    def _fill(self, context, guidance_scale, uncond_list, added_cond_kwargs=None):
        """(re)compute every table the step graph reads, IN PLACE once the buffers exist (re-use of a captured loop)"""
        dev = self.unet.device
        ts = self.sched.timesteps.tolist()
        if self.mode == "invert":
            ts = ts[::-1]
            coef = [self.sched.reverse_coeffs(t) for t in ts]
        else:
            coef = [self.sched.step_coeffs(t) for t in ts]
        g = float(guidance_scale) if self.cfg else 1.0
        self.num_steps = len(ts)

        def put(name, value):
            cur = getattr(self, name)
            if cur is None:
                setattr(self, name, value)
            else:
                cur.copy_(value)

        put("coef_table", torch.tensor([[a, b, g, 0.0] for a, b in coef], dtype=torch.float32, device=dev))
        aug = self.unet.aug_embedding(added_cond_kwargs)          # None unless the UNet has SDXL's additional embedding
        if aug is not None and aug.shape[0] != self.B:
            raise ValueError(f"added_cond_kwargs batch {aug.shape[0]} != UNet batch {self.B}")
        rows = self.unet.time_rows(torch.tensor(ts, dtype=torch.float32, device=dev), aug)
        nb = 1 if aug is None else self.B
        put("temb_table", rows.reshape(len(ts), nb, -1).contiguous())
        if self.temb_cur is None:
            self.temb_cur = torch.zeros(nb, self.temb_table.shape[2], dtype=torch.float32, device=dev)
        ctx16 = self.unet._act(context.to(dev))               # the model's activation dtype (fp16; fp32 in the exact mode)
        if uncond_list is not None:  # null-text embeddings: the uncond half changes every step
            rows = []
            for u in uncond_list:
                u16 = self.unet._act(u.to(dev)).expand(self.Bp, *ctx16.shape[1:])
                rows.append(torch.cat([u16, ctx16[self.Bp:]], 0))
            put("ctx_table", torch.stack(rows).contiguous())
            if self.ctx is None:
                self.ctx = torch.zeros_like(ctx16)
        else:
            # a COPY: in the fp32-storage modes `_act` hands back the caller's own tensor, and a pooled loop re-pointed at the
            # next image writes into this buffer -- the previous image's context (still held by the caller for its null-text
            # optimisation / edit) must not change under it
            put("ctx", ctx16.clone())

    def _key(self, context, uncond_list):
        plan = self.unet._plan
        return (id(self.unet), self.mode, self.Bp, tuple(self.lat.shape[-2:]), self.cfg, tuple(context.shape),
                None if uncond_list is None else len(uncond_list), len(self.sched.timesteps),
                None if plan is None else plan.signature(self.unet))

    def rebind(self, context, guidance_scale, uncond_list, added_cond_kwargs=None):
        """point a captured loop at the next image: new tables, new cross-attention K/V, the new controller's plan"""
        self._fill(context, guidance_scale, uncond_list, added_cond_kwargs)
        if self.ctx_table is None:          # K/V of the fixed context live in tensors the graph reads: refresh in place, by the
            # same kernels the forward used (the model's contraction mode: "x3" and "f32" differ in the last bits, and an
            # image must not depend on whether its loop was captured for it or re-pointed at it)
            with hip.f32_contraction(getattr(self.unet, "contract", "f32")):
                for m, kv in self._keep:
                    hip.gemm(self.ctx, m.w_kv, out=kv)
        fresh = self.unet._plan
        if self.plan is not None:
            self.plan.load_from(fresh, self.B)
            self.plan.sync_step()
        self.step.zero_()
        self._done = 0

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
        used0 = hip.counters_used()
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
        # the split-K launches of this graph combine their slabs inside the launch and need arrival counters nobody else
        # touches while the graph may run: an arena of the size the warm-up pass used, owned by this loop
        self._arena = hip.counter_arena(hip.counters_used() - used0, self.lat.device)
        self.graph = torch.cuda.CUDAGraph()
        with self._arena, torch.cuda.graph(self.graph):
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
        self._done = 0
        if self.use_graph and self.graph is None:
            self._capture()
        elif not self.use_graph:
            self._set_kv_cache(self.ctx_table is None)
        # tensors the graph READS but does not own (cross-attention K/V projected during the warm-up): keep them
        # alive for as long as this loop may be replayed, whatever another loop's warm-up caches afterwards
        if getattr(self, "_keep", None) is None or not self._pooled_started:
            self._keep = [(m, m._kv) for m in self.unet.attention_modules() if m.is_cross and m._kv is not None]
            self._pooled_started = True
        if self.plan is not None and self.graph is not None:
            self.plan.sync_step()

    def rewind(self, latents: Optional[torch.Tensor] = None):
        """back to step 0 of the schedule (device and host counters, the plan's counter from its controller)"""
        if latents is not None:
            self.lat.copy_(latents.to(self.lat.device).float().expand_as(self.lat))
        self.step.zero_()
        self._done = 0
        if self.plan is not None:
            self.plan.sync_step()

    def step_once(self):
        # the per-step tables have num_steps rows: a replay past them would re-read the last row (the kernel clamps the row
        # index) and silently apply the final timestep again
        if self._done >= self.num_steps:
            raise IndexError(f"step {self._done} is past the {self.num_steps}-step schedule this loop was built for; "
                             "rewind() or start() it first")
        self._done += 1
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
        _check_finite(out, self.unet)
        return (out, traj) if keep_all else out

    def release(self):
        key = getattr(self, "_pool_key", None)
        if REUSE_GRAPHS and key is not None and self.graph is not None and len(_POOL.setdefault(key, [])) < _POOL_CAP:
            _POOL[key].append(self)          # keep the captured graph for the next image of this shape
            return
        if self.plan is not None:
            self.plan.captured = False
        self.graph = None


def cfg_split_rows(context: torch.Tensor, num_latents: int, half: int) -> torch.Tensor:
    """rows of the CFG context [uncond..., cond...] that rank `half` of a 2-GPU CFG split runs (0: unconditional)"""
    if context.shape[0] != 2 * num_latents or half not in (0, 1):
        raise ValueError("cfg_split_rows: context must be [2 * num_latents, ...] and half 0 or 1")
    return context[:num_latents] if half == 0 else context[num_latents:]


def exchange_eps(eps_all: torch.Tensor, mine: torch.Tensor, group=None):
    """the ONE exchange of a CFG-split step: every rank contributes its half of eps ([Bp,4,h,w] fp32, 64 KiB per latent at
    64x64) and receives both, rank order = [unconditional, conditional] (`/root/reference/p2p/model/sd_utils.py:74`: the
    chunk(2) the CFG combine reads).  RCCL all-gather on device buffers; on a gloo group (CPU tests, two ranks sharing one
    GPU) the halves are staged through host memory."""
    import torch.distributed as dist
    if dist.get_backend(group) == "gloo":
        parts = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, mine.detach().cpu().contiguous(), group=group)
        eps_all.copy_(torch.cat(parts).to(eps_all.device))
    else:
        dist.all_gather_into_tensor(eps_all, mine.contiguous(), group=group)
    return eps_all


class CfgSplitDenoiser(FusedDenoiser):
    """ONE edit on TWO GPUs: the CFG batch [uncond_src, uncond_tgt | cond_src, cond_tgt] has no data
    dependence between its halves inside the UNet — controllers act on the conditional rows only
    (`/root/reference/p2p/model/attention_base.py:20-22`) — so rank 0 of the pair runs the unconditional rows, rank 1
    the conditional rows with the controller's plan, and each step needs one exchange: the two eps halves (128 KiB at
    64x64 latents) are all-gathered before the CFG combine (`sd_utils.py:74-75`).  Both ranks then apply the same fused
    CFG + DDIM update to their own copy of the latents, which therefore stay identical without a second exchange.

    Per step: captured graph A {row selects, UNet on this rank's rows} -> all-gather -> captured graph B {CFG + DDIM
    update, advance}.  The controller must be registered with `rows="uncond"` / `rows="cond"` on the two ranks
    (`p2p.model.register.register_attention_control`); its counters advance on both."""

    def __init__(self, model, context, num_latents, latent_hw, guidance_scale: float, group=None, uncond_list=None,
                 use_graph: bool = True):
        import torch.distributed as dist
        if dist.get_world_size(group) != 2:
            raise ValueError("a CFG split runs on a group of exactly two ranks")
        self.group = group
        self.half = dist.get_rank(group)
        self._guidance = float(guidance_scale)
        plan = model.unet._plan
        want = num_latents
        if plan is not None and plan.kind == "p2p" and (not plan.cond_only or self.half != 1 or plan.batch != want):
            raise ValueError('CFG split: register the controller with rows="cond" on rank 1 and rows="uncond" on rank 0')
        # every other plan kind (MasaCtrl, Plug-and-Play) and every Python hook was built for the FULL [uncond..., cond...]
        # batch: on a half batch its row indirections would point at the wrong or at missing rows without any error
        if plan is not None and plan.kind not in ("p2p", "empty"):
            raise ValueError(f"CFG split: a '{plan.kind}' control plan addresses the full CFG batch; only Prompt-to-Prompt "
                             'controllers registered with rows="cond" / "uncond" (or no controller) can be split')
        if plan is None and not all(m.is_native() for m in model.unet.attention_modules()):
            raise ValueError("CFG split: an attention hook is installed that was written for the full CFG batch")
        super().__init__(model, cfg_split_rows(context, num_latents, self.half), num_latents, latent_hw, None, mode="denoise",
                         uncond_list=uncond_list if self.half == 0 else None, use_graph=use_graph)
        C = self.unet.config.in_channels
        h, w = latent_hw
        self.eps_all = torch.zeros(2 * self.Bp, C, h, w, dtype=torch.float32, device=self.lat.device)
        self.eps_mine = torch.zeros(self.Bp, C, h, w, dtype=torch.float32, device=self.lat.device)
        self.graph_b = None

    def _fill(self, context, guidance_scale, uncond_list, added_cond_kwargs=None):
        super()._fill(context, None, uncond_list, added_cond_kwargs)
        self.coef_table[:, 2] = self._guidance          # the parent wrote 1.0 (no CFG inside ITS step)

    # ---- the two captured halves of a step
    def _forward_part(self):
        hip.select_step(self.temb_table, self.temb_cur, self.step)
        hip.select_step(self.coef_table, self.coef_cur, self.step)
        if self.ctx_table is not None:
            hip.select_step(self.ctx_table, self.ctx, self.step)
        eps = self.unet(self.lat, encoder_hidden_states=self.ctx, temb_row=self.temb_cur)["sample"]
        self.eps_mine.copy_(eps)

    def _update_part(self):
        hip.cfg_ddim_step(self.eps_all[: self.Bp], self.eps_all[self.Bp:], self.lat, self.coef_cur, out=self.lat)
        hip.advance_step(self.step)

    def _step_body(self):
        self._forward_part()
        exchange_eps(self.eps_all, self.eps_mine, self.group)
        self._update_part()

    def _capture(self):
        plan = self.plan
        self._set_kv_cache(self.ctx_table is None)
        saved = self.lat.clone()
        if plan is not None:
            plan.prepare(self.B)
            plan.muted = True
        used0 = hip.counters_used()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._forward_part()            # warm-up of the forward half only: no collective on a side stream
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if plan is not None:
            plan.muted = False
        self.step.zero_()
        if plan is not None:
            plan.sync_step()
            plan.captured = True
        self._arena = hip.counter_arena(hip.counters_used() - used0, self.lat.device)
        self.graph = torch.cuda.CUDAGraph()
        with self._arena, torch.cuda.graph(self.graph):
            self._forward_part()
        self.graph_b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_b):
            self._update_part()
        self.lat.copy_(saved)
        self.step.zero_()
        if plan is not None:
            plan.sync_step()

    def step_once(self):
        if self._done >= self.num_steps:
            raise IndexError(f"step {self._done} is past the {self.num_steps}-step schedule this loop was built for")
        self._done += 1
        if self.graph is not None:
            self.graph.replay()
            exchange_eps(self.eps_all, self.eps_mine, self.group)
            self.graph_b.replay()
            if self.plan is not None:
                self.plan.replay_done()
        else:
            self._step_body()

    def release(self):
        if self.plan is not None:
            self.plan.captured = False
        self.graph = self.graph_b = None


def acquire(model, context, num_latents, latent_hw, guidance_scale, mode="denoise", uncond_list=None,
            use_graph=True, added_cond_kwargs=None) -> FusedDenoiser:
    """a FusedDenoiser for this job: a pooled one with a captured graph of the same shape / plan signature, re-pointed
    at the new context, tables and controller — or a new one"""
    if REUSE_GRAPHS and use_graph:
        probe = FusedDenoiser.__new__(FusedDenoiser)
        probe.unet, probe.sched, probe.mode = model.unet, model.scheduler, mode
        probe.cfg, probe.Bp = guidance_scale is not None, num_latents
        probe.lat = torch.empty(0, 0, *latent_hw)
        key = probe._key(context, uncond_list)
        free = _POOL.get(key)
        if free:
            loop = free.pop()
            loop.rebind(context, guidance_scale, uncond_list, added_cond_kwargs)
            return loop
        loop = FusedDenoiser(model, context, num_latents, latent_hw, guidance_scale, mode, uncond_list, use_graph,
                             added_cond_kwargs)
        loop._pool_key = key
        return loop
    return FusedDenoiser(model, context, num_latents, latent_hw, guidance_scale, mode, uncond_list, use_graph,
                         added_cond_kwargs)


def drop_pool():
    """free every pooled graph (tests; before changing weights)"""
    for loops in _POOL.values():
        for l in loops:
            l.graph = None
    _POOL.clear()


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
