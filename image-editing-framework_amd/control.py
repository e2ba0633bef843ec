"""Device-side control plan: a known controller lowered to data the fused attention kernels read.

The reference hands every materialised attention map to Python
(`/root/reference/p2p/model/register.py:47-48`).  For the controller classes whose arithmetic is
known (`EmptyControl`, `AttentionReplace / Refine / Reweight`; MasaCtrl's mutual self-attention)
the same effect is expressed as small device tables, so no map is ever written to HBM:

  cross-attention edit (attention_base.py:118-121, attention_control.py:15-46)
      P' = c1[step][w] * (P_src @ M)[w] + c2[step][w] * P_tgt[w]
      M^T fp16 [slots,96,96], coef table fp32 [steps+1, slots, 2, 96]
  self-attention replace (attention_base.py:123,132-136), only for maps with N <= 16*16 and
      num_self_replace[0] <= step < num_self_replace[1]:
      target rows take the source row's Q and K (=> identical map), keep their own V
      source-row table int32 [steps+1, 2B]  (identity rows outside the window)
  MasaCtrl mutual self-attention (/root/reference/masactrl/model/attention_control.py:37-68):
      K,V source rows per (step, layer)
  Plug-and-Play injection (/root/reference/pnp/model/register.py:27-90,100-182), batch = 4 blocks of s rows
      [uncond_src, uncond_tgt, cond_src, cond_tgt]: during the first qk_steps timesteps the self-attention of the chosen
      decoder layers computes rows of blocks 1 and 3 with the Q and K of block 2 (:45-52), and during the first
      conv_steps timesteps `up_blocks[1].resnets[1]` replaces their conv2 output by block 2's (:161-166):
      a Q/K source-row table and a feature source-row table, int32 [steps+1, B]

The step index lives in device memory (`step`), the per-step rows are copied into fixed
"current" buffers by `ief_select_step` at the start of every UNet forward, and the counter is
bumped by `ief_advance_step` at its end — all stream-ordered kernels, so ONE captured hipGraph
replays correctly for all 50 steps.  The Python controller's own counters (`cur_step`,
`cur_att_layer`, `between_steps()`) are advanced exactly as `AttentionControl.__call__` does
(attention_base.py:23-27) so user code observing them sees reference behaviour.
"""
from typing import Optional

import torch

from . import hip

XL = 96  # padded key count of the cross-attention kernel


def _step_hook(c):
    # P2P controllers call it between_steps (p2p/model/attention_base.py:27), MasaCtrl editors after_step
    # (masactrl/model/attention_base.py:21)
    fn = getattr(c, "between_steps", None) or getattr(c, "after_step", None)
    if fn is not None:
        fn()


def advance_controller(c):
    """`attention_base.py:23-27` for any object following the controller protocol."""
    c.cur_att_layer += 1
    uncond = c.num_att_layers if getattr(c, "LOW_RESOURCE", False) else 0
    if c.cur_att_layer == c.num_att_layers + uncond:
        c.cur_att_layer = 0
        c.cur_step += 1
        _step_hook(c)


class StepCounter:
    """the counters of `AttentionControl` (attention_base.py:10-27) for plans that have no controller object (PnP)"""

    def __init__(self, num_att_layers: int = 0):
        self.cur_step, self.cur_att_layer, self.num_att_layers = 0, 0, num_att_layers

    def between_steps(self):
        pass

    def reset(self):
        self.cur_step, self.cur_att_layer = 0, 0


class ControlPlan:
    """kind: 'empty' | 'p2p' | 'masactrl' | 'pnp'"""

    def __init__(self, controller, kind: str, device, num_prompts: int = 1, num_steps: int = 0,
                 mt: Optional[torch.Tensor] = None, coef_table: Optional[torch.Tensor] = None,
                 self_window=(0, 0), self_max_tokens: int = 256, masa_steps=(), masa_layers=(),
                 pnp_layers=(), pnp_qk_steps: int = 0, pnp_conv_steps: int = 0, cond_only: bool = False):
        """cond_only: the UNet batch holds ONLY the conditional rows [cond_src, cond_tgt...] — the half a controller acts
        on (`attention_base.py:20-22`) — as on the conditional rank of a 2-GPU CFG split (`denoise.CfgSplitDenoiser`) and
        in the reference's LOW_RESOURCE protocol (:18-19)"""
        self.controller = controller
        self.kind = kind
        self.device = torch.device(device)
        self.num_prompts = num_prompts
        self.cond_only = bool(cond_only)
        self.batch = num_prompts if cond_only else 2 * num_prompts
        self.num_steps = num_steps
        self.self_window = self_window
        self.self_max_tokens = self_max_tokens
        self.captured = False   # True while a hipGraph replays the forward (Python bookkeeping moves to the replay wrapper)
        self.muted = False      # True during a warm-up forward that must not move any counter
        dev = self.device
        self.step = torch.zeros(1, dtype=torch.int32, device=dev)
        B, Bp = self.batch, num_prompts
        if kind == "p2p":
            slots = Bp - 1
            assert mt.shape == (slots, XL, XL) and coef_table.shape == (num_steps + 1, slots, 2, XL)
            self.mt = mt.to(device=dev, dtype=torch.float16).contiguous()
            self.mt32 = mt.to(device=dev, dtype=torch.float32).contiguous()      # the reference-precision kernels' copy
            self.coef_table = coef_table.to(device=dev, dtype=torch.float32).contiguous()
            self.coef_cur = torch.zeros(slots, 2, XL, dtype=torch.float32, device=dev)
            # the edited maps P' = c1 T + c2 P reach max (|c1| + |c2|) (AttentionReweight: c1 = alpha * equalizer): the
            # split-operand kernels size the hi / lo split of P' from it (hip.map_split_scale); the scale is a launch
            # argument, hence part of the signature a pooled step graph is matched on
            ct = coef_table.detach().float()
            self.coef_bound = float((ct[:, :, 0].abs() + ct[:, :, 1].abs()).max()) if ct.numel() else 1.0
            off = 0 if cond_only else Bp          # first conditional row (= the source prompt's) of this UNet batch
            es = torch.full((B,), -1, dtype=torch.int32)
            sl = torch.zeros(B, dtype=torch.int32)
            for k in range(slots):
                es[off + 1 + k] = off
                sl[off + 1 + k] = k
            self.edit_src, self.edit_slot = es.to(dev), sl.to(dev)
            ident = torch.arange(B, dtype=torch.int32)
            repl = ident.clone()
            repl[off + 1:] = off
            tab = ident.repeat(num_steps + 1, 1)
            lo, hi = self_window
            tab[lo:hi] = repl
            self.self_table = tab.contiguous().to(dev)
            self.self_cur = ident.clone().to(dev)
        # MasaCtrl mutual self-attention (/root/reference/masactrl/model/attention_control.py:52-66): in the chosen
        # transformer layers and steps every row of the uncond half attends to K,V of that half's FIRST row (the
        # source image), likewise for the cond half.  The batch size is not known at registration: tables per B.
        self.masa_steps = set(int(s) for s in masa_steps)
        self.masa_layers = set(int(l) for l in masa_layers)
        self._masa = {}
        self._step_synced = -1
        self.pnp_layers = set(pnp_layers)          # id() of the Attention modules whose Q/K are injected
        self.pnp_qk_steps, self.pnp_conv_steps = int(pnp_qk_steps), int(pnp_conv_steps)
        self._pnp = {}                             # B -> (qk_table, qk_cur, conv_table, conv_cur)

    def prepare(self, B: int):
        """allocate per-batch device tables OUTSIDE any graph capture"""
        if self.kind == "masactrl" and B not in self._masa:
            n = (max(self.masa_steps) + 2) if self.masa_steps else 1
            self.num_steps = max(self.num_steps, n - 1)
            ident = torch.arange(B, dtype=torch.int32)
            src = ident.clone()
            half = B // 2
            if half > 0:
                src[:half] = 0
                src[half:] = half
            tab = ident.repeat(n, 1)
            for st in self.masa_steps:
                tab[st] = src
            self._masa[B] = (tab.contiguous().to(self.device), ident.clone().to(self.device))
        if self.kind == "pnp" and B not in self._pnp:
            ident = torch.arange(B, dtype=torch.int32)
            inj = ident.clone()
            s = B // 4
            if s > 0 and B == 4 * s:
                inj[s:2 * s] = ident[2 * s:3 * s]
                inj[3 * s:4 * s] = ident[2 * s:3 * s]
            n = self.num_steps + 1
            qk, cv = ident.repeat(n, 1), ident.repeat(n, 1)
            qk[: self.pnp_qk_steps] = inj
            cv[: self.pnp_conv_steps] = inj
            dev = self.device
            self._pnp[B] = (qk.contiguous().to(dev), ident.clone().to(dev), cv.contiguous().to(dev), ident.clone().to(dev))
    # ------------------------------------------------------------------ re-use of a captured loop (denoise.acquire)
    def signature(self, unet):
        """everything about this plan that is BAKED into a captured step graph (which kernels run, on which modules,
        with tables of which shape); two plans with equal signatures differ only in table contents"""
        if self.kind == "p2p":
            return ("p2p", self.num_prompts, self.num_steps, self.self_max_tokens, tuple(self.mt.shape), self.cond_only,
                    hip.map_split_scale(self.coef_bound))
        if self.kind == "masactrl":
            return ("masactrl", tuple(sorted(self.masa_layers)), (max(self.masa_steps) + 2) if self.masa_steps else 1)
        if self.kind == "pnp":
            idx = {id(m): m._exec_index for m in unet.attention_modules()}
            # the injected resnet as `pnp/model/register.py:_conv_module` picks it: resnets[0] on the SDXL family, else [1]
            inj = (unet.up_blocks[1].resnets[0 if unet.cfg.addition_embed else 1]._inject is self
                   if len(unet.up_blocks) > 1 else False)
            return ("pnp", tuple(sorted(idx[i] for i in self.pnp_layers)), self.num_steps, bool(inj))
        return (self.kind,)

    def load_from(self, other: "ControlPlan", B: int):
        """take over `other`'s controller and table CONTENTS (same signature): the captured graph keeps reading this
        plan's device tensors"""
        self.controller = other.controller
        if self.kind == "p2p":
            self.mt.copy_(other.mt)
            self.mt32.copy_(other.mt32)
            self.coef_table.copy_(other.coef_table)
            self.coef_bound = other.coef_bound       # same split scale (the signatures matched), possibly another bound below it
            self.self_table.copy_(other.self_table)
            self.self_window = other.self_window
        elif self.kind == "masactrl":
            other.prepare(B)
            self.masa_steps = set(other.masa_steps)
            self._masa[B][0].copy_(other._masa[B][0])
        elif self.kind == "pnp":
            other.prepare(B)
            self.pnp_qk_steps, self.pnp_conv_steps = other.pnp_qk_steps, other.pnp_conv_steps
            self._pnp[B][0].copy_(other._pnp[B][0])
            self._pnp[B][2].copy_(other._pnp[B][2])

    # ------------------------------------------------------------------ per-forward protocol
    def applies(self, B: int) -> bool:
        if self.kind == "empty" or self.muted:
            return False
        if self.kind in ("masactrl", "pnp"):
            return True
        if B != self.batch:
            raise RuntimeError(
                f"controller was built for {self.num_prompts} prompts (UNet batch {self.batch}) but the UNet was "
                f"called with batch {B}; the reference's AttentionControlEdit.forward would mis-reshape here")
        return True

    def begin_forward(self, B: int):
        if not self.applies(B):
            return
        if not self.captured:
            # keep the device counter equal to the controller's (covers reset() / manual edits)
            cs = int(self.controller.cur_step)
            if cs > self.num_steps and self.kind == "p2p":
                raise IndexError(f"cur_step {cs} exceeds the controller's {self.num_steps}-step tables")
            self.step.fill_(cs)
        if self.kind == "p2p":
            hip.select_step(self.coef_table, self.coef_cur, self.step)
            hip.select_step(self.self_table, self.self_cur, self.step)
        elif self.kind == "masactrl":
            if B not in self._masa:
                if self.captured:
                    raise RuntimeError("ControlPlan.prepare(B) must run before graph capture")
                self.prepare(B)
            tab, cur = self._masa[B]
            if not self.captured and int(self.controller.cur_step) >= tab.shape[0]:
                self.step.fill_(tab.shape[0] - 1)   # past the last controlled step: identity row
            hip.select_step(tab, cur, self.step)
        elif self.kind == "pnp":
            if B not in self._pnp:
                if self.captured:
                    raise RuntimeError("ControlPlan.prepare(B) must run before graph capture")
                self.prepare(B)
            qk, qk_cur, cv, cv_cur = self._pnp[B]
            if not self.captured and int(self.controller.cur_step) >= qk.shape[0]:
                self.step.fill_(qk.shape[0] - 1)    # past the schedule: identity rows
            hip.select_step(qk, qk_cur, self.step)
            hip.select_step(cv, cv_cur, self.step)

    def end_forward(self, B: int):
        if self.kind in ("masactrl", "pnp") and not self.muted:
            hip.advance_step(self.step)
        elif self.kind != "empty" and not self.muted and B == self.batch:
            hip.advance_step(self.step)

    def layer_done(self, attn):
        if not self.captured and not self.muted and self.controller is not None:
            advance_controller(self.controller)

    def sync_step(self):
        """device step counter <- controller.cur_step (before the first replay of a captured loop)"""
        self.step.fill_(int(self.controller.cur_step))

    def replay_done(self):
        """one whole forward was replayed from a graph: num_att_layers controller calls happened"""
        c = self.controller
        if c is not None:
            c.cur_step += 1
            _step_hook(c)

    # ------------------------------------------------------------------ what the kernels read
    def self_sources(self, B: int, N: int, attn):
        if self.kind == "p2p" and self.applies(B) and N <= self.self_max_tokens:
            return self.self_cur, self.self_cur, None
        if self.kind == "masactrl" and not self.muted and (attn._exec_index // 2) in self.masa_layers:
            cur = self._masa[B][1]
            return None, cur, cur
        if self.kind == "pnp" and not self.muted and id(attn) in self.pnp_layers:
            cur = self._pnp[B][1]
            return cur, cur, None
        return None, None, None

    def feature_source(self, B: int):
        """source rows of the Plug-and-Play feature injection for the CURRENT step (identity outside its schedule)"""
        if self.kind == "pnp" and not self.muted and B in self._pnp:
            return self._pnp[B][3]
        return None

    def cross_edit(self, B: int, attn):
        if self.kind == "p2p" and self.applies(B):
            mt = self.mt32 if attn.to_q.weight.dtype == torch.float32 else self.mt
            args = dict(edit_src=self.edit_src, edit_slot=self.edit_slot, mt=mt, coef=self.coef_cur)
            if attn.to_q.weight.dtype == torch.float32:
                args["coef_bound"] = self.coef_bound
            return args
        return {}
