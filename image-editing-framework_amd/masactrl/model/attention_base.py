"""Editor protocol of MasaCtrl (`/root/reference/masactrl/model/attention_base.py:5-31`).

    out = editor(q, k, v, sim, attn, is_cross, place_in_unet, num_heads, scale=...)   # -> [B, N, heads*d]
    editor.cur_step / cur_att_layer / num_att_layers, editor.after_step(), editor.reset()

q, k, v are [(B*heads), N, d]; `attn` the materialised softmax map.  The Python bodies below are the reference
semantics on whatever tensors they are given (used by the CPU tests against fixture G7).  On the GPU the two
editor classes of this package are lowered to a device plan instead (`register.py`), so no map is built.
"""
import abc

import torch


class AttentionBase(abc.ABC):
    def __init__(self):
        self.cur_step = 0
        self.num_att_layers = -1
        self.cur_att_layer = 0

    def after_step(self):
        pass

    def __call__(self, q, k, v, sim, attn, is_cross, place_in_unet, num_heads, **kwargs):
        out = self.forward(q, k, v, sim, attn, is_cross, place_in_unet, num_heads, **kwargs)
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1
            self.after_step()
        return out

    def forward(self, q, k, v, sim, attn, is_cross, place_in_unet, num_heads, **kwargs):
        out = torch.bmm(attn, v)                                   # (b h) n d
        bh, n, d = out.shape
        b = bh // num_heads
        return out.reshape(b, num_heads, n, d).permute(0, 2, 1, 3).reshape(b, n, num_heads * d)

    def reset(self):
        self.cur_step = 0
        self.cur_att_layer = 0


class AttentionStore(AttentionBase):
    """The map-collecting editor the reference's scripts import (`/root/reference/masactrl/edit_syn.py:7`; class at
    `masactrl/model/attention_base.py:33-66`; never instantiated upstream).  Per call: maps of up to 64^2 queries are put on
    the step's self / cross list, the attention output is the plain `AttentionBase.forward`.  After a step whose (already
    advanced) counter lies strictly between `min_step` and `max_step`, `valid_steps` grows and the step's maps are folded
    into the running store.

    Kept quirk (fixture G14, made by the reference's class): on the first valid step the running store is BOUND to the step
    list itself, and the step lists are cleared at the end of every step — so the running lists are emptied with them and
    every later valid step re-binds instead of adding.  A caller reading `self_attns` between steps sees what the reference
    shows: empty lists, `valid_steps` counting.  `reset()` leaves `valid_steps` alone, as upstream.

    An unknown editor class to the lowering in `register.py`: it runs on the generic path (materialised `sim` / `attn`)."""

    def __init__(self, res=[32], min_step=0, max_step=1000):
        super().__init__()
        self.res, self.min_step, self.max_step = res, min_step, max_step
        self.valid_steps = 0
        self.self_attns, self.cross_attns = [], []                 # running store
        self.self_attns_step, self.cross_attns_step = [], []       # maps of the step in progress

    def after_step(self):
        if self.min_step < self.cur_step < self.max_step:
            self.valid_steps += 1
            if not self.self_attns:
                self.self_attns, self.cross_attns = self.self_attns_step, self.cross_attns_step        # bound, not copied
            else:
                for kept, new in zip(self.self_attns, self.self_attns_step):
                    kept += new
                for kept, new in zip(self.cross_attns, self.cross_attns_step):
                    kept += new
        self.self_attns_step.clear()
        self.cross_attns_step.clear()

    def forward(self, q, k, v, sim, attn, is_cross, place_in_unet, num_heads, **kwargs):
        if attn.shape[1] <= 64 ** 2:
            (self.cross_attns_step if is_cross else self.self_attns_step).append(attn)
        return super().forward(q, k, v, sim, attn, is_cross, place_in_unet, num_heads, **kwargs)
