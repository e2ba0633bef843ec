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
