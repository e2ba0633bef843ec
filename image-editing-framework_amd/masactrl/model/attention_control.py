"""Mutual self-attention control (`/root/reference/masactrl/model/attention_control.py:10-68`).

From `start_step` on, in transformer layers >= `start_layer` (layer = cur_att_layer // 2, counted in execution order:
6 down, 1 mid, 9 up for SD), every SELF-attention of the uncond half uses the K and V of that half's first sample
(the source image), and likewise for the cond half (:59-66).  Only this class is reachable from the reference CLIs
(`masactrl/edit_syn.py:108`, `edit_real.py:136`); the Union / Mask / MaskAuto variants are never instantiated.
"""
import torch

from .attention_base import AttentionBase


class MutualSelfAttentionControl(AttentionBase):
    MODEL_TYPE = {"SD": 16, "SDXL": 70}

    def __init__(self, start_step=4, start_layer=10, layer_idx=None, step_idx=None, total_steps=50, model_type="SD"):
        super().__init__()
        self.total_steps = total_steps
        self.total_layers = self.MODEL_TYPE.get(model_type, 16)
        self.start_step = start_step
        self.start_layer = start_layer
        self.layer_idx = layer_idx if layer_idx is not None else list(range(start_layer, self.total_layers))
        self.step_idx = step_idx if step_idx is not None else list(range(start_step, total_steps))
        print("MasaCtrl at denoising steps: ", self.step_idx)
        print("MasaCtrl at U-Net layers: ", self.layer_idx)

    def attn_batch(self, q, k, v, sim, attn, is_cross, place_in_unet, num_heads, **kwargs):
        """all samples of q attend to ONE sample's k, v: [(b h), n, d] x [h, n, d] -> [b, n, h*d]"""
        bh, n, d = q.shape
        b = bh // num_heads
        qh = q.reshape(b, num_heads, n, d).permute(1, 0, 2, 3).reshape(num_heads, b * n, d)
        s = torch.bmm(qh, k.transpose(1, 2)) * kwargs.get("scale")
        out = torch.bmm(s.softmax(-1), v)                           # h (b n) d
        return out.reshape(num_heads, b, n, d).permute(1, 2, 0, 3).reshape(b, n, num_heads * d)

    def forward(self, q, k, v, sim, attn, is_cross, place_in_unet, num_heads, **kwargs):
        if is_cross or self.cur_step not in self.step_idx or self.cur_att_layer // 2 not in self.layer_idx:
            return super().forward(q, k, v, sim, attn, is_cross, place_in_unet, num_heads, **kwargs)
        qu, qc = q.chunk(2)
        ku, kc = k.chunk(2)
        vu, vc = v.chunk(2)
        out_u = self.attn_batch(qu, ku[:num_heads], vu[:num_heads], None, None, is_cross, place_in_unet, num_heads, **kwargs)
        out_c = self.attn_batch(qc, kc[:num_heads], vc[:num_heads], None, None, is_cross, place_in_unet, num_heads, **kwargs)
        return torch.cat([out_u, out_c], dim=0)
