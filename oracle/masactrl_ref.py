"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement of MasaCtrl's mutual self-attention (`/root/reference/masactrl/model/attention_control.py:37-68`)
as a q/k/v hook for `oracle/unet_ref.attention`: in the controlled self-attention layers the keys and values of
every sample in a half of the batch are replaced by those of the half's first sample, which is what
`attn_batch(qu, ku[:num_heads], vu[:num_heads], ...)` computes.  The call counting follows
`/root/reference/masactrl/model/attention_base.py:14-22`.  `AttentionBase.forward` itself is pinned by fixture G7
(`tests/golden/masactrl.npz`, made by importing the reference module); `attention_control.py` cannot be imported
here (needs torchvision), so the mutual rule is pinned only against the einops formulas restated in the test.
"""
from dataclasses import dataclass, field
from typing import List


@dataclass
class MasaCtrlRef:
    step_idx: List[int]
    layer_idx: List[int]
    num_att_layers: int = -1
    cur_step: int = 0
    cur_att_layer: int = 0

    def __call__(self, q, k, v, is_cross, place, heads):
        active = (not is_cross) and self.cur_step in self.step_idx and (self.cur_att_layer // 2) in self.layer_idx
        if active:
            bh = q.shape[0]
            half = bh // 2
            k, v = k.clone(), v.clone()
            for lo in (0, half):
                k[lo:lo + half] = k[lo:lo + heads].repeat(half // heads, 1, 1)
                v[lo:lo + half] = v[lo:lo + heads].repeat(half // heads, 1, 1)
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1
        return q, k, v
