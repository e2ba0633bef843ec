"""The three Prompt-to-Prompt edit controllers.

Constructor signatures and attributes follow `/root/reference/p2p/model/attention_control.py`
(AttentionReplace :8-16, AttentionRefine :19-31, AttentionReweight :34-46):

  replace : P_edit[b,h,p,n] = sum_w P_src[h,p,w] * mapper[b,w,n]           (dense 77x77)
  refine  : P_edit = P_src[..., mapper[b]] * alphas + P_tgt * (1 - alphas)  (int64 gather; -1 wraps)
  reweight: P_edit = (prev_controller edit | P_src) * equalizer[b, n]
"""
from typing import Optional

import torch

from . import seq_aligner
from .attention_base import AttentionControlEdit
from .ptp_utils import LocalBlend


class AttentionReplace(AttentionControlEdit):
    def __init__(self, prompts, tokenizer, num_steps: int, cross_replace_steps: float,
                 self_replace_steps: float, local_blend: Optional[LocalBlend] = None,
                 device=torch.device("cuda:0"), LOW_RESOURCE=False, dtype=torch.float32):
        super().__init__(prompts, tokenizer, num_steps, cross_replace_steps, self_replace_steps,
                         local_blend, device, LOW_RESOURCE)
        self.mapper = seq_aligner.get_replacement_mapper(prompts, tokenizer).to(device).to(dtype)

    def replace_cross_attention(self, attn_base, att_replace):
        return torch.einsum("hpw,bwn->bhpn", attn_base, self.mapper.to(attn_base.dtype))


class AttentionRefine(AttentionControlEdit):
    def __init__(self, prompts, tokenizer, num_steps: int, cross_replace_steps: float,
                 self_replace_steps: float, local_blend: Optional[LocalBlend] = None,
                 device=torch.device("cuda:0"), LOW_RESOURCE=False):
        super().__init__(prompts, tokenizer, num_steps, cross_replace_steps, self_replace_steps,
                         local_blend, device, LOW_RESOURCE)
        mapper, alphas = seq_aligner.get_refinement_mapper(prompts, tokenizer)
        self.mapper = mapper.to(device)
        alphas = alphas.to(device)
        self.alphas = alphas.reshape(alphas.shape[0], 1, 1, alphas.shape[1])

    def replace_cross_attention(self, attn_base, att_replace):
        a = self.alphas.to(attn_base.dtype)
        from_source = attn_base[:, :, self.mapper].permute(2, 0, 1, 3)
        return from_source * a + att_replace * (1 - a)


class AttentionReweight(AttentionControlEdit):
    def __init__(self, prompts, tokenizer, num_steps: int, cross_replace_steps: float,
                 self_replace_steps: float, equalizer, local_blend: Optional[LocalBlend] = None,
                 controller: Optional[AttentionControlEdit] = None,
                 device=torch.device("cuda:0"), LOW_RESOURCE=False, dtype=torch.float32):
        super().__init__(prompts, tokenizer, num_steps, cross_replace_steps, self_replace_steps,
                         local_blend, device, LOW_RESOURCE)
        self.equalizer = equalizer.to(device).to(dtype)
        self.prev_controller = controller

    def replace_cross_attention(self, attn_base, att_replace):
        if self.prev_controller is not None:
            attn_base = self.prev_controller.replace_cross_attention(attn_base, att_replace)
        return attn_base[None, :, :, :] * self.equalizer[:, None, None, :].to(attn_base.dtype)
