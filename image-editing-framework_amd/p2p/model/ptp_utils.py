"""Step x word gating tables and LocalBlend for Prompt-to-Prompt.

Public surface of `/root/reference/p2p/model/ptp_utils.py`:
  get_time_words_attention_alpha (:66-84), update_alpha_time_word (:54-64),
  get_word_inds (:34-52), LocalBlend (:6-32).
Host-side, run once per edit; checked against the reference's outputs in
`tests/golden/p2p_host.npz` (G2) and `p2p_ctrl.npz` (G5).
"""
from typing import Dict, List, Optional, Tuple, Union

import torch
import torch.nn.functional as nnf

from .seq_aligner import get_word_inds  # same function in both reference files (:34-52 / seq_aligner:131-149)


def update_alpha_time_word(alpha, bounds: Union[float, Tuple[float, float]], prompt_ind: int,
                           word_inds: Optional[torch.Tensor] = None):
    """alpha[step, prompt, word] <- 1 inside [bounds) (fractions of the table length), else 0."""
    if type(bounds) is float:
        bounds = 0, bounds
    n = alpha.shape[0]
    lo, hi = int(bounds[0] * n), int(bounds[1] * n)
    if word_inds is None:
        word_inds = torch.arange(alpha.shape[2])
    alpha[:lo, prompt_ind, word_inds] = 0
    alpha[lo:hi, prompt_ind, word_inds] = 1
    alpha[hi:, prompt_ind, word_inds] = 0
    return alpha


def get_time_words_attention_alpha(prompts, num_steps,
                                   cross_replace_steps: Union[float, Dict[str, Tuple[float, float]]],
                                   tokenizer, max_num_words: int = 77):
    """-> fp32 [num_steps + 1, len(prompts) - 1, 1, 1, max_num_words] of 0/1 gates."""
    if type(cross_replace_steps) is not dict:
        cross_replace_steps = {"default_": cross_replace_steps}
    if "default_" not in cross_replace_steps:
        cross_replace_steps["default_"] = (0.0, 1.0)
    n_edit = len(prompts) - 1
    table = torch.zeros(num_steps + 1, n_edit, max_num_words)
    for p in range(n_edit):
        table = update_alpha_time_word(table, cross_replace_steps["default_"], p)
    for word, bounds in cross_replace_steps.items():
        if word == "default_":
            continue
        for p in range(n_edit):
            inds = get_word_inds(prompts[p + 1], word, tokenizer)
            if len(inds) > 0:
                table = update_alpha_time_word(table, bounds, p, inds)
    return table.reshape(num_steps + 1, n_edit, 1, 1, max_num_words)


class LocalBlend:
    """Word-masked latent blending from the stored 16x16 cross-attention maps (:6-32).

    Not reachable from the reference CLIs (they pass `local_blend=None`);
    kept because `AttentionControlEdit.step_callback` accepts one.
    """

    def __init__(self, tokenizer, prompts: List[str], words, threshold: float = 0.3,
                 device=torch.device("cuda:0"), MAX_NUM_WORDS: int = 77):
        layers = torch.zeros(len(prompts), 1, 1, 1, 1, MAX_NUM_WORDS)
        for i, (prompt, ws) in enumerate(zip(prompts, words)):
            if type(ws) is str:
                ws = [ws]
            for w in ws:
                layers[i, :, :, :, :, get_word_inds(prompt, w, tokenizer)] = 1
        self.alpha_layers = layers.to(device)
        self.threshold = threshold
        self.MAX_NUM_WORDS = MAX_NUM_WORDS

    def __call__(self, x_t, attention_store):
        k = 1
        maps = attention_store["down_cross"][2:4] + attention_store["up_cross"][:3]
        maps = [m.reshape(self.alpha_layers.shape[0], -1, 1, 16, 16, self.MAX_NUM_WORDS) for m in maps]
        maps = torch.cat(maps, dim=1)
        maps = (maps * self.alpha_layers.to(maps.dtype)).sum(-1).mean(1)
        mask = nnf.max_pool2d(maps, (2 * k + 1, 2 * k + 1), (1, 1), padding=(k, k))
        mask = nnf.interpolate(mask, size=x_t.shape[2:])
        mask = mask / mask.max(2, keepdims=True)[0].max(3, keepdims=True)[0]
        mask = mask.gt(self.threshold)
        mask = (mask[:1] + mask[1:]).to(x_t.dtype)
        return x_t[:1] + mask * (x_t - x_t[:1])
