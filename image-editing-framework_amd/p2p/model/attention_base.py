"""Controller protocol of Prompt-to-Prompt editing.

Same classes, constructor arguments, attributes and call protocol as
`/root/reference/p2p/model/attention_base.py` (AttentionControl :8-46, EmptyControl :49-55,
AttentionStore :57-91, AttentionControlEdit :94-140):

    controller(attn, is_cross, place_in_unet) -> attn     once per Attention module per forward
    controller.step_callback(latents), .between_steps(), .reset()
    attributes cur_step, cur_att_layer, num_att_layers (written by register_attention_control)

`attn` is [B*heads, N, L] with batch order [uncond..., cond...]; only the cond half is edited,
IN PLACE, and the returned tensor aliases the argument (:22).

These Python bodies are what the GENERIC hook path executes (materialised maps, any user
subclass).  For the classes defined in this package `register_attention_control` lowers the
same arithmetic into the fused HIP attention kernels instead (see `register.py`,
`lower_controller`); both paths are held to the oracle and to each other by
`tests/test_gpu_unet.py::test_p2p_controlled_forward_fused_generic_oracle` (every lowerable class) and, at full SD1.5 size,
`tests/test_gpu_zz_fullsize.py`.
"""
import abc
from typing import Dict, Optional, Tuple, Union

import torch

from . import ptp_utils
from .ptp_utils import LocalBlend


class AttentionControl(abc.ABC):
    def __init__(self, LOW_RESOURCE):
        self.cur_step = 0
        self.num_att_layers = -1
        self.cur_att_layer = 0
        self.LOW_RESOURCE = LOW_RESOURCE

    @property
    def num_uncond_att_layers(self):
        return self.num_att_layers if self.LOW_RESOURCE else 0

    def __call__(self, attn, is_cross: bool, place_in_unet: str):
        if self.cur_att_layer >= self.num_uncond_att_layers:
            if self.LOW_RESOURCE:
                attn = self.forward(attn, is_cross, place_in_unet)
            else:
                half = attn.shape[0] // 2
                attn[half:] = self.forward(attn[half:], is_cross, place_in_unet)
        self._advance()
        return attn

    def _advance(self):
        """Layer / step bookkeeping of `__call__` (:23-27), shared with the fused path."""
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers + self.num_uncond_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1
            self.between_steps()

    @abc.abstractmethod
    def forward(self, attn, is_cross: bool, place_in_unet: str):
        raise NotImplementedError

    def step_callback(self, x_t):
        return x_t

    def between_steps(self):
        return None

    def reset(self):
        self.cur_step = 0
        self.cur_att_layer = 0


class EmptyControl(AttentionControl):
    def __init__(self, LOW_RESOURCE):
        super().__init__(LOW_RESOURCE)

    def forward(self, attn, is_cross: bool, place_in_unet: str):
        return attn


class AttentionStore(AttentionControl):
    """Accumulates every map with N <= 32*32 per (place, kind) key across steps."""

    def __init__(self, LOW_RESOURCE):
        super().__init__(LOW_RESOURCE)
        self.step_store = self.get_empty_store()
        self.attention_store = {}

    @staticmethod
    def get_empty_store():
        return {k: [] for k in ("down_cross", "mid_cross", "up_cross", "down_self", "mid_self", "up_self")}

    def forward(self, attn, is_cross: bool, place_in_unet: str):
        if attn.shape[1] <= 32 ** 2:
            self.step_store[f"{place_in_unet}_{'cross' if is_cross else 'self'}"].append(attn)
        return attn

    def between_steps(self):
        if len(self.attention_store) == 0:
            self.attention_store = self.step_store
        else:
            for key, maps in self.attention_store.items():
                for i in range(len(maps)):
                    maps[i] += self.step_store[key][i]
        self.step_store = self.get_empty_store()

    def get_average_attention(self):
        return {key: [m / self.cur_step for m in maps] for key, maps in self.attention_store.items()}

    def reset(self):
        super().reset()
        self.step_store = self.get_empty_store()
        self.attention_store = {}


class AttentionControlEdit(AttentionControl, abc.ABC):
    def __init__(self, prompts, tokenizer, num_steps: int,
                 cross_replace_steps: Union[float, Tuple[float, float], Dict[str, Tuple[float, float]]],
                 self_replace_steps: Union[float, Tuple[float, float]],
                 local_blend: Optional[LocalBlend],
                 device=torch.device("cuda:0"), LOW_RESOURCE=False):
        super().__init__(LOW_RESOURCE)
        self.batch_size = len(prompts)
        self.cross_replace_alpha = ptp_utils.get_time_words_attention_alpha(
            prompts, num_steps, cross_replace_steps, tokenizer).to(device)
        if type(self_replace_steps) is float:
            self_replace_steps = 0, self_replace_steps
        self.num_self_replace = int(num_steps * self_replace_steps[0]), int(num_steps * self_replace_steps[1])
        self.local_blend = local_blend

    def forward(self, attn, is_cross: bool, place_in_unet: str):
        lo, hi = self.num_self_replace
        if is_cross or (lo <= self.cur_step < hi):
            heads = attn.shape[0] // self.batch_size
            attn = attn.reshape(self.batch_size, heads, *attn.shape[1:])
            source, targets = attn[0], attn[1:]
            if is_cross:
                gate = self.cross_replace_alpha[self.cur_step].to(attn.dtype)
                edited = self.replace_cross_attention(source, targets)
                attn[1:] = edited * gate + (1 - gate) * targets
            else:
                attn[1:] = self.replace_self_attention(source, targets)
            attn = attn.reshape(self.batch_size * heads, *attn.shape[2:])
        return attn

    def step_callback(self, x_t):
        if self.local_blend is not None:
            x_t = self.local_blend(x_t, self.attention_store)
        return x_t

    def replace_self_attention(self, attn_base, att_replace):
        if att_replace.shape[2] <= 16 ** 2:
            return attn_base.unsqueeze(0).expand(att_replace.shape[0], *attn_base.shape)
        return att_replace

    @abc.abstractmethod
    def replace_cross_attention(self, attn_base, att_replace):
        raise NotImplementedError
