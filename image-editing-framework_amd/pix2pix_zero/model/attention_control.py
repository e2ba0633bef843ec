"""`prep_unet` / `restore_original_processors` — `/root/reference/pix2pix-zero/model/attention_control.py:75-96`.

The reference swaps every Attention module's processor for `MyAttnProcessor`, whose only difference from the stock one is
`attn.attn_probs = attention_probs` (:44-47), and flips `requires_grad` on the `attn2` parameters (no optimiser ever
reads those gradients: only x_in is stepped, `sd_utils.py:160`).  Here the cross-attention maps are written by
`ief_attn_probs_f16` into a buffer the sampler hands each module (`Attention.map_out`) and the gradient is the hand-written
reverse pass (`grad.UNetAdjoint(mode="input")`), so preparing the UNet means: no foreign processor or hook may be
installed, and the native path must be the one that runs.
"""


def prep_unet(unet):
    original_processors = {}
    for name, module in unet.named_modules():
        if type(module).__name__ == "Attention":
            original_processors[name] = module.get_processor()
    if unet._plan is not None or not all(m.is_native() for m in unet.attention_modules()):
        raise RuntimeError("pix2pix-zero: an attention controller / processor is installed on this UNet; the reference "
                           "runs P2P_Zero on an otherwise unmodified pipeline")
    return unet, original_processors


def restore_original_processors(unet, original_processors):
    for name, module in unet.named_modules():
        if type(module).__name__ == "Attention" and name in original_processors:
            module.set_processor(original_processors[name])
