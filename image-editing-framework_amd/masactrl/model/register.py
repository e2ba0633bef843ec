"""`regiter_attention_editor_diffusers(model, editor)` (sic) / `unregister_attention_control(model, editor)` —
the hook API of `/root/reference/masactrl/model/register.py:6-89`.

The reference patches every `Attention.forward` with a closure that materialises `sim` and `attn`
([B*heads, N, N]: 2 GiB each in fp32 at 64x64) and hands them to the editor (:35-48).  Here the two editor classes
the reference CLIs use — `AttentionBase` (plain attention) and `MutualSelfAttentionControl` — are lowered to a
device plan: the fused flash-attention kernel takes per-batch K/V source rows, which IS mutual self-attention
(`ief_attn_flash_f16`, k_src / v_src).  Any OTHER editor (a user subclass of `AttentionBase`, the store / mask
variants) takes the GENERIC path: a closure with the reference's dataflow (:10-48) on our kernels materialises `sim` and
`attn` ([B*heads, N, L]) and calls the editor's Python, exactly as `p2p/model/register.py` does for controllers.
"""
import torch

from ... import hip
from ...control import ControlPlan


def _attention_modules(unet):
    out = []
    for name, child in unet.named_children():
        if "down" in name or "mid" in name or "up" in name:
            out += [m for m in child.modules() if m.__class__.__name__ == "Attention"]
    return out


def lower_editor(editor, device):
    name = type(editor).__name__
    if name == "AttentionBase":
        return ControlPlan(editor, "empty", device)
    if name == "MutualSelfAttentionControl":
        return ControlPlan(editor, "masactrl", device, masa_steps=editor.step_idx, masa_layers=editor.layer_idx)
    return None


def _generic_forward(attn, editor, place_in_unet):
    """`ca_forward` of `/root/reference/masactrl/model/register.py:10-48` on our kernels: q, k, v as [(B*heads), N, d],
    `sim` = scaled scores, `attn` = their row softmax, all materialised; the editor returns [B, N, heads*d]"""
    to_out = attn.to_out[0] if isinstance(attn.to_out, torch.nn.ModuleList) else attn.to_out

    def forward(x, encoder_hidden_states=None, attention_mask=None, context=None, mask=None, **unused):
        if encoder_hidden_states is not None:
            context = encoder_hidden_states
        if attention_mask is not None or mask is not None:
            raise NotImplementedError("attention masks are not on the reference path (always None)")
        is_cross = context is not None
        context = context if is_cross else x
        q, k, v = attn.to_q(x), attn.to_k(context), attn.to_v(context)
        sim, probs = hip.attn_scores(q.contiguous(), k.contiguous(), attn.heads, attn.scale)
        out = editor(attn.head_to_batch_dim(q), attn.head_to_batch_dim(k), attn.head_to_batch_dim(v), sim, probs, is_cross,
                     place_in_unet, attn.heads, scale=attn.scale)
        return to_out(out.to(x.dtype).contiguous())

    return forward


def _places(unet):
    """(place, module) in the reference's registration order (:64-72: "down" / "mid" / "up" by child name)"""
    out = []
    for name, child in unet.named_children():
        for key in ("down", "mid", "up"):
            if key in name:
                out += [(key, m) for m in child.modules() if m.__class__.__name__ == "Attention"]
                break
    return out


def regiter_attention_editor_diffusers(model, editor):
    unet = model.unet
    plan = lower_editor(editor, unet.device)
    mods = _places(unet)
    for place, m in mods:
        m.__dict__.pop("forward", None)
        m._plan = plan
        if plan is None:                       # generic path: the editor's own Python on materialised tensors
            m._original_forward = m.forward
            m.forward = _generic_forward(m, editor, place)
    unet._plan = plan
    editor.num_att_layers = len(mods)
    return editor


def unregister_attention_control(model, editor):
    unet = model.unet
    for m in _attention_modules(unet):
        m.__dict__.pop("forward", None)
        m.__dict__.pop("_original_forward", None)
        m._plan = None
    unet._plan = None
    if editor is not None:
        editor.num_att_layers = 0
