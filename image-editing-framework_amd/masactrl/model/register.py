"""`regiter_attention_editor_diffusers(model, editor)` (sic) / `unregister_attention_control(model, editor)` —
the hook API of `/root/reference/masactrl/model/register.py:6-89`.

The reference patches every `Attention.forward` with a closure that materialises `sim` and `attn`
([B*heads, N, N]: 2 GiB each in fp32 at 64x64) and hands them to the editor (:35-48).  Here the two editor classes
the reference CLIs use — `AttentionBase` (plain attention) and `MutualSelfAttentionControl` — are lowered to a
device plan: the fused flash-attention kernel takes per-batch K/V source rows, which IS mutual self-attention
(`ief_attn_flash_f16`, k_src / v_src).  Other editors (the store / mask variants, never instantiated by the
reference's scripts) need the materialised `sim` tensor and are rejected loudly.
"""
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


def regiter_attention_editor_diffusers(model, editor):
    unet = model.unet
    plan = lower_editor(editor, unet.device)
    if plan is None:
        raise NotImplementedError(
            f"{type(editor).__name__}: only AttentionBase and MutualSelfAttentionControl are lowered to the fused "
            "attention kernel; editors that read the materialised `sim`/`attn` tensors are not built (DESIGN.md §7)")
    mods = _attention_modules(unet)
    for m in mods:
        m.__dict__.pop("forward", None)
        m._plan = plan
    unet._plan = plan
    editor.num_att_layers = len(mods)
    return editor


def unregister_attention_control(model, editor):
    unet = model.unet
    for m in _attention_modules(unet):
        m.__dict__.pop("forward", None)
        m._plan = None
    unet._plan = None
    if editor is not None:
        editor.num_att_layers = 0
