"""`register_attention_control(model, controller)` / `unregister_attention_control(model, controller)`.

Same entry points and side effects as `/root/reference/p2p/model/register.py:3-117`:
  - every module named 'Attention' under the UNet children whose names contain "down" / "up" /
    "mid" is attached to the controller, counted, and `controller.num_att_layers` is set (:78-98);
  - `controller is None` installs a pass-through controller (:67-76);
  - unregister restores the modules and zeroes `num_att_layers` (:100-117).

Two ways a controller is attached:

  lowered (fused)   for the controller classes whose arithmetic is known — `EmptyControl`,
      `AttentionReplace`, `AttentionRefine`, `AttentionReweight` from this package OR the reference's
      own classes of the same names (duck-typed on their attributes) — `lower_controller` turns
      the controller into a `control.ControlPlan` (device tables); `Attention.forward` stays the
      native fused kernel path and reads the plan.  No attention map is materialised.
  generic           anything else (user subclasses, `AttentionStore`, `LocalBlend` users):
      `Attention.forward` is replaced by a closure with the reference's dataflow
      (register.py:11-64): Q/K/V projections -> materialised maps -> `controller(maps, is_cross,
      place)` in Python (in place on the cond half) -> maps x V -> out-projection.

`fused=False` forces the generic path for a lowerable controller (used by the parity tests
that hold both paths to identical results).
"""
from typing import Optional

import torch

from ... import control
from ...control import ControlPlan, XL

_LOWERABLE = {"EmptyControl", "DummyController", "AttentionReplace", "AttentionRefine", "AttentionReweight"}


class DummyController:
    def __call__(self, *args):
        return args[0]

    def __init__(self):
        self.num_att_layers = 0
        self.cur_step = 0
        self.cur_att_layer = 0

    def between_steps(self):
        return None


def _attention_modules(unet):
    """(place, module) in the reference's registration order."""
    out = []
    for name, child in unet.named_children():
        place = "down" if "down" in name else "up" if "up" in name else "mid" if "mid" in name else None
        if place is None:
            continue
        out += [(place, m) for m in child.modules() if m.__class__.__name__ == "Attention"]
    return out


# --------------------------------------------------------------------------------------- lowering
def _edit_tables(c):
    """(M [slots,77,77] fp32, scale1 [slots,77], keep_tgt [slots,77]) such that
    replace_cross_attention(P_src, P_tgt) == (P_src @ M) * scale1 + P_tgt * keep_tgt."""
    name = type(c).__name__
    if name == "AttentionReplace":
        M = c.mapper.detach().float().cpu()
        slots = M.shape[0]
        return M, torch.ones(slots, 77), torch.zeros(slots, 77)
    if name == "AttentionRefine":
        mapper = c.mapper.detach().cpu().long()
        a = c.alphas.detach().float().cpu().reshape(mapper.shape[0], 77)
        slots = mapper.shape[0]
        M = torch.zeros(slots, 77, 77)
        cols = torch.arange(77)
        for s in range(slots):
            M[s, mapper[s] % 77, cols] = 1.0  # column n gathers source word mapper[n]; -1 wraps like torch indexing
        return M, a, 1.0 - a
    if name == "AttentionReweight":
        eq = c.equalizer.detach().float().cpu()
        prev = getattr(c, "prev_controller", None)
        if prev is None:
            slots = eq.shape[0]
            return torch.eye(77).repeat(slots, 1, 1), eq.clone(), torch.zeros(slots, 77)
        M, s1, keep = _edit_tables(prev)
        return M, s1 * eq, keep * eq
    raise TypeError(name)


def lower_controller(controller, device, rows: str = "all") -> Optional[ControlPlan]:
    """ControlPlan for a known controller class, else None (generic path).
    rows: which rows of the CFG batch this UNet runs — "all" ([uncond..., cond...], the reference's batch), "cond" (only
    the conditional rows: the half every controller acts on) or "uncond" (only the unconditional rows: nothing to edit,
    the plan just keeps the controller's counters moving).  "cond" / "uncond" are the two ranks of a 2-GPU CFG split."""
    if rows not in ("all", "cond", "uncond"):
        raise ValueError('rows must be "all", "cond" or "uncond"')
    name = type(controller).__name__
    if name not in _LOWERABLE:
        return None
    if getattr(controller, "LOW_RESOURCE", False):
        return None  # two half-batch forwards per step: keep the Python protocol
    if name in ("EmptyControl", "DummyController") or rows == "uncond":
        return ControlPlan(controller, "empty", device)
    if getattr(controller, "local_blend", None) is not None:
        return None  # LocalBlend needs stored maps
    try:
        M, s1, keep = _edit_tables(controller)
    except (TypeError, AttributeError):
        return None
    alpha = controller.cross_replace_alpha.detach().float().cpu()  # [steps+1, slots, 1, 1, 77]
    steps1, slots = alpha.shape[0], alpha.shape[1]
    alpha = alpha.reshape(steps1, slots, 77)
    if M.shape[0] != slots or controller.batch_size != slots + 1:
        return None
    # P' = alpha * ((P_src M) s1 + P_tgt keep) + (1 - alpha) P_tgt
    c1 = alpha * s1[None]
    c2 = alpha * keep[None] + (1.0 - alpha)
    coef = torch.zeros(steps1, slots, 2, XL)
    coef[:, :, 0, :77], coef[:, :, 1, :77] = c1, c2
    mt = torch.zeros(slots, XL, XL)
    mt[:, :77, :77] = M.transpose(1, 2)
    lo, hi = controller.num_self_replace
    return ControlPlan(controller, "p2p", device, num_prompts=slots + 1, num_steps=steps1 - 1, mt=mt, coef_table=coef,
                       self_window=(int(lo), int(hi)), cond_only=(rows == "cond"))


# --------------------------------------------------------------------------------------- generic hook
def _generic_forward(attn, controller, place_in_unet):
    """Closure with the dataflow of `/root/reference/p2p/model/register.py:11-64` on our kernels."""
    to_out = attn.to_out[0] if isinstance(attn.to_out, torch.nn.ModuleList) else attn.to_out

    def forward(hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        is_cross = encoder_hidden_states is not None
        query = attn.to_q(hidden_states)
        context = encoder_hidden_states if is_cross else hidden_states
        key, value = attn.to_k(context), attn.to_v(context)
        probs = attn.get_attention_scores(attn.head_to_batch_dim(query), attn.head_to_batch_dim(key), None)
        probs = controller(probs, is_cross, place_in_unet)
        out = attn.apply_probs(probs, attn.head_to_batch_dim(value))
        return to_out(out)

    return forward


def register_attention_control(model, controller, fused: Optional[bool] = None, rows: str = "all"):
    """`rows` (not a reference argument): see `lower_controller`; anything but "all" needs a lowerable controller"""
    if controller is None:
        controller = DummyController()
    unet = model.unet
    mods = _attention_modules(unet)
    plan = None
    if fused is not False and all(getattr(m, "is_native", None) is not None for _, m in mods):
        plan = lower_controller(controller, unet.device, rows)
    if rows != "all" and plan is None:
        raise ValueError(f"{type(controller).__name__} cannot be lowered: a CFG-split forward needs a device plan")
    if fused is True and plan is None:
        raise ValueError(f"{type(controller).__name__} cannot be lowered to the fused path")
    for place, m in mods:
        if plan is not None:
            m.__dict__.pop("forward", None)      # back to the native fused forward
            m._plan = plan
        else:
            m._plan = None
            if "forward" not in m.__dict__:
                m._original_forward = m.forward
            m.forward = _generic_forward(m, controller, place)
    unet._plan = plan
    controller.num_att_layers = len(mods)
    return controller


def unregister_attention_control(model, controller):
    unet = model.unet
    for _, m in _attention_modules(unet):
        m.__dict__.pop("forward", None)
        m.__dict__.pop("_original_forward", None)
        m._plan = None
    unet._plan = None
    if controller is not None:
        controller.num_att_layers = 0
