"""Plug-and-Play hooks — same functions as `/root/reference/pnp/model/register.py`:

    register_time(model, t)                                            (:5-19)
    register_attention_control_efficient(model, injection_schedule)    (:27-90)    + unregister_... (:91-98)
    register_conv_control_efficient(model, injection_schedule)         (:100-182)  + unregister_... (:184-186)
    ..._xl variants (:188-364): on the SDXL family the self-attention of EVERY transformer block of `up_blocks[1]` and the
    conv2 output of `up_blocks[1].resnets[0]`

The reference replaces `forward` of eight decoder self-attention modules and of `up_blocks[1].resnets[1]` by closures
that overwrite, in place, the rows of the uncond_tgt / cond_tgt batch blocks with the cond_src block's Q, K (attention)
or conv2 output (resnet) whenever the current timestep is in the schedule.  Here the two schedules are lowered to
ONE device plan (`control.ControlPlan(kind="pnp")`): per step a Q/K source-row vector for the fused attention kernel
(`ief_attn_flash_f16`'s q_src / k_src) and a feature source-row vector for a row gather in front of conv2
(`ief_gather_rows_f16`).  The schedule is `scheduler.timesteps[:n]` in both reference call sites
(`pnp/model/sd_utils.py:16-21`), i.e. "the first n steps"; that is what the tables encode, and any other schedule is
rejected loudly.  `register_time` keeps the eager (un-captured) path in step; the captured loop counts on the device.
"""
from ...control import ControlPlan, StepCounter

QK_BLOCKS = {1: [1, 2], 2: [0, 1, 2], 3: [0, 1, 2]}     # decoder blocks 4-11 (register.py:84)
QK_BLOCKS_XL = {1: [0, 1, 2]}                            # the `_xl` functions (:246): every transformer block of up_blocks[1]


def _is_xl(unet) -> bool:
    return bool(getattr(unet.cfg, "addition_embed", False))


def _conv_module(unet):
    """the resnet whose conv2 output is injected: `up_blocks[1].resnets[1]` (:180), on the SDXL family
    `up_blocks[1].resnets[0]` (`register_conv_control_efficient_xl`, :339)"""
    return unet.up_blocks[1].resnets[0 if _is_xl(unet) else 1]


def _attention_modules(unet):
    return [m for m in unet.attention_modules()]


def _prefix_len(model, injection_schedule) -> int:
    """`injection_schedule` must be the first n timesteps of the current schedule; returns n"""
    ts = [int(t) for t in model.scheduler.timesteps]
    sched = [int(t) for t in (injection_schedule if injection_schedule is not None else [])]
    if sched != ts[: len(sched)]:
        raise ValueError("Plug-and-Play injection schedules are prefixes of scheduler.timesteps in the reference "
                         "(pnp/model/sd_utils.py:16-21); an arbitrary timestep set is not built")
    return len(sched)


def _plan(model) -> ControlPlan:
    unet = model.unet
    plan = getattr(unet, "_plan", None)
    if plan is not None and plan.kind != "pnp":
        raise RuntimeError("another attention controller is registered on this UNet; unregister it first")
    if plan is not None and plan.captured:      # a captured (pooled) loop owns that plan's tables: start a new one
        _detach(model)
        plan = None
    if plan is None:
        mods = _attention_modules(unet)
        plan = ControlPlan(StepCounter(len(mods)), "pnp", unet.device, num_steps=len(model.scheduler.timesteps))
        for m in mods:
            m._plan = plan
        unet._plan = plan
    return plan


def register_attention_control_efficient(model, injection_schedule, _table=None):
    plan = _plan(model)
    unet = model.unet
    layers = set()
    xl = _table is QK_BLOCKS_XL
    for res, blocks in (_table or QK_BLOCKS).items():
        for block in blocks:
            if res < len(unet.up_blocks) and block < len(unet.up_blocks[res].attentions):
                tbs = unet.up_blocks[res].attentions[block].transformer_blocks
                for module in ([tb.attn1 for tb in tbs] if xl else [tbs[0].attn1]):      # `_xl`: every block (:247-251)
                    layers.add(id(module))
                    setattr(module, "injection_schedule", injection_schedule)
    plan.pnp_layers = layers
    plan.pnp_qk_steps = _prefix_len(model, injection_schedule)
    plan._pnp.clear()


def register_conv_control_efficient(model, injection_schedule):
    plan = _plan(model)
    conv_module = _conv_module(model.unet)
    setattr(conv_module, "injection_schedule", injection_schedule)
    conv_module._inject = plan
    plan.pnp_conv_steps = _prefix_len(model, injection_schedule)
    plan._pnp.clear()


def _detach(model):
    for m in _attention_modules(model.unet):
        m._plan = None
    model.unet._plan = None
    _conv_module(model.unet)._inject = None


def _drop_plan_if_idle(model):
    plan = getattr(model.unet, "_plan", None)
    if plan is not None and plan.kind == "pnp" and not plan.pnp_layers and _conv_module(model.unet)._inject is None:
        for m in _attention_modules(model.unet):
            m._plan = None
        model.unet._plan = None


def unregister_attention_control_efficient(model):
    plan = getattr(model.unet, "_plan", None)
    if plan is not None and plan.kind == "pnp" and plan.captured:
        return _detach(model)                  # the tables stay with the captured loop (denoise pool); nothing to undo
    if plan is not None and plan.kind == "pnp":
        plan.pnp_layers = set()
        plan.pnp_qk_steps = 0
        plan._pnp.clear()
    _drop_plan_if_idle(model)


def unregister_conv_control_efficient(model):
    plan = getattr(model.unet, "_plan", None)
    if plan is not None and plan.kind == "pnp" and plan.captured:
        return _detach(model)
    _conv_module(model.unet)._inject = None
    if plan is not None and plan.kind == "pnp":
        plan.pnp_conv_steps = 0
        plan._pnp.clear()
    _drop_plan_if_idle(model)


def register_time(model, t):
    """the reference stamps the timestep on the hooked modules before every UNet call (:5-19); the plan needs the STEP
    INDEX, which for the eager path is the position of t in the schedule"""
    plan = getattr(model.unet, "_plan", None)
    conv_module = _conv_module(model.unet)
    setattr(conv_module, "t", t)
    if plan is not None and plan.kind == "pnp" and not plan.captured:
        ts = [int(x) for x in model.scheduler.timesteps]
        tv = int(t)
        plan.controller.cur_step = ts.index(tv) if tv in ts else len(ts)
        plan.controller.cur_att_layer = 0


# ---- the `_xl` functions of the reference (:188-364): same hooks, the SDXL family's injection sites
def register_attention_control_efficient_xl(model, injection_schedule):
    if not _is_xl(model.unet):
        raise ValueError("register_attention_control_efficient_xl: not an SDXL-family UNet")
    return register_attention_control_efficient(model, injection_schedule, _table=QK_BLOCKS_XL)


def register_conv_control_efficient_xl(model, injection_schedule):
    if not _is_xl(model.unet):
        raise ValueError("register_conv_control_efficient_xl: not an SDXL-family UNet")
    return register_conv_control_efficient(model, injection_schedule)


unregister_attention_control_efficient_xl = unregister_attention_control_efficient
unregister_conv_control_efficient_xl = unregister_conv_control_efficient
register_time_xl = register_time
