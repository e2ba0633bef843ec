"""One small invocation of the hot path on cuda:0, checked against the CPU oracle.

A 3-step Prompt-to-Prompt edit (refine controller lowered to the device plan, hipGraph loop,
2 prompts x CFG = UNet batch 4) on the TINY shape family, compared with `oracle/p2p_ref.edit_loop`, in the
mode the benchmark reports ("f16x3": fp32 storage, split fp16 operands; bound 1e-4) and in the fp16-storage mode
(bound 3e-2).  The oracle is used as the CHECKER only.
"""
import torch


def run(verbose: bool = True):
    errs = {p: _run(p, bound, verbose) for p, bound in (("f16x3", 1e-4), ("f16", 3e-2))}
    return errs["f16x3"]


def _run(precision: str, bound: float, verbose: bool):
    from . import hip
    from .pipeline import StableDiffusionPipeline
    from .denoise import FusedDenoiser
    from .p2p.model.attention_control import AttentionRefine
    from .p2p.model.register import register_attention_control, unregister_attention_control
    from .p2p.model.sd_utils import _encode_prompts
    from oracle import p2p_ref

    hip.load()
    dev = torch.device("cuda:0")
    pipe = StableDiffusionPipeline.from_pretrained("synthetic:tiny", device=dev, keep_state_dict=True, precision=precision)
    cfg = pipe.cfg
    prompts = ["a photo of a house on a mountain", "a photo of a house on a mountain at fall"]
    steps = 3
    pipe.scheduler.set_timesteps(50)
    x_T = torch.randn(1, 4, cfg.sample_size, cfg.sample_size, generator=torch.Generator().manual_seed(8888))
    with torch.no_grad():
        u, c = _encode_prompts(pipe, prompts)
    ctx = torch.cat([u, c])
    ctrl = AttentionRefine(prompts, pipe.tokenizer, 50, 0.8, 0.4, device=dev)
    register_attention_control(pipe, ctrl)
    loop = FusedDenoiser(pipe, ctx, 2, (cfg.sample_size, cfg.sample_size), 7.5)
    try:
        lat = loop.run(x_T.to(dev), num_steps=steps).cpu()
    finally:
        loop.release()
        unregister_attention_control(pipe, ctrl)
    assert ctrl.cur_step == steps
    ref_ctrl = p2p_ref.P2PControlRef(mode="refine", num_prompts=2, cross_alpha=ctrl.cross_replace_alpha.float().cpu(),
                                     num_self_replace=ctrl.num_self_replace, mapper=ctrl.mapper.cpu(),
                                     alphas=ctrl.alphas.float().cpu())
    ref = p2p_ref.edit_loop(pipe._state_dict, cfg, ctx.float().cpu(), x_T, ref_ctrl, p2p_ref.DDIMRef(50), 7.5, num_steps=steps)
    err = ((lat - ref).abs().max() / ref.abs().max()).item()
    if verbose:
        print(f"smoke: 3-step P2P edit on {torch.cuda.get_device_name(0)}, precision {precision}: rel err vs fp32 oracle {err:.3e}")
    assert torch.isfinite(lat).all() and err < bound, (precision, err)
    return err
