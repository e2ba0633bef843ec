"""SDXL shape family (BASELINE.json config 5) on a real MI355X against the fp32 CPU oracle: three levels with no attention
at the first, several BasicTransformerBlocks per Transformer2DModel (LayerNorm folding chained block to block), head dim
64, linear projections, the text-time additional embedding.  Tolerances as in test_gpu_unet.py: one UNet forward
<= 2e-2 of max |reference| (measured ~2e-3)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from ief_amd import config, weights  # noqa: E402
from ief_amd.pipeline import StableDiffusionPipeline  # noqa: E402
from oracle import unet_ref  # noqa: E402

DEV = torch.device("cuda:0")


def rel_err(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert torch.isfinite(got).all()
    return ((got - ref).abs().max() / ref.abs().max()).item()


@pytest.fixture(scope="module")
def smallxl():
    return StableDiffusionPipeline.from_pretrained("synthetic:smallxl", keep_state_dict=True)


def _inputs(cfg, B, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4, cfg.sample_size, cfg.sample_size, generator=g)
    ctx = torch.randn(B, 77, cfg.cross_attention_dim, generator=g) * 0.1
    size = float(cfg.sample_size * 8)
    added = {"text_embeds": torch.randn(B, cfg.pooled_text_dim, generator=g) * 0.5,
             "time_ids": torch.tensor([[size, size, 0.0, 0.0, size, size]] * B)}
    return x, ctx, added


@pytest.mark.parametrize("B", [1, 2, 4])
def test_smallxl_forward_vs_oracle(smallxl, B):
    cfg = smallxl.cfg
    x, ctx, added = _inputs(cfg, B)
    ref = unet_ref.unet_forward(smallxl._state_dict, cfg, x, 501, ctx, added_cond_kwargs=added)
    got = smallxl.unet(x.to(DEV), 501, encoder_hidden_states=ctx.to(DEV),
                       added_cond_kwargs={k: v.to(DEV) for k, v in added.items()})["sample"]
    e = rel_err(got, ref)
    # the additional embedding must matter, or the test would not see a wrong one
    other = dict(added, text_embeds=added["text_embeds"].flip(1))
    moved = rel_err(unet_ref.unet_forward(smallxl._state_dict, cfg, x, 501, ctx, added_cond_kwargs=other), ref)
    print(f"smallxl B={B}: forward {e:.2e}; another pooled embedding moves the output by {moved:.2e}")
    assert e < 2e-2 and moved > 5 * e
    with pytest.raises(ValueError):
        smallxl.unet(x.to(DEV), 501, encoder_hidden_states=ctx.to(DEV))


def test_smallxl_folded_layernorm_chain_equals_unfolded(smallxl, monkeypatch):
    """depth > 1: block k + 1 folds its first LayerNorm on the row moments block k's last GEMM emitted"""
    from ief_amd import unet as unet_mod
    cfg = smallxl.cfg
    x, ctx, added = _inputs(cfg, 2, seed=1)
    kw = dict(encoder_hidden_states=ctx.to(DEV), added_cond_kwargs={k: v.to(DEV) for k, v in added.items()})
    folded = smallxl.unet(x.to(DEV), 301, **kw)["sample"]
    monkeypatch.setattr(unet_mod, "FOLD_LN", False)
    plain = smallxl.unet(x.to(DEV), 301, **kw)["sample"]
    ref = unet_ref.unet_forward(smallxl._state_dict, cfg, x, 301, ctx, added_cond_kwargs=added)
    e1, e2 = rel_err(folded, ref), rel_err(plain, ref)
    print(f"smallxl folded {e1:.2e}, LayerNorm launches {e2:.2e}")
    assert e1 < 2e-2 and e2 < 2e-2 and not torch.equal(folded, plain)


def test_module_counts():
    from oracle.unet_ref import count_attention_layers
    assert weights.num_params(config.SDXL) == 2_567_463_684          # the published size of the SDXL base UNet
    assert count_attention_layers(config.SDXL) == 140                # 70 BasicTransformerBlocks


# ------------------------------------------------------------------------------------------------ P2P on the XL family
import os  # noqa: E402
import subprocess  # noqa: E402
import sys  # noqa: E402

import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402

from _procs import call_main, run_mixed  # noqa: E402

from ief_amd.pipeline import StableDiffusionXLPipeline  # noqa: E402
from ief_amd.p2p.model.sd_utils import P2P_XL  # noqa: E402
from ief_amd.p2p.model.attention_base import EmptyControl  # noqa: E402
from ief_amd.p2p.model.attention_control import AttentionRefine  # noqa: E402
from oracle import p2p_ref  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROMPTS = ["a green apple on a wooden table", "a red apple on a wooden table"]


@pytest.fixture(scope="module")
def xlpipe():
    return StableDiffusionXLPipeline.from_pretrained("synthetic:smallxl", keep_state_dict=True)


def test_p2p_xl_loop_vs_oracle(xlpipe):
    """the captured-graph sampler with the additional embedding folded into its per-step time rows, against the oracle's
    eager loop (Python controller on materialised maps) — plain CFG sampling and an AttentionRefine edit"""
    cfg = xlpipe.cfg
    steps = 4
    size = cfg.sample_size * 8
    editor = P2P_XL(xlpipe, steps)
    x_T = torch.randn(1, 4, cfg.sample_size, cfg.sample_size, generator=torch.Generator().manual_seed(3))
    unc, cond, added = editor._encode(xlpipe, PROMPTS, size, size)
    assert unc.abs().max() == 0 and added["time_ids"].shape == (4, 6) and added["text_embeds"].shape == (4, cfg.pooled_text_dim)
    ctx = torch.cat([unc, cond]).float().cpu()
    added_cpu = {k: v.float().cpu() for k, v in added.items()}
    sched = p2p_ref.DDIMRef(num_inference_steps=steps)
    sd = xlpipe._state_dict
    with torch.no_grad():
        plain_ref = p2p_ref.edit_loop(sd, cfg, ctx, x_T, None, sched, 7.5, added_cond_kwargs=added_cpu)
    got, _ = editor.text2image_ldm_stable(xlpipe, PROMPTS, EmptyControl(LOW_RESOURCE=False), num_inference_steps=steps, guidance_scale=7.5,
                                          latent=x_T, return_latents=True)
    e0 = rel_err(got, plain_ref)
    ctrl = AttentionRefine(prompts=PROMPTS, tokenizer=xlpipe.tokenizer, num_steps=steps, cross_replace_steps=0.8,
                           self_replace_steps=0.4, device=DEV)
    edit, _ = editor.text2image_ldm_stable(xlpipe, PROMPTS, ctrl, num_inference_steps=steps, guidance_scale=7.5,
                                           latent=x_T, return_latents=True)
    ref_ctrl = p2p_ref.P2PControlRef(mode="refine", num_prompts=2, cross_alpha=ctrl.cross_replace_alpha.float().cpu(),
                                     num_self_replace=ctrl.num_self_replace, mapper=ctrl.mapper.cpu(),
                                     alphas=ctrl.alphas.float().cpu())
    with torch.no_grad():
        edit_ref = p2p_ref.edit_loop(sd, cfg, ctx, x_T, ref_ctrl, sched, 7.5, added_cond_kwargs=added_cpu)
    e1, moved = rel_err(edit, edit_ref), rel_err(plain_ref, edit_ref)
    print(f"P2P_XL {steps} steps: plain {e0:.2e}, AttentionRefine edit {e1:.2e}; the edit moves the latents by {moved:.2e}")
    from ief_amd.p2p.model.register import unregister_attention_control
    unregister_attention_control(xlpipe, None)       # the sampler leaves its controller registered, as the reference does
    assert e0 < 5e-2 and e1 < 5e-2


def test_p2p_xl_cli(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "image-editing-framework_amd", "p2p", "edit_syn.py"),
                        "--sd_version", "smallxl"], cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    src = np.array(Image.open(tmp_path / "exp" / "source.png")).astype(int)
    edit = np.array(Image.open(tmp_path / "exp" / "edit.png")).astype(int)
    assert src.shape == edit.shape == (128, 128, 3) and np.abs(src - edit).max() > 0


# ------------------------------------------------------------------------------------------------ Pix2Pix-zero on the XL family
from ief_amd.grad import UNetAdjoint  # noqa: E402
from ief_amd.p2p.inversion.ddim import ddim_inversion_xl  # noqa: E402
from ief_amd.pix2pix_zero.model.sd_utils import P2P_Zero_XL  # noqa: E402
from oracle import p2pzero_ref  # noqa: E402


def _exec_order(modules):
    rank = lambda m: (0 if m.layer_name.startswith("down") else 1 if m.layer_name.startswith("mid") else 2)
    return sorted(range(len(modules)), key=lambda i: (rank(modules[i]), i))


def test_unet_input_gradient_smallxl(smallxl):
    """the reverse pass through transformers several blocks deep, with per-batch-row time-embedding rows"""
    cfg, unet = smallxl.cfg, smallxl.unet
    B = 2
    x, ctx, added = _inputs(cfg, B, seed=4)
    ctx = ctx.half().float()
    added_dev = {k: v.to(DEV) for k, v in added.items()}
    unet(x.to(DEV), 401, encoder_hidden_states=ctx.to(DEV), added_cond_kwargs=added_dev)
    cross = [m for m in unet.attention_modules() if m.is_cross]
    g = torch.Generator().manual_seed(5)
    refs = [torch.softmax(torch.randn(B * m.heads, m.last_tokens, 77, generator=g) * 1.5, -1).half() for m in cross]
    loss_ref, grad_ref = p2pzero_ref.input_gradient(smallxl._state_dict, cfg, x, 401, ctx,
                                                    [refs[i].float() for i in _exec_order(cross)], added)
    gs = 1024.0
    adj = UNetAdjoint(unet, gs, mode="input")
    adj.set_reference_maps([r.to(DEV) for r in refs])
    temb = unet.time_rows(torch.tensor([401.0], device=DEV), unet.aug_embedding(added_dev))
    assert temb.shape[0] == B
    adj.forward(x.to(DEV), temb, ctx.half().to(DEV))
    d_x = adj.backward(torch.zeros_like(x, device=DEV)) / gs
    e, el = rel_err(d_x, grad_ref), abs(adj.loss_parts.sum().item() - loss_ref) / loss_ref
    print(f"smallxl B={B}: d objective / d latent {e:.2e} (max |grad| {grad_ref.abs().max():.3e}); objective {el:.2e}; "
          f"{len(cross)} cross-attention modules")
    assert e < 5e-2 and el < 1e-2


def test_context_gradient_smallxl(smallxl):
    """mode "context" (null-text inversion's gradient) through the deeper transformers: the chain must stop at the first
    block of the first transformer and still collect every module's dK / dV"""
    cfg, unet = smallxl.cfg, smallxl.unet
    x, ctx, added = _inputs(cfg, 1, seed=6)
    ctx16 = ctx.half()
    added_dev = {k: v.to(DEV) for k, v in added.items()}
    g = torch.Generator().manual_seed(7)
    d_eps = torch.randn(1, 4, cfg.sample_size, cfg.sample_size, generator=g)
    cf = ctx16.float().requires_grad_(True)
    ref = unet_ref.unet_forward(smallxl._state_dict, cfg, x, 401, cf, added_cond_kwargs=added)
    (ref * d_eps).sum().backward()
    adj = UNetAdjoint(unet, 1.0)
    temb = unet.time_rows(torch.tensor([401.0], device=DEV), unet.aug_embedding(added_dev))
    eps = adj.forward(x.to(DEV), temb, ctx16.to(DEV))
    got = adj.backward(d_eps.to(DEV))
    e_f, e_g = rel_err(eps, ref.detach()), rel_err(got, cf.grad)
    print(f"smallxl: forward(tape) {e_f:.2e}; d/d ctx {e_g:.2e}")
    assert e_f < 2e-2 and e_g < 5e-2


def test_ddim_inversion_xl_vs_oracle(xlpipe):
    cfg = xlpipe.cfg
    steps = 4
    xlpipe.scheduler.set_timesteps(steps)
    x0 = torch.randn(1, 4, cfg.sample_size, cfg.sample_size, generator=torch.Generator().manual_seed(8))
    inv = ddim_inversion_xl()
    lats, context = inv.ddim_inversion_loop(xlpipe, x0.to(DEV), PROMPTS[:1])
    emb, _, pooled, _ = context
    size = float(cfg.sample_size * 8)
    added = {"text_embeds": pooled.float().cpu(), "time_ids": torch.tensor([[size, size, 0.0, 0.0, size, size]])}
    with torch.no_grad():
        ref = p2p_ref.ddim_inversion_loop(xlpipe._state_dict, cfg, emb.float().cpu(), x0, p2p_ref.DDIMRef(steps),
                                          added_cond_kwargs=added)
    e = max(rel_err(a, b) for a, b in zip(lats, ref))
    print(f"ddim_inversion_xl {steps} steps: worst latent {e:.2e}")
    assert len(lats) == steps + 1 and e < 2e-2


def test_p2pzero_xl_two_pass_vs_oracle(xlpipe):
    cfg = xlpipe.cfg
    steps, run_steps, gscale, amount = 10, 3, 7.5, 0.1
    size = cfg.sample_size * 8
    sched = p2p_ref.DDIMRef(num_inference_steps=steps)
    x_T = torch.randn(1, 4, cfg.sample_size, cfg.sample_size, generator=torch.Generator().manual_seed(9))
    editor = P2P_Zero_XL(xlpipe, steps)
    (e_s, a_s), (e_t, a_t) = (editor._encode(xlpipe, p, size, size) for p in PROMPTS)
    cpu = lambda d: {k: v.float().cpu() for k, v in d.items()}
    rec_ref, edit_ref, losses_ref = p2pzero_ref.p2pzero(xlpipe._state_dict, cfg, e_s.float().cpu(), e_t.float().cpu(), x_T, sched,
                                                        gscale, amount, num_steps=run_steps, added_src=cpu(a_s),
                                                        added_tgt=cpu(a_t))
    rec, edit = editor(prompt=PROMPTS, num_inference_steps=steps, guidance_scale=gscale, guidance_amount=amount, latents=x_T,
                       return_latents=True, num_steps=run_steps)
    e_rec, e_edit = rel_err(rec, rec_ref), rel_err(edit, edit_ref)
    e_loss = max(abs(a - b) / b for a, b in zip(editor.last_losses, losses_ref))
    print(f"P2P_Zero_XL {run_steps} steps: reconstruction {e_rec:.2e}, edit {e_edit:.2e}, objective {e_loss:.2e} ({losses_ref})")
    assert e_rec < 5e-2 and e_edit < 5e-2 and e_loss < 2e-2


def test_p2pzero_xl_clis(tmp_path):
    folder = os.path.join(ROOT, "image-editing-framework_amd", "pix2pix_zero")
    rng = np.random.RandomState(0)
    img = np.kron(rng.randint(0, 255, (8, 8, 3)), np.ones((16, 16, 1))).astype(np.uint8)
    Image.fromarray(img).save(tmp_path / "test.jpg")
    run_mixed([([os.path.join(folder, "edit_syn.py"), "--sd_version", "smallxl"], tmp_path / "syn"),
                  ([os.path.join(folder, "edit_real.py"), "--sd_version", "smallxl", "--inversion_type", "ddim", "--source_image",
                    str(tmp_path / "test.jpg")], tmp_path / "real")], in_process=(0,))
    for name in ("source.png", "edit.png"):
        assert (tmp_path / "syn" / "exp" / name).exists()
    for name in ("source.png", "inversion.png", "edit.png"):
        assert (tmp_path / "real" / "exp" / name).exists()


# ------------------------------------------------------------------------------------------------ MasaCtrl on the XL family
def test_masactrl_xl_sampler_vs_oracle(xlpipe):
    from ief_amd.masactrl.model.attention_control import MutualSelfAttentionControl
    from ief_amd.masactrl.model.register import regiter_attention_editor_diffusers, unregister_attention_control as unreg
    from ief_amd.masactrl.model.sd_utils import MasaCtrl_XL
    from ief_amd.p2p.model.sd_utils import encode_prompt_xl
    from oracle.masactrl_ref import MasaCtrlRef
    cfg = xlpipe.cfg
    steps, run = 10, 6
    size = cfg.sample_size * 8
    nlayers = unet_ref.count_attention_layers(cfg)                   # 56 Attention modules = 28 self-attention layers
    layers = list(range(4, 28))
    editor = MasaCtrl_XL(xlpipe, steps)
    c = MutualSelfAttentionControl(1, 4, layer_idx=layers, total_steps=steps, model_type="SDXL")
    regiter_attention_editor_diffusers(xlpipe, c)
    x_T = torch.randn(1, 4, cfg.sample_size, cfg.sample_size, generator=torch.Generator().manual_seed(11))
    xlpipe.scheduler.set_timesteps(steps)
    from ief_amd.denoise import FusedDenoiser
    emb, added = encode_prompt_xl(xlpipe, PROMPTS, DEV, True, size, size, 2)
    loop = FusedDenoiser(xlpipe, emb, 2, (cfg.sample_size, cfg.sample_size), 7.5, added_cond_kwargs=added)
    try:
        lat = loop.run(x_T.to(DEV), num_steps=run).cpu()
    finally:
        loop.release()
    assert c.cur_step == run
    unreg(xlpipe, c)
    sched = p2p_ref.DDIMRef(steps)
    added_cpu = {k: v.float().cpu() for k, v in added.items()}

    def ref_loop(hook):
        lat_ref = x_T.expand(2, -1, -1, -1).clone()
        for t in sched.timesteps[:run]:
            with torch.no_grad():
                eps = unet_ref.unet_forward(xlpipe._state_dict, cfg, torch.cat([lat_ref] * 2), t, emb.float().cpu(), qkv_hook=hook,
                                            added_cond_kwargs=added_cpu)
            eu, ec = eps.chunk(2)
            lat_ref = sched.step(eu + 7.5 * (ec - eu), int(t), lat_ref)
        return lat_ref

    ref = ref_loop(MasaCtrlRef(step_idx=list(range(1, steps)), layer_idx=layers, num_att_layers=nlayers))
    plain = ref_loop(None)
    e, effect = rel_err(lat, ref), rel_err(plain, ref)
    print(f"MasaCtrl_XL {run}-step sampler: {e:.2e}; the mutual attention moves the latents by {effect:.2e}")
    assert e < 5e-2
    # one forward on DIFFERENT batch rows (in the sampler both rows start from one x_T and the synthetic text encoder
    # separates the prompts little): here the control must move the output by far more than the tolerance
    x, ctx, add4 = _inputs(cfg, 4, seed=12)
    x = torch.cat([x[:1], 0.6 * x[:1] + 0.8 * x[1:2]] * 2)
    c = MutualSelfAttentionControl(1, 4, layer_idx=layers, total_steps=steps, model_type="SDXL")
    regiter_attention_editor_diffusers(xlpipe, c)
    c.cur_step = 3
    got = xlpipe.unet(x.to(DEV), 501, encoder_hidden_states=ctx.to(DEV), added_cond_kwargs={k: v.to(DEV) for k, v in add4.items()})["sample"]
    unreg(xlpipe, c)
    r = MasaCtrlRef(step_idx=list(range(1, steps)), layer_idx=layers, num_att_layers=nlayers, cur_step=3)
    with torch.no_grad():
        ref1 = unet_ref.unet_forward(xlpipe._state_dict, cfg, x, 501, ctx, qkv_hook=r, added_cond_kwargs=add4)
        plain1 = unet_ref.unet_forward(xlpipe._state_dict, cfg, x, 501, ctx, added_cond_kwargs=add4)
    e1, effect1 = rel_err(got, ref1), rel_err(plain1, ref1)
    print(f"MasaCtrl_XL forward: fused-vs-oracle {e1:.2e}, size of the control {effect1:.2e}")
    assert e1 < 2e-2 and effect1 > 10 * e1
    # the class the CLI uses, end to end to uint8 images
    regiter_attention_editor_diffusers(xlpipe, MutualSelfAttentionControl(2, 20, layer_idx=layers, total_steps=steps,
                                                                           model_type="SDXL"))
    imgs, _ = editor(prompt=PROMPTS, latents=torch.cat([x_T, x_T]).to(DEV), guidance_scale=7.5, num_inference_steps=steps)
    unreg(xlpipe, None)
    assert imgs.shape == (2, size, size, 3) and imgs.dtype == np.uint8


# ------------------------------------------------------------------------------------------------ null-text inversion on the XL family
def test_nti_xl_loop_vs_oracle(xlpipe):
    """`NTI_XL`: lr 5e-2, restart from the negative embedding at every timestep, separate added_cond_kwargs for the two
    UNet calls.  epsilon = 0 disables the early stop.  One Adam step per timestep: with lr = 5e-2 on an embedding of
    magnitude 0.1 every further inner step amplifies the +-lr sign noise of the first (measured with 3 inner steps: 99 %
    of the elements agree at timestep 0, 86 % at timestep 1, while the GRADIENT of each first inner step, taken at the
    product's own state, agrees with autograd to <= 9e-3 of its maximum and 99.5-99.9 % of its signs)."""
    from ief_amd.nti import NullTextOptimizer
    from ief_amd.p2p.inversion.nti import NTI_XL
    cfg = xlpipe.cfg
    steps, inner, outer, gs = 4, 1, 3, 7.5
    xlpipe.scheduler.set_timesteps(steps)
    size = float(cfg.sample_size * 8)
    x0 = torch.randn(1, 4, cfg.sample_size, cfg.sample_size, generator=torch.Generator().manual_seed(13))
    inv = NTI_XL()
    lats, context = inv.ddim_inversion_loop(xlpipe, x0.to(DEV), PROMPTS[:1])
    emb, neg, pooled, neg_pooled = (c.float().cpu() for c in context)
    # SDXL's negative embedding is all zeros: its 77 token rows stay identical and most channels' gradients sit at the fp16
    # noise floor, where Adam's +-lr first steps take either sign — a degenerate comparison.  Both sides therefore start
    # from the same small random embedding here; the all-zeros start runs below through the class the CLIs use.
    neg = (torch.randn(neg.shape, generator=torch.Generator().manual_seed(14)) * 0.1).half().float()
    ids = torch.tensor([[size, size, 0.0, 0.0, size, size]])
    a_c, a_u = {"text_embeds": pooled, "time_ids": ids}, {"text_embeds": neg_pooled, "time_ids": ids}
    lat_cpu = [l.float().cpu() for l in lats]
    ref = p2p_ref.null_optimization(xlpipe._state_dict, cfg, lat_cpu, torch.cat([neg, emb]), p2p_ref.DDIMRef(steps),
                                    num_inner_steps=inner, epsilon=0.0, guidance_scale=gs, num_outer=outer, added_cond=a_c,
                                    added_uncond=a_u, lr=5e-2, restart=True, grad_trace=(trace := []))
    dev = lambda d: {k: v.to(DEV) for k, v in d.items()}
    opt = NullTextOptimizer(xlpipe, emb, gs, tuple(x0.shape[-2:]), added_cond=dev(a_c), added_uncond=dev(a_u), lr=5e-2,
                            restart=True)
    got = opt.run(lats, neg, inner, 0.0, num_outer=outer)
    opt.release()
    assert opt.inner_steps_run == [inner] * outer
    for i, (a, b) in enumerate(zip(got, ref)):
        a = a.cpu()
        moved = (b - neg).abs().max().item()
        diff = (a - b).abs()
        frac_close = (diff <= 0.1 * moved).float().mean().item()
        # one Adam step from the SAME start (restart): an element whose gradient is significant (>= 5 % of the largest) must
        # move the oracle's way by the oracle's amount; elements at the fp16 noise floor may take either sign of +-lr.  The
        # latents of timestep i come from the product's own earlier steps, so later timesteps keep only the bulk criterion.
        g = next(gr for ti, tj, gr in trace if (ti, tj) == (i, 0))
        strong = g.abs() >= 0.05 * g.abs().max()
        ok = (diff[strong] <= 0.1 * moved)
        print(f"NTI_XL step {i}: moved {moved:.3e}, max diff {diff.max().item():.3e}, within 10% of movement: {frac_close:.4f}; "
              f"significant elements {int(strong.sum())}, of which off {int((~ok).sum())}")
        if i == 0:
            assert strong.sum() > 50
        assert ok.float().mean().item() > 0.98
        assert frac_close > 0.93 and diff.max().item() <= 2.1 * moved
    # the class the CLIs use (early stop active)
    out = inv.null_optimization(xlpipe, lats, context, 2, 1e-5, gs)
    assert len(out) == steps and out[0].shape == (1, 77, cfg.cross_attention_dim)


def test_xl_null_text_clis(tmp_path):
    rng = np.random.RandomState(0)
    img = np.kron(rng.randint(0, 255, (8, 8, 3)), np.ones((16, 16, 1))).astype(np.uint8)
    Image.fromarray(img).save(tmp_path / "test.jpg")
    folders = ("p2p", "pix2pix_zero")
    run_mixed([([os.path.join(ROOT, "image-editing-framework_amd", folder, "edit_real.py"), "--sd_version", "smallxl",
                    "--inversion_type", "null-text", "--source_image", str(tmp_path / "test.jpg")], tmp_path / folder)
                  for folder in folders], in_process=(0,))
    for folder in folders:
        for name in ("source.png", "inversion.png", "edit.png"):
            assert (tmp_path / folder / "exp" / name).exists()


# ------------------------------------------------------------------------------------------------ local SDXL-layout directory
def test_local_sdxl_directory_with_clip_text_encoders(tmp_path):
    """`sd_maps["xl-base"] -> local dir` (README.md:30-32 of the reference): unet/ + vae/ safetensors in diffusers' layout and
    two CLIP text encoders loaded by `transformers` (random weights of a tiny CLIP configuration here: there are no
    checkpoints offline; the tokenizer falls back to the seeded one, a CLIP vocabulary cannot be made up).  The pipeline
    must produce the same UNet outputs as the synthetic one built from the same tensors, and run P2P_XL end to end."""
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    from test_cabi_and_host import _write_diffusers_dir
    cfg = config.SMALLXL
    sd = weights.synthetic_state_dict(cfg, 0)
    root = str(tmp_path / "sdxl_dir")
    _write_diffusers_dir(root, cfg, sd, xl=True)
    d2 = cfg.pooled_text_dim
    d1 = cfg.cross_attention_dim - d2
    torch.manual_seed(0)
    kw = dict(vocab_size=49408, num_hidden_layers=2, num_attention_heads=2, max_position_embeddings=77)
    CLIPTextModel(CLIPTextConfig(hidden_size=d1, intermediate_size=2 * d1, **kw)).save_pretrained(os.path.join(root, "text_encoder"))
    CLIPTextModelWithProjection(CLIPTextConfig(hidden_size=d2, intermediate_size=2 * d2, projection_dim=d2, **kw)).save_pretrained(
        os.path.join(root, "text_encoder_2"))
    pipe = StableDiffusionXLPipeline.from_pretrained(root)
    assert type(pipe.text_encoder).__name__ == "CLIPTextModel" and pipe.cfg.depth(2) == cfg.depth(2)
    emb, neg, pooled, neg_pooled = pipe.encode_prompt(PROMPTS)
    assert emb.shape == (2, 77, cfg.cross_attention_dim) and pooled.shape == (2, d2) and neg.abs().max() == 0
    # same tensors, rounded to fp16 on disk (biases, norm affines and the folded LayerNorm weights included, which the
    # synthetic pipeline packs from fp32): the same UNet up to that rounding
    x, ctx, added = _inputs(cfg, 2, seed=21)
    kwargs = dict(encoder_hidden_states=ctx.to(DEV), added_cond_kwargs={k: v.to(DEV) for k, v in added.items()})
    ref_pipe = StableDiffusionXLPipeline.from_pretrained("synthetic:smallxl")
    e = rel_err(pipe.unet(x.to(DEV), 301, **kwargs)["sample"], ref_pipe.unet(x.to(DEV), 301, **kwargs)["sample"])
    print(f"local directory vs synthetic pipeline of the same tensors: {e:.2e}")
    assert e < 1e-2
    editor = P2P_XL(pipe, 3)
    images, _ = editor.text2image_ldm_stable(pipe, PROMPTS, EmptyControl(LOW_RESOURCE=False), num_inference_steps=3)
    assert images.shape == (2, cfg.sample_size * 8, cfg.sample_size * 8, 3) and images.dtype == np.uint8


def test_xl_drivers_and_masactrl_clis(tmp_path):
    """the `StableDiffusionXLPipeline` dispatch of the remaining scripts: masactrl/edit_real.py (MasaCtrl_XL after
    ddim_inversion_xl) and the PIE drivers p2p/test.py, masactrl/test.py on the small XL family"""
    import json
    rng = np.random.RandomState(0)
    img = np.kron(rng.randint(0, 255, (8, 8, 3)), np.ones((16, 16, 1))).astype(np.uint8)
    Image.fromarray(img).save(tmp_path / "test.jpg")
    pkg = os.path.join(ROOT, "image-editing-framework_amd")
    jobs = [([os.path.join(pkg, "masactrl", "edit_real.py"), "--sd_version", "smallxl", "--inversion_type", "ddim", "--source_image",
              str(tmp_path / "test.jpg")], tmp_path)]
    for folder, inv in (("p2p", "null-text"), ("masactrl", "ddim")):
        jobs.append(([os.path.join(pkg, folder, "test.py"), "--sd_version", "smallxl", "--synthetic", "1", "--inversion_type", inv,
                      "--exp_path", str(tmp_path / folder)], tmp_path / ("cwd_" + folder)))
    done = run_mixed(jobs, in_process=(1,))
    assert (tmp_path / "exp" / "edit.png").exists()
    for d in done[1:]:
        assert d.last_json()["images"] == 1, d.args


def test_xl_pie_drivers_batched_and_in_flight_match_per_image(tmp_path):
    """`--invert_batch` / `--in_flight` on the SDXL family (round 2 raised NotImplementedError): batched `ddim_inversion_xl`,
    `P2P_XL.edit_many` / `PnP_XL.edit_many` with the prompts' `added_cond_kwargs`, `NTI_XL.null_optimization_many` -- the
    schedules regroup independent images, so the PNGs must equal the per-image run's (`/root/reference/p2p/test.py:114-181`
    is the per-image loop)"""
    import json
    pkg = os.path.join(ROOT, "image-editing-framework_amd")

    def drive(folder, out, *flags):
        d = call_main(os.path.join(pkg, folder, "test.py"), ["--sd_version", "smallxl", "--synthetic", "2", "--exp_path", str(out)]
                      + list(flags), tmp_path / ("cwd_" + out.name))
        assert d.last_json()["images"] == 2, d.args

    def same(a, b):
        dirs = sorted(x for x in os.listdir(a) if x.startswith("syn_"))
        assert len(dirs) == 2
        for d in dirs:
            for name in ("inversion.png", "edit.png"):
                pa, pb = np.array(Image.open(a / d / name)).astype(int), np.array(Image.open(b / d / name)).astype(int)
                assert pa.shape == pb.shape and np.abs(pa - pb).max() <= 1, (d, name, np.abs(pa - pb).max())

    # null-text: in flight (every mode) the values are the per-image run's bit for bit.  A BATCHED inversion is bit-identical
    # per row only on the fp16-storage path (its kernels' tiles and split-K do not depend on the batch); in the fp32-storage
    # modes a row of a batch-2 launch differs from the batch-1 launch in the last bits (3e-6 on the latents: split-K follows
    # the row count), which the null-text optimisation of a RANDOM-weight net amplifies without bound -- so the batched
    # schedule is pinned where it is exact
    pairs = []
    for folder in ("p2p", "pnp"):
        pairs.append((folder, tmp_path / (folder + "_one"), (), tmp_path / (folder + "_many"), ("--invert_batch", "2", "--in_flight", "2")))
    pairs.append(("p2p", tmp_path / "nti_one", ("--inversion_type", "null-text"),
                  tmp_path / "nti_many", ("--inversion_type", "null-text", "--in_flight", "2")))
    pairs.append(("p2p", tmp_path / "nti16_one", ("--inversion_type", "null-text", "--precision", "f16"),
                  tmp_path / "nti16_many", ("--inversion_type", "null-text", "--precision", "f16", "--invert_batch", "2", "--in_flight", "2")))
    for folder, one, f1, many, f2 in pairs:            # eight driver runs, in this process (tests/_procs.py)
        drive(folder, one, *f1)
        drive(folder, many, *f2)
    for folder, one, f1, many, f2 in pairs:
        same(one, many)


# ------------------------------------------------------------------------------------------------ Plug-and-Play on the XL family
def test_pnp_xl_forward_loop_and_cli(xlpipe, tmp_path):
    """`PnP_XL`: the `_xl` injection sites (self-attention of every transformer block of up_blocks[1]; conv2 output of
    up_blocks[1].resnets[0]) against the oracle, the sampler loop, and the CLI dispatch"""
    from ief_amd.pnp.model.register import (register_attention_control_efficient_xl, register_conv_control_efficient_xl,
                                            register_time_xl, unregister_attention_control_efficient_xl,
                                            unregister_conv_control_efficient_xl)
    from ief_amd.pnp.model.sd_utils import PnP_XL
    from ief_amd.p2p.model.sd_utils import encode_prompt_xl
    from oracle import pnp_ref
    cfg = xlpipe.cfg
    sd = xlpipe._state_dict
    xlpipe.scheduler.set_timesteps(10)
    ts = xlpipe.scheduler.timesteps
    x, ctx, added = _inputs(cfg, 4, seed=31)
    t = int(ts[0])
    added_dev = {k: v.to(DEV) for k, v in added.items()}
    ref = pnp_ref.pnp_forward(sd, cfg, x, t, ctx, True, True, added)
    plain = unet_ref.unet_forward(sd, cfg, x, t, ctx, added_cond_kwargs=added)
    register_attention_control_efficient_xl(xlpipe, ts[:10])
    register_conv_control_efficient_xl(xlpipe, ts[:10])
    try:
        register_time_xl(xlpipe, t)
        got = xlpipe.unet(x.to(DEV), t, encoder_hidden_states=ctx.to(DEV), added_cond_kwargs=added_dev)["sample"]
    finally:
        unregister_attention_control_efficient_xl(xlpipe)
        unregister_conv_control_efficient_xl(xlpipe)
    assert xlpipe.unet._plan is None
    e, moved = rel_err(got, ref), rel_err(plain, ref)
    print(f"PnP_XL forward: {e:.2e}; the injection moves the output by {moved:.2e}")
    assert e < 2e-2 and moved > 10 * e
    # sampler loop
    steps = 5
    size = cfg.sample_size * 8
    sched = p2p_ref.DDIMRef(num_inference_steps=steps)
    x_T = torch.randn(1, 4, cfg.sample_size, cfg.sample_size, generator=torch.Generator().manual_seed(32))
    editor = PnP_XL(xlpipe, steps)
    emb, add2 = encode_prompt_xl(xlpipe, PROMPTS, DEV, True, size, size, 2)
    ref_lat = pnp_ref.pnp_loop(sd, cfg, emb.float().cpu(), x_T, sched, 7.5, pnp_attn_t=0.6, pnp_f_t=1.0,
                               added_cond_kwargs={k: v.float().cpu() for k, v in add2.items()})
    lat = editor(prompt=PROMPTS, num_inference_steps=steps, guidance_scale=7.5, pnp_attn_t=0.6, pnp_f_t=1.0, latents=x_T,
                 return_latents=True)
    e2 = rel_err(lat, ref_lat)
    print(f"PnP_XL {steps}-step loop: {e2:.2e}")
    assert e2 < 5e-2 and xlpipe.unet._plan is None
    r = subprocess.run([sys.executable, os.path.join(ROOT, "image-editing-framework_amd", "pnp", "edit_syn.py"),
                        "--sd_version", "smallxl"], cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert (tmp_path / "exp" / "edit.png").exists()
