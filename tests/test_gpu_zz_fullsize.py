"""Parity at the sizes BASELINE.json's configurations actually run, on a real MI355X against the fp32 CPU oracle.

  * the workload `bench.py` times: SD1.5 shapes, 64x64 latents, UNet batch 4 (2 prompts x CFG), AttentionRefine lowered
    into the fused kernels, at two points of the schedule (self-replace window open / closed);
  * MasaCtrl's mutual self-attention at SD1.5 size (BASELINE.json configs[2]), step >= 4 and layer >= 10 active;
  * the SD2.1 shape family (96x96 latents, d = 64, context 1024) at FULL size with both Plug-and-Play injections active
    (configs[3]), the SDXL family (128x128 latents, depth-10 transformers, additional embedding; configs[4]) and the SD1.5
    net on 128x128 latents (N = 16384 self-attention) at FULL size;
  * the full 50-step edit (fixture G13), and the reverse passes at the timed sizes;
every one of them in BOTH storage modes that ship as defaults: `precision="f16x3"` (fp32 storage, split-operand contractions:
the mode that meets north_star's 1e-3 image bound and the default of all four method folders) and `"f16"`.

The oracle's outputs are FIXTURES (G15, `tests/golden/fullsize_eps.npz`, made in the build container by
`tests/golden/make_golden_fullsize.py` from the same seeded weights and inputs): the GPU suite no longer runs 10-20 s CPU
forwards (they cost ~170 s of the suite and put it 64 s from the driver's 900 s limit).

Stated tolerance for ONE forward, max |eps - eps_oracle| / max |eps_oracle|:
    f16x3   <= 1e-4   (measured 2-4e-6: operand split 2^-22, fp32 accumulation in another order than the CPU's;
                       the oracle's own run-to-run spread with the thread count is ~2e-6)
    f16     <= 5e-3   (measured 1.7-3.6e-3: fp16 operand rounding)

This module runs FIRST among the GPU tests (tests/conftest.py orders the files by what they pin; the suite-wide time guard --
IEF_GPU_SUITE_BUDGET, default 780 s of the driver's 900 s -- lives there too and skips whatever is left at the least critical end).
"""
import gc
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from ief_amd import config  # noqa: E402
from ief_amd.pipeline import StableDiffusionPipeline  # noqa: E402
from ief_amd.p2p.model.attention_control import AttentionRefine  # noqa: E402
from ief_amd.p2p.model.register import register_attention_control, unregister_attention_control  # noqa: E402
from oracle import p2p_ref  # noqa: E402

DEV = torch.device("cuda:0")
PROMPTS = ["a photo of a house on a mountain", "a photo of a house on a mountain at fall"]   # edit_syn.py:20-21
FWD_TOL = {"f16x3": 1e-4, "f16": 5e-3}
FULL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_eps.npz")
PRECISIONS = ["f16x3", "f16"]


def rel_err(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert torch.isfinite(got).all()
    return ((got - ref).abs().max() / ref.abs().max()).item()


def _inputs(cfg, B, seed=0, hw=None):
    g = torch.Generator().manual_seed(seed)
    hw = hw or cfg.sample_size
    x = torch.randn(B, 4, hw, hw, generator=g)
    ctx = torch.randn(B, 77, cfg.cross_attention_dim, generator=g)
    return x, ctx


def oracle_eps(name, x, ctx):
    """the CPU oracle's eps of fixture G15 for this case, after checking that the test regenerated the generator's inputs"""
    g = np.load(FULL)
    probe = np.concatenate([x.flatten()[:8].numpy(), ctx.flatten()[:8].numpy()])
    assert np.array_equal(probe, g[name + "__probe"]), f"{name}: the regenerated inputs are not the fixture's"
    return torch.from_numpy(g[name])


_PIPES = {}


def pipe_of(family, precision):
    """one model per (shape family, precision), kept while consecutive tests use it; a new family drops the others (HBM is
    not the limit: the host-side fp32 state dicts are)"""
    key = (family, precision)
    if key not in _PIPES:
        for k in [k for k in _PIPES if k[0] != family]:
            del _PIPES[k]
        gc.collect()
        torch.cuda.empty_cache()
        _PIPES[key] = StableDiffusionPipeline.from_pretrained(f"synthetic:{family}", precision=precision)
    return _PIPES[key]


def _p2p_batch(cfg, seed):
    """[uncond_src, uncond_tgt, cond_src, cond_tgt]: source / target latents already diverged, CFG-duplicated; unit-variance
    embeddings (peaky cross-attention maps, so the edit moves the output)"""
    x1, ctx = _inputs(cfg, 4, seed=seed)
    x = torch.cat([x1[:1], 0.8 * x1[:1] + 0.6 * x1[1:2]] * 2)
    return x, ctx


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("step", [0, 25])
def test_sd15_p2p_refine_step_b4_fused_vs_oracle(step, precision):
    """THE timed workload (bench.py): one P2P edit step's UNet forward at batch 4 with AttentionRefine in the fused
    kernels, against the oracle running the Python controller on materialised fp32 maps
    (`/root/reference/p2p/model/attention_base.py:113-136`).  step 0: cross edit + self-replace (N <= 256) active;
    step 25: self-replace window (0.4 x 50 = 20) closed, cross edit still gated on."""
    cfg = config.SD15
    pipe = pipe_of("sd15", precision)
    x, ctx = _p2p_batch(cfg, seed=3)
    ref = oracle_eps(f"sd15_refine_step{step}", x, ctx)
    t = int(p2p_ref.DDIMRef(50).timesteps[step])
    c = AttentionRefine(PROMPTS, pipe.tokenizer, 50, 0.8, 0.4, device=DEV)
    register_attention_control(pipe, c, fused=True)
    c.cur_step = step
    got = pipe.unet(x.to(DEV), t, encoder_hidden_states=ctx.to(DEV))["sample"].cpu()
    assert c.cur_step == step + 1 and c.cur_att_layer == 0 and c.num_att_layers == 32
    unregister_attention_control(pipe, c)
    e = rel_err(got, ref)
    effect = rel_err(pipe.unet(x.to(DEV), t, encoder_hidden_states=ctx.to(DEV))["sample"], ref)   # the product's own uncontrolled forward
    print(f"sd15 B=4 AttentionRefine step {step} (t={t}) {precision}: fused-vs-oracle {e:.3e}; the edit itself moves eps by {effect:.3e}")
    assert e < FWD_TOL[precision]
    assert effect > 10 * e, "the control must change the output by far more than the kernel error"


@pytest.mark.parametrize("precision", PRECISIONS)
def test_sd15_masactrl_mutual_step_b4_vs_oracle(precision):
    """BASELINE.json configs[2] at SD1.5 size: MutualSelfAttentionControl(4, 10) at step 6 (active: step >= 4, layers 10-15 take
    the source row's K / V, `/root/reference/masactrl/model/attention_control.py:37-68`) against the oracle's q/k/v hook"""
    from ief_amd.masactrl.model.attention_control import MutualSelfAttentionControl
    from ief_amd.masactrl.model.register import regiter_attention_editor_diffusers, unregister_attention_control as unreg
    cfg = config.SD15
    pipe = pipe_of("sd15", precision)
    x1, ctx = _inputs(cfg, 4, seed=9)
    x = torch.cat([x1[:1], 0.6 * x1[:1] + 0.8 * x1[1:2]] * 2)
    ref = oracle_eps("sd15_masactrl_step6", x, ctx)
    c = MutualSelfAttentionControl(4, 10, total_steps=50)
    regiter_attention_editor_diffusers(pipe, c)
    assert c.num_att_layers == 32
    c.cur_step = 6
    got = pipe.unet(x.to(DEV), 501, encoder_hidden_states=ctx.to(DEV))["sample"].cpu()
    assert c.cur_step == 7 and c.cur_att_layer == 0
    unreg(pipe, c)
    plain = pipe.unet(x.to(DEV), 501, encoder_hidden_states=ctx.to(DEV))["sample"].cpu()
    e, effect = rel_err(got, ref), rel_err(plain, ref)
    print(f"sd15 B=4 MasaCtrl mutual self-attention step 6 {precision}: fused-vs-oracle {e:.3e}; the control moves eps by {effect:.3e}")
    assert e < FWD_TOL[precision] and effect > 10 * e


@pytest.mark.parametrize("precision", PRECISIONS)
def test_sd15_1024px_forward_b1(precision):
    """the SD1.5-shaped net on 128x128 latents (1024x1024 px, north_star's second latent size): N = 16384 self-attention at
    d = 40 inside the whole UNet -- in f16x3 this is `attn_flash_x3_kernel` at the sequence length bench.py times"""
    cfg = config.SD15
    pipe = pipe_of("sd15", precision)
    x, ctx = _inputs(cfg, 1, seed=17, hw=128)
    ctx = ctx * 0.1
    ref = oracle_eps("sd15_1024_b1", x, ctx)
    eps = pipe.unet(x.to(DEV), 481, encoder_hidden_states=ctx.to(DEV))["sample"]
    e = rel_err(eps, ref)
    print(f"sd15 B=1 1024^2 (128x128 latents) {precision}: rel err {e:.3e}")
    assert e < FWD_TOL[precision]


@pytest.mark.parametrize("precision", PRECISIONS)
def test_sd21_full_size_pnp_injected_forward_b4(precision):
    """BASELINE.json configs[3]: Plug-and-Play on the SD2.1 shape family at 768x768 — 96x96 latents (N = 9216 self-attention
    at d = 64), OpenCLIP context 1024, linear projections — one forward at the sampler's batch 4 with BOTH injections active
    (self-attention Q / K of decoder blocks 4-11 and the conv feature of `up_blocks[1].resnets[1]` taken from the source
    rows, `/root/reference/pnp/model/register.py:45-52,161-166`) against the oracle's hooks"""
    from ief_amd.pnp.model.register import (register_attention_control_efficient, register_conv_control_efficient, register_time,
                                            unregister_attention_control_efficient, unregister_conv_control_efficient)
    cfg = config.SD21
    pipe = pipe_of("sd21", precision)
    pipe.scheduler.set_timesteps(50)
    x, ctx = _inputs(cfg, 4, seed=11)
    ctx = ctx * 0.1
    ref = oracle_eps("sd21_pnp_b4", x, ctx)
    ts = pipe.scheduler.timesteps
    t = int(ts[0])
    register_attention_control_efficient(pipe, ts[:25])
    register_conv_control_efficient(pipe, ts[:40])
    try:
        register_time(pipe, t)
        got = pipe.unet(x.to(DEV), t, encoder_hidden_states=ctx.to(DEV))["sample"].cpu()
    finally:
        unregister_attention_control_efficient(pipe)
        unregister_conv_control_efficient(pipe)
    plain = pipe.unet(x.to(DEV), t, encoder_hidden_states=ctx.to(DEV))["sample"].cpu()
    e, moved = rel_err(got, ref), rel_err(plain, ref)
    print(f"sd21 B=4 768^2 (96x96 latents) PnP-injected forward {precision}: rel err {e:.3e}; the injection moves eps by {moved:.3e}")
    assert e < FWD_TOL[precision] and moved > 10 * e


@pytest.mark.parametrize("precision", PRECISIONS)
def test_sdxl_full_size_forward_b1(precision):
    """SDXL base shape family at 1024x1024 (BASELINE.json configs[4]): 128x128 latents, transformer depth 1 / 2 / 10
    (LayerNorm folding chained through ten blocks on the fp16 path), d = 64, context 2048, text-time additional embedding"""
    cfg = config.SDXL
    pipe = pipe_of("sdxl", precision)
    x, ctx = _inputs(cfg, 1, seed=13)
    ctx = ctx * 0.1
    ref = oracle_eps("sdxl_b1", x, ctx)
    g = torch.Generator().manual_seed(14)
    added = {"text_embeds": torch.randn(1, cfg.pooled_text_dim, generator=g) * 0.5,
             "time_ids": torch.tensor([[1024.0, 1024.0, 0.0, 0.0, 1024.0, 1024.0]])}
    eps = pipe.unet(x.to(DEV), 481, encoder_hidden_states=ctx.to(DEV),
                    added_cond_kwargs={k: v.to(DEV) for k, v in added.items()})["sample"]
    e = rel_err(eps, ref)
    print(f"sdxl B=1 1024^2 (128x128 latents) {precision}: rel err {e:.3e}")
    assert e < FWD_TOL[precision]


# ------------------------------------------------------------------------------------------- the reference's unit of work
EDIT50 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sd15_edit50.npz")
# image bound: north_star's "edited images within 1e-3 max-abs of reference" for the modes that compute at the reference's
# precision; the fp16-storage path is held to what it measures (printed), not to 1e-3
# (measured on MI355X: latents 2.0e-6 / 2.7e-6 / 1.4e-3, image crop 4.3e-6 / 5.3e-6 / 3.0e-3 for f16x3 / f32 / f16).
# The fixture itself is reproducible only to the CPU's summation order: re-running its generator with another thread count moves
# lat_10 / 25 / 50 by 1.4-1.7e-6 of max |latent| (img_crop 5.3e-6, one grey level on a few uint8 pixels) -- the fp32-storage
# modes sit AT that noise floor, and the 1e-5 latent bound is 6x above it; the generator now pins and records its thread count.
EDIT50_BOUNDS = {"f32": dict(lat=1e-5, img=1e-3, u8=1), "f16x3": dict(lat=1e-5, img=1e-3, u8=1), "f16": dict(lat=4e-3, img=8e-3, u8=2)}


@pytest.mark.parametrize("precision", ["f16x3", "f32", "f16"])
def test_sd15_edit50_vs_oracle_fixture(precision):
    """ONE FULL 50-step Prompt-to-Prompt edit at SD1.5 size (`/root/reference/p2p/model/sd_utils.py:24-65`: 512x512, UNet
    batch 4, CLI default prompts, AttentionRefine 0.8 / 0.4, guidance 7.5, seed 8888) in the captured step graph, against
    the trajectory `oracle.p2p_ref.edit_loop` computed on the CPU in fp32 (fixture G13, `tests/golden/make_golden_edit50.py`,
    ~25 CPU-minutes: not recomputable inside the GPU suite): latents after 10 / 25 / 50 steps, then the VAE-decoded images
    (centre crop at full resolution + the whole image average-pooled 8x8) in [0, 1] and as uint8."""
    from ief_amd.denoise import acquire
    from ief_amd.p2p.model.sd_utils import _encode_prompts
    g = np.load(EDIT50)
    pipe = pipe_of("sd15", precision)
    cfg = config.SD15
    n = int(g["steps"])
    pipe.scheduler.set_timesteps(n)
    with torch.no_grad():
        u, c = _encode_prompts(pipe, PROMPTS)
    context = torch.cat([u, c]).float()
    probe, sums = torch.from_numpy(g["context_probe"]), g["context_sums"]
    assert torch.allclose(context[:, :8, :16].cpu(), probe, rtol=0, atol=1e-6), "the regenerated context is not the fixture's"
    assert abs(context.double().sum().item() - sums[0]) < 1e-2 and abs(context.double().abs().sum().item() - sums[1]) < 1e-2
    x_T = torch.from_numpy(g["x_T"])
    assert torch.equal(x_T, torch.randn(1, 4, cfg.sample_size, cfg.sample_size, generator=torch.Generator().manual_seed(8888)))
    ctl = AttentionRefine(PROMPTS, pipe.tokenizer, n, 0.8, 0.4, device=DEV)
    register_attention_control(pipe, ctl)
    assert pipe.unet._plan is not None and pipe.unet._plan.kind == "p2p"
    loop = acquire(pipe, context.to(DEV), 2, (cfg.sample_size, cfg.sample_size), 7.5)
    assert loop.use_graph
    lat, traj = loop.run(x_T.to(DEV).expand(2, -1, -1, -1), keep_all=True)
    loop.release()
    assert ctl.cur_step == n and ctl.cur_att_layer == 0
    unregister_attention_control(pipe, ctl)
    b = EDIT50_BOUNDS[precision]
    errs = {k: rel_err(traj[k], torch.from_numpy(g[f"lat_{k}"])) for k in (10, 25, 50)}
    dec = pipe.vae.decode(lat / pipe.vae.config.scaling_factor)["sample"].float().cpu()
    a = int(g["crop_origin"])
    to01 = lambda t: (t / 2 + 0.5).clamp(0, 1)
    d_crop = (to01(dec[:, :, a:a + 64, a:a + 64]) - to01(torch.from_numpy(g["img_crop"]))).abs().max().item()
    d_pool = (torch.nn.functional.avg_pool2d(dec, 8) - torch.from_numpy(g["img_pool8"])).abs().max().item() / 2
    u8 = p2p_ref.latent_to_uint8(dec)[:, a:a + 64, a:a + 64].astype(int) - g["img_u8_crop"].astype(int)
    print(f"sd15 50-step edit, precision={precision}: latents rel err after 10 / 25 / 50 steps "
          f"{errs[10]:.3e} / {errs[25]:.3e} / {errs[50]:.3e}; decoded images in [0,1]: crop max |diff| {d_crop:.3e}, "
          f"8x8-pooled whole image {d_pool:.3e}; uint8 crop: max diff {abs(u8).max()}, identical {(u8 == 0).mean():.4f}")
    del loop
    assert max(errs.values()) < b["lat"] and d_crop <= b["img"] and d_pool <= b["img"] and abs(u8).max() <= b["u8"]


# ------------------------------------------------------------------------------------------- reverse passes at the timed sizes
@pytest.mark.parametrize("precision", PRECISIONS)
def test_sd15_null_text_inner_iterations_at_full_size(precision):
    """BASELINE.json configs[1] at the size `tests/bench_nti.py` times: SD1.5 shapes, 64x64 latents, UNet batch 1, the TUNED
    reverse-pass plans (split-K, large tiles) that the small-net tests never select.  No CPU oracle fits (an autograd pass
    through 860 M parameters per iteration), so properties: finite; one timestep of eight Adam steps lowers the objective
    (`/root/reference/p2p/inversion/nti.py:26-29`); graph replay == eager launches bit for bit; and the context gradient
    agrees with the same pass run on heuristic plans (plan table disabled: other tiles, other split-K) to fp16 rounding
    (f16x3: the fp32 reverse pass has one split-K rule and no table, so the two passes are the same launches: 0)."""
    from ief_amd import hip
    from ief_amd.grad import UNetAdjoint
    from ief_amd.nti import NullTextOptimizer
    pipe, cfg = pipe_of("sd15", precision), config.SD15
    hw = cfg.sample_size
    pipe.scheduler.set_timesteps(50)
    g = torch.Generator().manual_seed(0)
    ctx = torch.randn(2, 77, cfg.cross_attention_dim, generator=g) * 0.1
    x_T = torch.randn(1, 4, hw, hw, generator=g)
    lats = [x_T + 0.05 * torch.randn(1, 4, hw, hw, generator=g) for _ in range(50)] + [x_T]
    outs, losses = {}, {}
    for use_graph in (True, False):
        opt = NullTextOptimizer(pipe, ctx[1:], 7.5, (hw, hw), use_graph=use_graph)
        opt.begin([l.to(DEV) for l in lats], ctx[:1])
        opt.outer_begin(0)
        ls = []
        for _ in range(8):
            opt.inner_step()
            ls.append(opt.inner_loss())
        opt.outer_end()
        outs[use_graph], losses[use_graph] = opt.out[-1].cpu(), ls
        opt.release()
    assert all(torch.isfinite(torch.tensor(l)).all() for l in losses.values()) and torch.isfinite(outs[True]).all()
    print(f"sd15 64x64 NTI objective over 8 Adam steps: {[f'{v:.4e}' for v in losses[True]]}")
    assert losses[True][-1] < losses[True][0], "eight Adam steps must lower the null-text objective"
    assert torch.equal(outs[True], outs[False]) and losses[True] == losses[False]
    # the same gradient on heuristic plans
    de = torch.randn(1, 4, hw, hw, generator=g)
    de = (de / de.abs().max()).to(DEV)
    temb = pipe.unet.time_rows(torch.tensor([601.0], device=DEV))
    ctx16 = pipe.unet._act(ctx[:1].to(DEV))
    grads = []
    saved = hip._plan_table()
    for table in (saved, {}):
        hip._plans = table
        adj = UNetAdjoint(pipe.unet)
        adj.forward(x_T.to(DEV), temb, ctx16)
        grads.append(adj.backward(de.contiguous()).float().cpu())
    hip._plans = saved
    e = rel_err(grads[0], grads[1])
    print(f"sd15 64x64 B=1 context gradient, tuned vs heuristic plans: {e:.3e} (max |grad| {grads[1].abs().max():.3e})")
    assert e < 2e-2


def test_sdxl_pix2pix_zero_gradient_step_at_full_size():
    """BASELINE.json configs[4] at full size: the SDXL family at 128x128 latents (1024x1024 px), UNet batch 2 — one
    Pix2Pix-zero reference step and one edit step (`/root/reference/pix2pix-zero/model/sd_utils.py:160-180`) through the
    product's own engine on the tuned plans: the reference pass writes the 70 `attn2` maps, the edit pass runs forward keeping
    the adjoint's inputs, the map objective at every cross-attention module, the reverse pass down to the latents, the SGD
    step, the second forward and the DDIM update.  No CPU oracle fits (autograd through 2.6 G parameters); properties:
    finite, objective > 0, the SGD step along the gradient LOWERS the objective (a first-order property the gradient must
    have), and graph replay == eager launches bit for bit (fixed summation orders)."""
    from ief_amd.pix2pix_zero.model.sd_utils import _Engine
    cfg = config.SDXL
    pipe = StableDiffusionPipeline.from_pretrained("synthetic:sdxl")
    unet = pipe.unet
    hw = cfg.sample_size
    pipe.scheduler.set_timesteps(50)
    t = pipe.scheduler.timesteps.tolist()[20]
    g = torch.Generator().manual_seed(7)
    lat0 = torch.randn(1, 4, hw, hw, generator=g).to(DEV)
    ctx_ref = (torch.randn(2, 77, cfg.cross_attention_dim, generator=g)).to(DEV)
    ctx_edit = (ctx_ref + 0.5 * torch.randn(2, 77, cfg.cross_attention_dim, generator=g).to(DEV)).contiguous()
    size = float(hw * 8)
    added = {"text_embeds": (torch.randn(2, cfg.pooled_text_dim, generator=g) * 0.5).to(DEV),
             "time_ids": torch.tensor([[size, size, 0.0, 0.0, size, size]] * 2, device=DEV)}
    rows = unet.time_rows(torch.tensor([float(t)], device=DEV), unet.aug_embedding(added)).reshape(2, -1)
    coef = torch.tensor([*pipe.scheduler.step_coeffs(t), 7.5, 0.0], device=DEV)
    for m in unet.attention_modules():
        m.cache_kv = False
    res = {}
    for use_graph in (True, False):
        E = _Engine(pipe, hw, hw, 1, 0.1, use_graph, temb_rows=2)
        assert len(E.cross) == 70
        E.temb.copy_(rows), E.coef.copy_(coef)
        E.set_ctx(ctx_ref)
        for m, st in zip(E.cross, E.stage):
            m.map_out = st
        E.lat.copy_(lat0)
        E.ref_step()                                    # the maps of the reference prompt at this step
        for m in E.cross:
            m.map_out = None
        E.set_ctx(ctx_edit)
        E.lat.copy_(lat0)
        E.edit_step()
        l0 = E.step_loss.item()
        lat1, x1 = E.lat.clone(), E.x_in.clone()        # x_in now holds x - lr * grad
        if use_graph:
            E.adj.forward(x1, E.temb, E.ctx16)
            E.adj.backward(E.zero_eps)
            l1 = E.adj.loss_parts.sum().item()
            E.x_in.copy_(lat0.expand_as(E.x_in))
            E.adj.forward(E.x_in, E.temb, E.ctx16)
            gmax = E.adj.backward(E.zero_eps).abs().max().item()
            print(f"sdxl 128x128 B=2 Pix2Pix-zero objective {l0:.5e} -> {l1:.5e} after the SGD step along the gradient "
                  f"(max |scaled grad| {gmax:.3e})")
            assert l0 > 0 and l1 < l0
        res[use_graph] = (l0, lat1.cpu())
        assert torch.isfinite(lat1).all() and (lat1 - lat0).abs().max() > 0
        del E
        gc.collect()
        torch.cuda.empty_cache()
    for m in unet.attention_modules():
        m.cache_kv = True
        m._kv_key, m._kv = None, None
    del pipe
    gc.collect()
    torch.cuda.empty_cache()
    assert res[True][0] == res[False][0] and torch.equal(res[True][1], res[False][1])
