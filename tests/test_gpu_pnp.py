"""Plug-and-Play (SURVEY.md §8f rank 4, SD1.x shape family) on a real MI355X against the fp32 CPU oracle.

Tolerances as in test_gpu_unet.py: one injected UNet forward <= 2e-2 of max|reference| (measured ~2e-3), a short loop
<= 5e-2; the injection itself must move the output by far more than the tolerance, or the test would prove nothing.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
from PIL import Image

from _procs import run_mixed

pytestmark = pytest.mark.gpu

from ief_amd import hip  # noqa: E402
from ief_amd.pipeline import StableDiffusionPipeline  # noqa: E402
from ief_amd.p2p.model.sd_utils import _encode_prompts  # noqa: E402
from ief_amd.pnp.model.register import (register_attention_control_efficient, register_conv_control_efficient,  # noqa: E402
                                        register_time, unregister_attention_control_efficient,
                                        unregister_conv_control_efficient)
from ief_amd.pnp.model.sd_utils import PnP, PnP_NTI  # noqa: E402
from oracle import p2p_ref, pnp_ref, unet_ref  # noqa: E402

DEV = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROMPTS = ["a green apple on a wooden table", "a red apple on a wooden table"]


def rel_err(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert torch.isfinite(got).all()
    return ((got - ref).abs().max() / ref.abs().max()).item()


@pytest.fixture(scope="module")
def tiny():
    return StableDiffusionPipeline.from_pretrained("synthetic:tiny", keep_state_dict=True)


def test_gather_rows():
    x = (torch.randn(4, 16, 16, 64) * 1.0).half().to(DEV)
    src = torch.tensor([0, 2, 2, 2], dtype=torch.int32, device=DEV)
    out = hip.gather_rows(x, src)
    assert torch.equal(out, x[src.long()])


@pytest.mark.parametrize("qk,conv", [(True, False), (False, True), (True, True)])
def test_pnp_forward_vs_oracle(tiny, qk, conv):
    cfg = tiny.cfg
    tiny.scheduler.set_timesteps(10)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 4, cfg.sample_size, cfg.sample_size, generator=g)
    ctx = torch.randn(4, 77, cfg.cross_attention_dim, generator=g) * 0.1
    t = int(tiny.scheduler.timesteps[0])
    ref = pnp_ref.pnp_forward(tiny._state_dict, cfg, x, t, ctx, qk, conv)
    plain = unet_ref.unet_forward(tiny._state_dict, cfg, x, t, ctx)
    ts = tiny.scheduler.timesteps
    register_attention_control_efficient(tiny, ts[:10] if qk else ts[:0])
    register_conv_control_efficient(tiny, ts[:10] if conv else ts[:0])
    try:
        register_time(tiny, t)
        got = tiny.unet(x.to(DEV), t, encoder_hidden_states=ctx.to(DEV))["sample"]
        # both schedules empty: no injection at this timestep
        register_attention_control_efficient(tiny, ts[:0])
        register_conv_control_efficient(tiny, ts[:0])
        register_time(tiny, t)
        off = tiny.unet(x.to(DEV), t, encoder_hidden_states=ctx.to(DEV))["sample"]
    finally:
        unregister_attention_control_efficient(tiny)
        unregister_conv_control_efficient(tiny)
    assert tiny.unet._plan is None
    e, moved, e_off = rel_err(got, ref), rel_err(plain, ref), rel_err(off, plain)
    print(f"PnP forward qk={qk} conv={conv}: err {e:.2e}; injection moves the output by {moved:.2e}; schedule off {e_off:.2e}")
    assert e < 2e-2 and e_off < 2e-2 and moved > 10 * e


def test_pnp_loop_vs_oracle_graph_and_eager(tiny):
    cfg = tiny.cfg
    steps = 6
    sched = p2p_ref.DDIMRef(num_inference_steps=steps)
    x_T = torch.randn(1, 4, cfg.sample_size, cfg.sample_size, generator=torch.Generator().manual_seed(3))
    editor = PnP(tiny, steps)
    with torch.no_grad():
        u, c = _encode_prompts(tiny, PROMPTS)
    ctx = torch.cat([u, c]).float().cpu()
    ref = pnp_ref.pnp_loop(tiny._state_dict, cfg, ctx, x_T, sched, 7.5, pnp_attn_t=0.67, pnp_f_t=1.0)
    none = pnp_ref.pnp_loop(tiny._state_dict, cfg, ctx, x_T, sched, 7.5, pnp_attn_t=0.0, pnp_f_t=0.0)
    got = editor(prompt=PROMPTS, num_inference_steps=steps, guidance_scale=7.5, pnp_attn_t=0.67, pnp_f_t=1.0, latents=x_T,
                 return_latents=True)
    eager = editor(prompt=PROMPTS, num_inference_steps=steps, guidance_scale=7.5, pnp_attn_t=0.67, pnp_f_t=1.0, latents=x_T,
                   return_latents=True, use_graph=False)
    e, e2, moved = rel_err(got, ref), rel_err(eager, ref), rel_err(none, ref)
    print(f"PnP {steps}-step loop: graph {e:.2e}, eager (register_time per step) {e2:.2e}; injection moves latents by {moved:.2e}")
    assert e < 5e-2 and e2 < 5e-2 and e == e2 and moved > 2 * e
    images = editor(prompt=PROMPTS, num_inference_steps=steps, guidance_scale=7.5, pnp_attn_t=0.67, pnp_f_t=1.0, latents=x_T)
    assert images.shape == (2, cfg.sample_size * 8, cfg.sample_size * 8, 3) and images.dtype == np.uint8


def test_pnp_nti_per_step_unconditional_rows(tiny):
    """`PnP_NTI`: both unconditional rows take the null-text embedding of step i (`pnp/model/sd_utils.py:340`)"""
    cfg = tiny.cfg
    steps = 5
    sched = p2p_ref.DDIMRef(num_inference_steps=steps)
    x_T = torch.randn(1, 4, cfg.sample_size, cfg.sample_size, generator=torch.Generator().manual_seed(4))
    g = torch.Generator().manual_seed(9)
    rows = [torch.randn(1, 77, cfg.cross_attention_dim, generator=g) * 0.1 for _ in range(steps)]
    editor = PnP_NTI(tiny, steps)
    with torch.no_grad():
        u, c = _encode_prompts(tiny, PROMPTS)
    ctx = torch.cat([u, c]).float().cpu()
    ref = pnp_ref.pnp_loop(tiny._state_dict, cfg, ctx, x_T, sched, 7.5, pnp_attn_t=1.0, pnp_f_t=1.0, uncond_list=rows)
    fixed = pnp_ref.pnp_loop(tiny._state_dict, cfg, ctx, x_T, sched, 7.5, pnp_attn_t=1.0, pnp_f_t=1.0)
    got = editor(prompt=PROMPTS, num_inference_steps=steps, guidance_scale=7.5, pnp_attn_t=1.0, pnp_f_t=1.0, latents=x_T,
                 return_latents=True, uncond_embeddings_list=rows)
    e, moved = rel_err(got, ref), rel_err(fixed, ref)
    print(f"PnP_NTI {steps}-step loop: {e:.2e}; the per-step rows move the latents by {moved:.2e}")
    assert e < 5e-2 and moved > 2 * e
    with pytest.raises(ValueError):
        editor(prompt=PROMPTS, num_inference_steps=steps, latents=x_T)


def test_pnp_rejects_non_prefix_schedule(tiny):
    tiny.scheduler.set_timesteps(10)
    ts = tiny.scheduler.timesteps
    with pytest.raises(ValueError):
        register_attention_control_efficient(tiny, ts[2:5])
    unregister_attention_control_efficient(tiny)
    unregister_conv_control_efficient(tiny)
    assert tiny.unet._plan is None


def test_pnp_clis(tmp_path):
    pnp = os.path.join(ROOT, "image-editing-framework_amd", "pnp")
    rng = np.random.RandomState(0)
    img = np.kron(rng.randint(0, 255, (8, 8, 3)), np.ones((16, 16, 1))).astype(np.uint8)
    Image.fromarray(img).save(tmp_path / "test.jpg")
    jobs = [([os.path.join(pnp, "edit_syn.py"), "--sd_version", "tiny"], tmp_path / "syn")]
    for inv in ("ddim", "null-text"):
        jobs.append(([os.path.join(pnp, "edit_real.py"), "--sd_version", "tiny", "--inversion_type", inv, "--source_image",
                      str(tmp_path / "test.jpg")], tmp_path / inv))
    run_mixed(jobs, in_process=(2,))
    src = np.array(Image.open(tmp_path / "syn" / "exp" / "source.png")).astype(int)
    edit = np.array(Image.open(tmp_path / "syn" / "exp" / "edit.png")).astype(int)
    assert src.shape == edit.shape == (128, 128, 3) and np.abs(src - edit).max() > 0
    for inv in ("ddim", "null-text"):
        for name in ("source.png", "inversion.png", "edit.png"):
            assert (tmp_path / inv / "exp" / name).exists()


def test_sd21_shape_family_forward_and_pnp():
    """SD2.1 geometry (linear proj_in / proj_out, head dim 64, context 1024) on the two-level net: plain and injected
    forward against the oracle"""
    pipe = StableDiffusionPipeline.from_pretrained("synthetic:small21", keep_state_dict=True)
    cfg = pipe.cfg
    assert cfg.use_linear_projection and pipe._state_dict["mid_block.attentions.0.proj_in.weight"].dim() == 2
    pipe.scheduler.set_timesteps(10)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 4, cfg.sample_size, cfg.sample_size, generator=g)
    ctx = torch.randn(4, 77, cfg.cross_attention_dim, generator=g) * 0.1
    t = int(pipe.scheduler.timesteps[0])
    plain = unet_ref.unet_forward(pipe._state_dict, cfg, x, t, ctx)
    e0 = rel_err(pipe.unet(x.to(DEV), t, encoder_hidden_states=ctx.to(DEV))["sample"], plain)
    # the two-level net has a single decoder block with attention at index 1: PnP's block table (1: [1, 2]) applies
    ref = pnp_ref.pnp_forward(pipe._state_dict, cfg, x, t, ctx, True, True)
    ts = pipe.scheduler.timesteps
    register_attention_control_efficient(pipe, ts[:10])
    register_conv_control_efficient(pipe, ts[:10])
    try:
        register_time(pipe, t)
        e1 = rel_err(pipe.unet(x.to(DEV), t, encoder_hidden_states=ctx.to(DEV))["sample"], ref)
    finally:
        unregister_attention_control_efficient(pipe)
        unregister_conv_control_efficient(pipe)
    print(f"small21: plain forward {e0:.2e}, PnP forward {e1:.2e}; injection moves the output by {rel_err(plain, ref):.2e}")
    assert e0 < 2e-2 and e1 < 2e-2


def test_pnp_pie_driver_sd21_family(tmp_path):
    pnp = os.path.join(ROOT, "image-editing-framework_amd", "pnp")
    r = subprocess.run([sys.executable, os.path.join(pnp, "test.py"), "--sd_version", "small21", "--synthetic", "2",
                        "--invert_batch", "2", "--exp_path", str(tmp_path / "t")], cwd=str(tmp_path), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["images"] == 2 and rec["images_per_sec"] > 0


def test_pnp_edit_many_equals_one_at_a_time(tiny, tmp_path):
    """several Plug-and-Play edits in flight (`PnP.edit_many`): the same pixels as one sampler call per image"""
    cfg = tiny.cfg
    steps = 6
    editor = PnP(tiny, steps)
    g = torch.Generator().manual_seed(21)
    xs = [torch.randn(1, 4, cfg.sample_size, cfg.sample_size, generator=g) for _ in range(3)]
    prompts = [PROMPTS, ["a dog on the grass", "a cat on the grass"], PROMPTS[::-1]]
    kw = dict(num_inference_steps=steps, guidance_scale=7.5, pnp_attn_t=0.67, pnp_f_t=1.0)
    one = [editor(prompt=p, latents=torch.cat([x, x]), **kw) for p, x in zip(prompts, xs)]
    many = editor.edit_many([(p, torch.cat([x, x])) for p, x in zip(prompts, xs)], **kw)
    assert tiny.unet._plan is None
    for a, b in zip(one, many):
        assert a.shape == b.shape and np.array_equal(a, b)
    again = editor.edit_many([(p, torch.cat([x, x])) for p, x in zip(prompts, xs)], **kw)     # pooled graphs, re-pointed
    assert all(np.array_equal(a, b) for a, b in zip(one, again))
    pnp = os.path.join(ROOT, "image-editing-framework_amd", "pnp")
    r = subprocess.run([sys.executable, os.path.join(pnp, "test.py"), "--sd_version", "tiny", "--synthetic", "3",
                        "--invert_batch", "3", "--in_flight", "2", "--exp_path", str(tmp_path / "t")], cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    assert json.loads(r.stdout.strip().splitlines()[-1])["images"] == 3
