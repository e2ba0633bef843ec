"""SDXL shape family (BASELINE.json config 5) on a real MI355X against the fp32 CPU oracle: three levels with no attention
at the first, several BasicTransformerBlocks per Transformer2DModel (LayerNorm folding chained block to block), head dim
64, linear projections, the text-time additional embedding.  Tolerances as in test_gpu_unet.py: one UNet forward
<= 2e-2 of max |reference| (measured ~2e-3)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from ief_amd import config, hip, weights  # noqa: E402
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
    assert e0 < 5e-2 and e1 < 5e-2


def test_p2p_xl_cli(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "image-editing-framework_amd", "p2p", "edit_syn.py"),
                        "--sd_version", "smallxl"], cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    src = np.array(Image.open(tmp_path / "exp" / "source.png")).astype(int)
    edit = np.array(Image.open(tmp_path / "exp" / "edit.png")).astype(int)
    assert src.shape == edit.shape == (128, 128, 3) and np.abs(src - edit).max() > 0
