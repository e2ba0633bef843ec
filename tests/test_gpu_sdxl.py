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
