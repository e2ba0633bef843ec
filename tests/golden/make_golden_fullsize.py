"""Fixture G15 — the CPU oracle's eps of ONE forward at the full sizes of BASELINE.json's configurations.

    python tests/golden/make_golden_fullsize.py [--threads 8] [--only name,...]   ->  tests/golden/fullsize_eps.npz

`tests/test_gpu_zz_fullsize.py` used to recompute these forwards with the CPU oracle INSIDE the GPU suite (10-20 s each on the
GPU box's host cores, ~170 s in all: the suite ran 64 s from the driver's kill).  They are computed here, once, in the build
container (CPU, fp32, the oracle only: `oracle.unet_ref.unet_forward` with the controller / injection hooks of
`oracle.p2p_ref`, `oracle.pnp_ref`, `oracle.masactrl_ref`; no product kernel, no GPU, nothing of /root/reference), from inputs
the tests regenerate from the same seeds:

    sd15_refine_step0 / _step25   SD1.5 shapes, 64x64 latents, UNet batch 4, AttentionRefine (`/root/reference/p2p/model/attention_base.py:113-136`)
    sd15_masactrl_step6           SD1.5 shapes, 64x64, batch 4, mutual self-attention active (step >= 4, layer >= 10:
                                  `/root/reference/masactrl/model/attention_control.py:37-68`)
    sd15_1024_b1                  SD1.5 shapes, 128x128 latents (N = 16384 self-attention), batch 1
    sd21_pnp_b4                   SD2.1 shapes, 96x96 latents, batch 4, both Plug-and-Play injections active
                                  (`/root/reference/pnp/model/register.py:45-52,161-166`)
    sdxl_b1                       SDXL shapes, 128x128 latents, batch 1, text-time additional embedding

Stored per case: eps (fp32) and a probe of the inputs (first 8 values of x and ctx) so that a test whose regenerated inputs
differ fails on the probe, not on the comparison.  Also stored: torch version and thread count of the run.  Summation order on
the CPU depends on the thread count, so two runs differ in the last bits (measured on G13: 1.7e-6 of max |latent| over 50 steps);
the tests' bounds (1e-4 for the fp32-storage modes, 5e-3 for fp16 storage) are far above that.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import ief_amd  # noqa: E402,F401
from ief_amd import config  # noqa: E402
from ief_amd import weights as _weights  # noqa: E402
from ief_amd.tokenizer import WordPieceTokenizer  # noqa: E402
from ief_amd.p2p.model.attention_control import AttentionRefine  # noqa: E402
from oracle import masactrl_ref, p2p_ref, pnp_ref, unet_ref  # noqa: E402

PROMPTS = ["a photo of a house on a mountain", "a photo of a house on a mountain at fall"]
OUT = os.path.join(ROOT, "tests", "golden", "fullsize_eps.npz")


def inputs(cfg, B, seed=0, hw=None):
    """== tests/test_gpu_zz_fullsize.py::_inputs"""
    g = torch.Generator().manual_seed(seed)
    hw = hw or cfg.sample_size
    x = torch.randn(B, 4, hw, hw, generator=g)
    ctx = torch.randn(B, 77, cfg.cross_attention_dim, generator=g)
    return x, ctx


def p2p_batch(cfg, seed):
    x1, ctx = inputs(cfg, 4, seed=seed)
    x = torch.cat([x1[:1], 0.8 * x1[:1] + 0.6 * x1[1:2]] * 2)
    return x, ctx


def masa_batch(cfg, seed):
    x1, ctx = inputs(cfg, 4, seed=seed)
    x = torch.cat([x1[:1], 0.6 * x1[:1] + 0.8 * x1[1:2]] * 2)
    return x, ctx


def case_sd15_refine(step):
    cfg = config.SD15
    sd = _weights.synthetic_state_dict(cfg, 0)
    x, ctx = p2p_batch(cfg, seed=3)
    t = int(p2p_ref.DDIMRef(50).timesteps[step])
    c = AttentionRefine(PROMPTS, WordPieceTokenizer(cfg.text_max_length), 50, 0.8, 0.4, device=torch.device("cpu"))
    rc = p2p_ref.P2PControlRef(mode="refine", num_prompts=2, cross_alpha=c.cross_replace_alpha.float(),
                               num_self_replace=c.num_self_replace, mapper=c.mapper, alphas=c.alphas.float())
    rc.num_att_layers = unet_ref.count_attention_layers(cfg)
    rc.cur_step = step
    return unet_ref.unet_forward(sd, cfg, x, torch.tensor(t), ctx, hook=rc), x, ctx


def case_sd15_masactrl():
    cfg = config.SD15
    sd = _weights.synthetic_state_dict(cfg, 0)
    x, ctx = masa_batch(cfg, seed=9)
    r = masactrl_ref.MasaCtrlRef(step_idx=list(range(4, 50)), layer_idx=list(range(10, 16)),
                                 num_att_layers=unet_ref.count_attention_layers(cfg), cur_step=6)
    return unet_ref.unet_forward(sd, cfg, x, torch.tensor(501), ctx, qkv_hook=r), x, ctx


def case_sd15_1024():
    cfg = config.SD15
    sd = _weights.synthetic_state_dict(cfg, 0)
    x, ctx = inputs(cfg, 1, seed=17, hw=128)
    ctx = ctx * 0.1
    return unet_ref.unet_forward(sd, cfg, x, torch.tensor(481), ctx), x, ctx


def case_sd21_pnp():
    cfg = config.SD21
    sd = _weights.synthetic_state_dict(cfg, 0)
    x, ctx = inputs(cfg, 4, seed=11)
    ctx = ctx * 0.1
    t = int(p2p_ref.DDIMRef(50).timesteps[0])
    return pnp_ref.pnp_forward(sd, cfg, x, t, ctx, True, True), x, ctx


def case_sdxl():
    cfg = config.SDXL
    sd = _weights.synthetic_state_dict(cfg, 0)
    x, ctx = inputs(cfg, 1, seed=13)
    ctx = ctx * 0.1
    g = torch.Generator().manual_seed(14)
    added = {"text_embeds": torch.randn(1, cfg.pooled_text_dim, generator=g) * 0.5,
             "time_ids": torch.tensor([[1024.0, 1024.0, 0.0, 0.0, 1024.0, 1024.0]])}
    return unet_ref.unet_forward(sd, cfg, x, torch.tensor(481), ctx, added_cond_kwargs=added), x, ctx


CASES = {
    "sd15_refine_step0": lambda: case_sd15_refine(0),
    "sd15_refine_step25": lambda: case_sd15_refine(25),
    "sd15_masactrl_step6": case_sd15_masactrl,
    "sd15_1024_b1": case_sd15_1024,
    "sd21_pnp_b4": case_sd21_pnp,
    "sdxl_b1": case_sdxl,
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default=OUT)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    out = dict(np.load(args.out)) if os.path.exists(args.out) else {}
    names = [n for n in args.only.split(",") if n] or list(CASES)
    for name in names:
        t0 = time.time()
        with torch.no_grad():
            eps, x, ctx = CASES[name]()
        out[name] = eps.float().numpy()
        out[name + "__probe"] = np.concatenate([x.flatten()[:8].numpy(), ctx.flatten()[:8].numpy()])
        out["meta_threads"] = np.int64(args.threads)
        out["meta_torch"] = np.array(torch.__version__)
        np.savez_compressed(args.out, **out)
        print(f"{name}: eps {tuple(eps.shape)} max |eps| {eps.abs().max():.4f}  {time.time() - t0:.0f} s", flush=True)
    print("wrote", args.out, sorted(out))


if __name__ == "__main__":
    main()
