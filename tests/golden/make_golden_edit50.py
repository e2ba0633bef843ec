"""Fixture G13 — the reference's UNIT OF WORK at full size: one 50-step Prompt-to-Prompt edit, SD1.5 shapes, 512x512.

    python tests/golden/make_golden_edit50.py [--steps 50] [--threads N]      ->  tests/golden/sd15_edit50.npz

What is run (CPU, fp32, the oracle only — no product kernel, no GPU): `oracle.p2p_ref.edit_loop`, the restatement of
`/root/reference/p2p/model/sd_utils.py:24-79` (`text2image_ldm_stable` + `diffusion_step`), with

    weights      `weights.synthetic_state_dict(config.SD15, seed 0)`   (no SD checkpoint exists offline)
    prompts      the CLI defaults, `/root/reference/p2p/edit_syn.py:20-21`
    controller   AttentionRefine, cross_replace_steps 0.8, self_replace_steps 0.4 (`edit_syn.py:16-17,103-105`)
    x_T          CPU generator, seed 8888 (`edit_syn.py:19`), shared by both prompts (`sd_utils.py:13-21`)
    guidance     7.5, 50 DDIM steps, timesteps 981 ... 1
    context      the seeded stand-in text encoder of `pipeline.py` on the CPU (same module the product uploads)

and `oracle.vae_ref.decode` (SD VAE shapes, seeded weights) on the final latents.

What is stored (arrays only, ~1.3 MB): x_T, the context, the latents after steps 10 / 25 / 50 (both prompts, fp32), and of
the decoded images (fp32, before the clamp): the centre 64x64 crop, an 8x8 average-pooled copy of the whole image, and the
uint8 crop the reference's `latent2image` would write.  The GPU test (`tests/test_gpu_zz_fullsize.py`) rebuilds the same
seeded weights on the GPU box and compares the HIP path's trajectory with these arrays; nothing of `/root/reference` is
needed at test time.

Cost: ~50-90 s per step on 8 cores (materialised 2 GiB self-attention maps per 64x64 layer, as the reference does).
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
from ief_amd.pipeline import SyntheticTextEncoder  # noqa: E402
from ief_amd.tokenizer import WordPieceTokenizer  # noqa: E402
from ief_amd.p2p.model.attention_control import AttentionRefine  # noqa: E402
from ief_amd.vae import SD_VAE, synthetic_vae_state_dict  # noqa: E402
from oracle import p2p_ref, vae_ref  # noqa: E402

PROMPTS = ["a photo of a house on a mountain", "a photo of a house on a mountain at fall"]
KEEP = (10, 25, 50)


def encode(tok, enc, prompts):
    """`sd_utils.py:42-55`"""
    ti = tok(prompts, padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt")
    cond = enc(ti.input_ids)[0]
    ui = tok([""] * len(prompts), padding="max_length", max_length=ti.input_ids.shape[-1], return_tensors="pt")
    return enc(ui.input_ids)[0], cond


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "sd15_edit50.npz"))
    args = ap.parse_args()
    if args.threads:
        torch.set_num_threads(args.threads)
    cfg = config.SD15
    n = args.steps
    sd = _weights.synthetic_state_dict(cfg, 0)
    tok = WordPieceTokenizer(cfg.text_max_length)
    enc = SyntheticTextEncoder(cfg.cross_attention_dim)
    with torch.no_grad():
        u, c = encode(tok, enc, PROMPTS)
    context = torch.cat([u, c]).float()
    x_T = torch.randn(1, 4, cfg.sample_size, cfg.sample_size, generator=torch.Generator().manual_seed(8888))
    ctl = AttentionRefine(PROMPTS, tok, n, 0.8, 0.4, device=torch.device("cpu"))
    rc = p2p_ref.P2PControlRef(mode="refine", num_prompts=2, cross_alpha=ctl.cross_replace_alpha.float(),
                               num_self_replace=ctl.num_self_replace, mapper=ctl.mapper, alphas=ctl.alphas.float())

    class Trace(list):
        def append(self, lat):
            super().append(lat)
            print(f"step {len(self):2d}/{n}  {time.time() - t0:7.0f} s  max|lat| {lat.abs().max():.4f}", flush=True)

    trace = Trace()
    t0 = time.time()
    lat = p2p_ref.edit_loop(sd, cfg, context, x_T, rc, p2p_ref.DDIMRef(n), 7.5, trace=trace)
    assert rc.cur_step == n and rc.cur_att_layer == 0
    keep = {f"lat_{k}": trace[k - 1].numpy() for k in KEEP if k <= n}
    out = dict(x_T=x_T.numpy(), context=context.numpy(), lat_final=lat.numpy(), steps=np.int64(n), **keep)
    if n == 50:
        vsd = synthetic_vae_state_dict(SD_VAE, 2)
        with torch.no_grad():
            dec = vae_ref.decode(vsd, SD_VAE, lat / SD_VAE.scaling_factor)          # [2, 3, 512, 512], before the clamp
        H = dec.shape[-1]
        a = H // 2 - 32
        out["img_crop"] = dec[:, :, a:a + 64, a:a + 64].numpy()
        out["img_pool8"] = torch.nn.functional.avg_pool2d(dec, 8).numpy()
        out["img_u8_crop"] = p2p_ref.latent_to_uint8(dec)[:, a:a + 64, a:a + 64]
        out["crop_origin"] = np.int64(a)
    np.savez_compressed(args.out, **out)
    print("wrote", args.out, {k: getattr(v, "shape", v) for k, v in out.items()}, f"{time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
