"""Golden vectors for the SDXL-family inversion loops, made by IMPORTING the reference's own modules
`/root/reference/pix2pix-zero/inversion/{ddim,nti}.py` (torch / numpy / PIL / tqdm only: they import cleanly here) and
running `ddim_inversion_xl.ddim_inversion_loop` and `NTI_XL.null_optimization` on a toy differentiable UNet whose output
depends on `added_cond_kwargs`, so the routing of the conditional / unconditional kwargs, the lr 5e-2 schedule, the
restart of the embedding at every timestep and the early stop are all pinned.  Run in the build container only:

    python tests/golden/make_golden_xl.py        ->  tests/golden/nti_xl.npz   (G9, G10)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/pix2pix-zero")
from inversion import ddim as ref_ddim  # noqa: E402  (reference)
from inversion import nti as ref_nti  # noqa: E402


class _StubSched:
    """Scheduler constants of SURVEY.md §8a row S; only what ddim.py / nti.py read."""

    def __init__(self, n):
        betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.config = types.SimpleNamespace(num_train_timesteps=1000)
        self.num_inference_steps = n
        self.timesteps = torch.from_numpy((np.arange(0, n) * (1000 // n)).round()[::-1].copy().astype(np.int64) + 1)

    def step(self, eps, t, x):
        t = int(t)
        prev = t - 1000 // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        return types.SimpleNamespace(prev_sample=a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * eps)


class _Out(dict):
    def __init__(self, s):
        super().__init__(sample=s)
        self.sample = s


class _ToyUNetXL:
    """eps = tanh(x w1 + mean_tokens(ctx w2) + text_embeds w3 + 1e-4 sum(time_ids) + 1e-3 t): every input of the SDXL
    UNet call matters, batch row by batch row"""
    device = torch.device("cpu")

    def __init__(self):
        g = torch.Generator().manual_seed(17)
        self.w1 = torch.randn(4, 4, generator=g) * 0.5
        self.w2 = torch.randn(16, 4, generator=g) * 0.5
        self.w3 = torch.randn(8, 4, generator=g) * 0.5

    def __call__(self, x, t, encoder_hidden_states=None, added_cond_kwargs=None, **kw):
        c = (encoder_hidden_states @ self.w2).mean(1) + added_cond_kwargs["text_embeds"] @ self.w3 \
            + 1e-4 * added_cond_kwargs["time_ids"].sum(-1, keepdim=True)
        h = torch.einsum("bchw,cd->bdhw", x, self.w1) + c[:, :, None, None] + float(t) * 1e-3
        return _Out(torch.tanh(h))


def main():
    out = {}
    g = torch.Generator().manual_seed(23)
    sched = _StubSched(5)
    toy = _ToyUNetXL()
    emb = torch.randn(1, 6, 16, generator=g) * 0.3
    neg = torch.zeros(1, 6, 16)
    pooled = torch.randn(1, 8, generator=g) * 0.3
    neg_pooled = torch.zeros(1, 8)
    model = types.SimpleNamespace(
        scheduler=sched, unet=toy, _execution_device=torch.device("cpu"),
        encode_prompt=lambda **kw: (emb, neg, pooled, neg_pooled),
        _get_add_time_ids=lambda o, c, t, dtype=torch.float32: torch.tensor([list(o + c + t)], dtype=dtype))
    x0 = torch.randn(1, 4, 8, 8, generator=g)
    inv = ref_nti.NTI_XL()
    # G10: ddim_inversion_xl.ddim_inversion_loop (height = width = 64 so the time ids are small numbers)
    lat, context = inv.ddim_inversion_loop(model, x0, ["a prompt"], height=64, width=64)
    # G9: NTI_XL.null_optimization, 10 inner steps, the CLI's epsilon
    lst = inv.null_optimization(model, lat, context, 10, 1e-5, 7.5, height=64, width=64)
    out["w1"], out["w2"], out["w3"] = toy.w1.numpy(), toy.w2.numpy(), toy.w3.numpy()
    out["emb"], out["pooled"] = emb.numpy(), pooled.numpy()
    out["x0"] = x0.numpy()
    out["timesteps"] = sched.timesteps.numpy()
    out["inv_latents"] = np.stack([l.numpy() for l in lat])
    out["nti_uncond"] = np.stack([u.numpy() for u in lst])
    # G11: the P2P folder's copy of NTI_XL (`/root/reference/p2p/inversion/nti.py:47-96`): lr = 0.5 (1 - i / 500)
    import importlib.util
    for mod in ("inversion.ddim", "inversion.nti", "inversion"):
        sys.modules.pop(mod, None)
    sys.path[0] = "/root/reference/p2p"
    from inversion import nti as p2p_nti  # noqa: E402
    assert p2p_nti.__file__.startswith("/root/reference/p2p/")
    lst2 = p2p_nti.NTI_XL().null_optimization(model, lat, context, 10, 1e-5, 7.5, height=64, width=64)
    out["nti_uncond_p2p"] = np.stack([u.numpy() for u in lst2])
    np.savez_compressed(os.path.join(HERE, "nti_xl.npz"), **out)
    print("nti_xl.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
