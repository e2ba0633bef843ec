"""Generate golden vectors by IMPORTING the reference's own Python modules.

Run in the build container only (`/root/reference` does not travel to the GPU box):

    python tests/golden/make_golden.py

Writes small `.npz` fixtures next to this file.  Each fixture holds inputs and expected
outputs only — no reference source.  The reference modules imported are
`/root/reference/p2p/model/{seq_aligner,ptp_utils,attention_base,attention_control}.py` and
`/root/reference/p2p/inversion/{ddim,nti}.py` (all import cleanly here; `model/sd_utils.py`
needs diffusers and is NOT imported — SURVEY.md §8c).

Groups:
  G1 seq_aligner mappers            G2 step x word gate tables
  G3 controller sweeps              G4 AttentionStore     G5 LocalBlend
  G6 ddim_reverse                   G8 NTI loop on a toy differentiable UNet
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/p2p"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import ief_amd  # noqa: E402
from ief_amd.tokenizer import WordPieceTokenizer  # noqa: E402

from model import seq_aligner as ref_aligner  # noqa: E402  (reference)
from model import ptp_utils as ref_ptp  # noqa: E402
from model import attention_base as ref_base  # noqa: E402
from model import attention_control as ref_ctrl  # noqa: E402
from inversion import ddim as ref_ddim  # noqa: E402
from inversion import nti as ref_nti  # noqa: E402

CPU = torch.device("cpu")

sys.path.insert(0, HERE)
from make_golden_inputs import PROMPT_PAIRS, softmax_maps as _softmax_maps  # noqa: E402


def g1_g2(tok):
    out = {}
    for k, (a, b) in enumerate(PROMPT_PAIRS):
        mapper, alphas = ref_aligner.get_refinement_mapper([a, b], tok)
        out[f"refine_mapper_{k}"] = mapper.numpy()
        out[f"refine_alphas_{k}"] = alphas.numpy()
        out[f"ids_a_{k}"] = np.array(tok.encode(a))
        out[f"ids_b_{k}"] = np.array(tok.encode(b))
        if len(a.split(" ")) == len(b.split(" ")):
            out[f"replace_mapper_{k}"] = ref_aligner.get_replacement_mapper([a, b], tok).numpy()
    a, b = PROMPT_PAIRS[0]
    out["alpha_f08"] = ref_ptp.get_time_words_attention_alpha([a, b], 50, 0.8, tok).numpy()
    out["alpha_t0209"] = ref_ptp.get_time_words_attention_alpha([a, b], 50, (0.2, 0.9), tok).numpy()
    out["alpha_dict"] = ref_ptp.get_time_words_attention_alpha(
        [a, b], 50, {"default_": 1.0, "fall": (0.0, 0.4), "mountain": (0.3, 0.7)}, tok).numpy()
    out["alpha_3prompts"] = ref_ptp.get_time_words_attention_alpha(
        [a, b, "a photo of a castle on a mountain"], 50, 0.6, tok).numpy()
    out["word_inds_mountain"] = ref_ptp.get_word_inds(b, "mountain", tok)
    out["word_inds_5"] = ref_ptp.get_word_inds(b, 5, tok)
    out["word_inds_long"] = ref_ptp.get_word_inds(PROMPT_PAIRS[11][0], "characteristics", tok)
    out["equalizer"] = ref_aligner.get_equalizer(tok, b, ("fall", "mountain"), (4.0,)).numpy()
    np.savez_compressed(os.path.join(HERE, "p2p_host.npz"), **out)
    print("p2p_host.npz", len(out), "arrays")


def g3_g4_g5(tok):
    out = {}
    heads = 2
    a, b = PROMPT_PAIRS[0]
    a2, b2 = PROMPT_PAIRS[2]
    ctrls = {
        "refine": lambda: ref_ctrl.AttentionRefine([a, b], tok, 50, 0.8, 0.4, device=CPU),
        "replace": lambda: ref_ctrl.AttentionReplace([a2, b2], tok, 50, 0.8, 0.4, device=CPU),
        "reweight": lambda: ref_ctrl.AttentionReweight(
            [a, b], tok, 50, 0.8, 0.4,
            ref_aligner.get_equalizer(tok, b, ("fall",), (3.0,)), device=CPU),
        "refine3": lambda: ref_ctrl.AttentionRefine([a, b, "a photo of a tree house on a mountain"], tok, 50,
                                                   (0.1, 0.7), (0.1, 0.5), device=CPU),
    }
    for name, make in ctrls.items():
        nprompt = 3 if name == "refine3" else 2
        bh = 2 * nprompt * heads
        for step in (0, 4, 19, 20, 39, 40, 49):
            c = make()
            c.num_att_layers = 4
            c.cur_step = step
            # layer sweep: cross N=64, self N=64 (<=256: replaced in window), self N=320 (>256), cross N=16
            shapes = [(True, 64, 77), (False, 64, 64), (False, 320, 320), (True, 16, 77)]
            for li, (is_cross, n, l) in enumerate(shapes):
                x = _softmax_maps(1000 + li, bh, n, l)
                y = c(x.clone(), is_cross, "down")
                key = f"{name}_s{step}_l{li}"
                if not is_cross and n > 256:
                    out[key + "_unchanged"] = np.array(bool(torch.equal(x, y)))
                else:
                    out[key] = y[bh // 2:].numpy().astype(np.float32)  # cond half; uncond half checked below
                out[key + "_uncond_same"] = np.array(bool(torch.equal(x[: bh // 2], y[: bh // 2])))
            out[f"{name}_s{step}_after"] = np.array([c.cur_step, c.cur_att_layer])
    # G4 AttentionStore over 3 steps of 3 layers
    st = ref_base.AttentionStore(False)
    st.num_att_layers = 3
    for step in range(3):
        for li, (is_cross, n, l) in enumerate([(True, 64, 77), (False, 64, 64), (False, 1600, 8)]):
            st(_softmax_maps(2000 + 10 * step + li, 4, n, l), is_cross, ["down", "mid", "up"][li])
    avg = st.get_average_attention()
    for key, maps in avg.items():
        out[f"store_{key}_n"] = np.array(len(maps))
        for i, m in enumerate(maps):
            out[f"store_{key}_{i}"] = m.numpy()
    # G5 LocalBlend on a synthetic store (5 maps of 16x16, 2 prompts x 2 heads)
    lb = ref_ptp.LocalBlend(tok, [a, b], [["house"], ["fall"]], device=CPU)
    store = {"down_cross": [_softmax_maps(3000 + i, 4, 256, 77) for i in range(4)],
             "up_cross": [_softmax_maps(3100 + i, 4, 256, 77) for i in range(3)]}
    g = torch.Generator().manual_seed(7)
    x_t = torch.randn(2, 4, 64, 64, generator=g)
    out["localblend_out"] = lb(x_t, store).numpy()
    out["localblend_alpha_layers"] = lb.alpha_layers.numpy()
    np.savez_compressed(os.path.join(HERE, "p2p_ctrl.npz"), **out)
    print("p2p_ctrl.npz", len(out), "arrays")


def g3_chain(tok):
    """G3b: AttentionReweight CHAINED on an AttentionRefine (`attention_control.py:42-46`, `prev_controller`): the
    refinement edit of the source maps, then the per-word equalizer — the combination the P2P notebook uses"""
    out = {}
    heads = 2
    a, b = PROMPT_PAIRS[0]
    eq = ref_aligner.get_equalizer(tok, b, ("fall",), (3.0,))

    def make():
        prev = ref_ctrl.AttentionRefine([a, b], tok, 50, 0.8, 0.4, device=CPU)
        return ref_ctrl.AttentionReweight([a, b], tok, 50, 0.8, 0.4, eq, controller=prev, device=CPU)
    bh = 2 * 2 * heads
    for step in (0, 39, 40):
        c = make()
        c.num_att_layers = 2
        c.cur_step = step
        for li, (is_cross, n, l) in enumerate([(True, 64, 77), (True, 16, 77)]):
            x = _softmax_maps(1500 + li, bh, n, l)
            y = c(x.clone(), is_cross, "down")
            out[f"chain_s{step}_l{li}"] = y[bh // 2:].numpy().astype(np.float32)
            out[f"chain_s{step}_l{li}_uncond_same"] = np.array(bool(torch.equal(x[: bh // 2], y[: bh // 2])))
        out[f"chain_s{step}_after"] = np.array([c.cur_step, c.cur_att_layer])
    out["chain_equalizer"] = eq.numpy()
    np.savez_compressed(os.path.join(HERE, "p2p_ctrl_chain.npz"), **out)
    print("p2p_ctrl_chain.npz", len(out), "arrays")


class _StubSched:
    """Scheduler constants of SURVEY.md §8a row S; only the attributes ddim.py/nti.py read."""

    def __init__(self, n=50):
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


class _ToyUNet:
    """Differentiable stand-in: eps = tanh(x*w1 + mean_tokens(ctx @ w2) + t*1e-3)."""

    def __init__(self):
        g = torch.Generator().manual_seed(11)
        self.w1 = torch.randn(4, 4, generator=g) * 0.5
        self.w2 = torch.randn(16, 4, generator=g) * 0.5

    def __call__(self, x, t, encoder_hidden_states=None, **kw):
        ctx = encoder_hidden_states
        c = (ctx @ self.w2).mean(1)  # [B,4]
        h = torch.einsum("bchw,cd->bdhw", x, self.w1) + c[:, :, None, None] + float(t) * 1e-3
        s = torch.tanh(h)
        return _Out(s)


class _Out(dict):
    def __init__(self, s):
        super().__init__(sample=s)
        self.sample = s


def g6_g8():
    out = {}
    sched = _StubSched()
    model = types.SimpleNamespace(scheduler=sched)
    inv = ref_ddim.ddim_inversion()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 4, 8, 8, generator=g)
    e = torch.randn(1, 4, 8, 8, generator=g)
    out["ddim_x"], out["ddim_eps"] = x.numpy(), e.numpy()
    out["ddim_timesteps"] = sched.timesteps.numpy()
    out["alphas_cumprod"] = sched.alphas_cumprod.numpy()
    for t in sched.timesteps:
        out[f"ddim_reverse_t{int(t)}"] = inv.ddim_reverse(model, e, t, x).numpy()
    # G8: NTI on the toy UNet over a 5-step schedule
    sched5 = _StubSched(5)
    toy = _ToyUNet()
    model = types.SimpleNamespace(scheduler=sched5, unet=toy)
    ctx = torch.randn(2, 6, 16, generator=g) * 0.3
    lat = [torch.randn(1, 4, 8, 8, generator=g)]
    for i in range(5):
        t = sched5.timesteps[len(sched5.timesteps) - i - 1]
        lat.append(inv.ddim_reverse(model, toy(lat[-1], t, ctx[1:]).sample, t, lat[-1]))
    lst = ref_nti.NTI().null_optimization(model, lat, ctx, 10, 1e-5, 7.5)
    out["nti_w1"], out["nti_w2"] = toy.w1.numpy(), toy.w2.numpy()
    out["nti_ctx"] = ctx.numpy()
    out["nti_latents"] = np.stack([l.numpy() for l in lat])
    out["nti_uncond"] = np.stack([u.numpy() for u in lst])
    out["nti_timesteps"] = sched5.timesteps.numpy()
    np.savez_compressed(os.path.join(HERE, "ddim_nti.npz"), **out)
    print("ddim_nti.npz", len(out), "arrays")




def g7_masactrl():
    """G7: MasaCtrl `AttentionBase` (importable; `attention_control.py` needs torchvision and is not)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_masa_base", "/root/reference/masactrl/model/attention_base.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    g = torch.Generator().manual_seed(21)
    heads, b, n, d = 4, 2, 16, 8
    q, k, v = (torch.randn(b * heads, n, d, generator=g) for _ in range(3))
    sim = torch.bmm(q, k.transpose(1, 2)) * d ** -0.5
    attn = sim.softmax(-1)
    e = mod.AttentionBase()
    e.num_att_layers = 3
    outs = [e(q, k, v, sim, attn, False, "down", heads, scale=d ** -0.5) for _ in range(4)]
    out["q"], out["k"], out["v"] = q.numpy(), k.numpy(), v.numpy()
    out["base_out"] = outs[0].numpy()
    out["counters"] = np.array([e.cur_step, e.cur_att_layer])
    np.savez_compressed(os.path.join(HERE, "masactrl.npz"), **out)
    print("masactrl.npz", len(out), "arrays")


if __name__ == "__main__":
    torch.set_num_threads(1)
    tok = WordPieceTokenizer()
    g1_g2(tok)
    g3_g4_g5(tok)
    g3_chain(tok)
    g6_g8()
    g7_masactrl()
