"""CPU tests: host-side mirror AND oracle restatement against golden vectors produced by the
reference's own modules (`tests/golden/make_golden.py`).  Integer outputs bit-exact; float outputs
exact where the reference arithmetic is reproduced operation for operation, else 1e-6."""
import os

import numpy as np
import pytest
import torch

from ief_amd.tokenizer import WordPieceTokenizer
from ief_amd.p2p.model import attention_base, attention_control, ptp_utils, seq_aligner
from ief_amd.scheduler import DDIMScheduler
from oracle import p2p_ref

import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from make_golden_inputs import PROMPT_PAIRS, softmax_maps  # noqa: E402

CPU = torch.device("cpu")


@pytest.fixture(scope="module")
def tok():
    return WordPieceTokenizer()


@pytest.fixture(scope="module")
def host(golden_dir):
    return np.load(os.path.join(golden_dir, "p2p_host.npz"))


@pytest.fixture(scope="module")
def ctrl(golden_dir):
    return np.load(os.path.join(golden_dir, "p2p_ctrl.npz"))


@pytest.fixture(scope="module")
def ddim(golden_dir):
    return np.load(os.path.join(golden_dir, "ddim_nti.npz"))


# ------------------------------------------------------------------------------------ G1 / G2
def test_tokenizer_ids_stable(tok, host):
    for k, (a, b) in enumerate(PROMPT_PAIRS):
        assert np.array_equal(np.array(tok.encode(a)), host[f"ids_a_{k}"])
        assert np.array_equal(np.array(tok.encode(b)), host[f"ids_b_{k}"])


def test_refinement_mapper_bit_exact(tok, host):
    for k, (a, b) in enumerate(PROMPT_PAIRS):
        mapper, alphas = seq_aligner.get_refinement_mapper([a, b], tok)
        assert mapper.dtype == torch.int64
        assert np.array_equal(mapper.numpy(), host[f"refine_mapper_{k}"]), k
        assert np.array_equal(alphas.numpy(), host[f"refine_alphas_{k}"]), k


def test_replacement_mapper_exact(tok, host):
    n = 0
    for k, (a, b) in enumerate(PROMPT_PAIRS):
        if len(a.split(" ")) == len(b.split(" ")):
            m = seq_aligner.get_replacement_mapper([a, b], tok)
            assert np.array_equal(m.numpy(), host[f"replace_mapper_{k}"]), k
            n += 1
        else:
            with pytest.raises(ValueError):
                seq_aligner.get_replacement_mapper([a, b], tok)
    assert n >= 4


def test_alpha_tables_and_word_inds(tok, host):
    a, b = PROMPT_PAIRS[0]
    f = ptp_utils.get_time_words_attention_alpha
    assert np.array_equal(f([a, b], 50, 0.8, tok).numpy(), host["alpha_f08"])
    assert np.array_equal(f([a, b], 50, (0.2, 0.9), tok).numpy(), host["alpha_t0209"])
    assert np.array_equal(f([a, b], 50, {"default_": 1.0, "fall": (0.0, 0.4), "mountain": (0.3, 0.7)}, tok).numpy(),
                          host["alpha_dict"])
    assert np.array_equal(f([a, b, "a photo of a castle on a mountain"], 50, 0.6, tok).numpy(), host["alpha_3prompts"])
    assert np.array_equal(ptp_utils.get_word_inds(b, "mountain", tok), host["word_inds_mountain"])
    assert np.array_equal(ptp_utils.get_word_inds(b, 5, tok), host["word_inds_5"])
    assert np.array_equal(ptp_utils.get_word_inds(PROMPT_PAIRS[11][0], "characteristics", tok), host["word_inds_long"])
    assert np.array_equal(seq_aligner.get_equalizer(tok, b, ("fall", "mountain"), (4.0,)).numpy(), host["equalizer"])


# ------------------------------------------------------------------------------------ G3
def _controllers(tok):
    a, b = PROMPT_PAIRS[0]
    a2, b2 = PROMPT_PAIRS[2]
    third = "a photo of a tree house on a mountain"
    return {
        "refine": (2, lambda: attention_control.AttentionRefine([a, b], tok, 50, 0.8, 0.4, device=CPU)),
        "replace": (2, lambda: attention_control.AttentionReplace([a2, b2], tok, 50, 0.8, 0.4, device=CPU)),
        "reweight": (2, lambda: attention_control.AttentionReweight(
            [a, b], tok, 50, 0.8, 0.4, seq_aligner.get_equalizer(tok, b, ("fall",), (3.0,)), device=CPU)),
        "refine3": (3, lambda: attention_control.AttentionRefine([a, b, third], tok, 50, (0.1, 0.7), (0.1, 0.5), device=CPU)),
    }


SHAPES = [(True, 64, 77), (False, 64, 64), (False, 320, 320), (True, 16, 77)]
STEPS = (0, 4, 19, 20, 39, 40, 49)


def _as_oracle(c, nprompt):
    name = type(c).__name__
    return p2p_ref.P2PControlRef(
        mode={"AttentionRefine": "refine", "AttentionReplace": "replace", "AttentionReweight": "reweight"}[name],
        num_prompts=nprompt, cross_alpha=c.cross_replace_alpha, num_self_replace=c.num_self_replace,
        mapper=getattr(c, "mapper", None), alphas=getattr(c, "alphas", None), equalizer=getattr(c, "equalizer", None))


@pytest.mark.parametrize("which", ["mirror", "oracle"])
@pytest.mark.parametrize("name", ["refine", "replace", "reweight", "refine3"])
def test_controller_sweeps_match_reference(which, name, tok, ctrl):
    nprompt, make = _controllers(tok)[name]
    bh = 2 * nprompt * 2
    for step in STEPS:
        c = make()
        if which == "oracle":
            c = _as_oracle(c, nprompt)
        c.num_att_layers = 4
        c.cur_step = step
        for li, (is_cross, n, l) in enumerate(SHAPES):
            x = softmax_maps(1000 + li, bh, n, l)
            y = c(x.clone(), is_cross, "down")
            key = f"{name}_s{step}_l{li}"
            assert torch.equal(x[: bh // 2], y[: bh // 2]) == bool(ctrl[key + "_uncond_same"])
            if not is_cross and n > 256:
                assert torch.equal(x, y) == bool(ctrl[key + "_unchanged"])
            else:
                assert np.array_equal(y[bh // 2:].numpy(), ctrl[key]), key
        assert [c.cur_step, c.cur_att_layer] == list(ctrl[f"{name}_s{step}_after"])


def _oracle_of(c, nprompt):
    o = _as_oracle(c, nprompt)
    if getattr(c, "prev_controller", None) is not None:
        o.prev = _as_oracle(c.prev_controller, nprompt)
    return o


@pytest.mark.parametrize("which", ["mirror", "oracle", "lowered"])
def test_chained_reweight_matches_reference(which, tok, golden_dir):
    """G3b: AttentionReweight(controller=AttentionRefine) — the host mirror, the oracle's `prev` chain, and the device
    tables the fused cross-attention kernel reads (`register._edit_tables`: P' = c1 (P_src M) + c2 P_tgt) all reproduce the
    reference's maps"""
    from ief_amd.p2p.model import register
    chain = np.load(os.path.join(golden_dir, "p2p_ctrl_chain.npz"))
    a, b = PROMPT_PAIRS[0]
    eq = seq_aligner.get_equalizer(tok, b, ("fall",), (3.0,))
    assert np.array_equal(eq.numpy(), chain["chain_equalizer"])

    def make():
        prev = attention_control.AttentionRefine([a, b], tok, 50, 0.8, 0.4, device=CPU)
        return attention_control.AttentionReweight([a, b], tok, 50, 0.8, 0.4, eq, controller=prev, device=CPU)
    bh = 8
    for step in (0, 39, 40):
        c = make()
        if which == "lowered":
            plan = register.lower_controller(c, CPU)
            assert plan is not None and plan.kind == "p2p"
            c1, c2 = plan.coef_table[step, 0, 0, :77], plan.coef_table[step, 0, 1, :77]
            M = plan.mt[0, :77, :77].float().t()
        elif which == "oracle":
            c = _oracle_of(c, 2)
        if which != "lowered":
            c.num_att_layers = 2
            c.cur_step = step
        for li, (n, l) in enumerate([(64, 77), (16, 77)]):
            x = softmax_maps(1500 + li, bh, n, l)
            if which == "lowered":
                cond = x[bh // 2:].reshape(2, 2, n, l)
                y = c1 * (cond[0] @ M) + c2 * cond[1]
                assert np.allclose(y.numpy(), chain[f"chain_s{step}_l{li}"][2:], rtol=0, atol=1e-6), (step, li)
                continue
            y = c(x.clone(), True, "down")
            assert torch.equal(x[: bh // 2], y[: bh // 2]) == bool(chain[f"chain_s{step}_l{li}_uncond_same"])
            assert np.array_equal(y[bh // 2:].numpy(), chain[f"chain_s{step}_l{li}"]), (step, li)
        if which != "lowered":
            assert [c.cur_step, c.cur_att_layer] == list(chain[f"chain_s{step}_after"])


def test_controller_edits_in_place_and_aliases(tok):
    _, make = _controllers(tok)["refine"]
    c = make()
    c.num_att_layers = 10
    x = softmax_maps(1, 8, 64, 77)
    y = c(x, True, "up")
    assert y.data_ptr() == x.data_ptr()          # returned tensor aliases the argument (attention_base.py:22)
    assert not torch.equal(x, softmax_maps(1, 8, 64, 77))  # ... and was edited in place


def test_attention_store_matches_reference(ctrl):
    st = attention_base.AttentionStore(False)
    st.num_att_layers = 3
    for step in range(3):
        for li, (is_cross, n, l) in enumerate([(True, 64, 77), (False, 64, 64), (False, 1600, 8)]):
            st(softmax_maps(2000 + 10 * step + li, 4, n, l), is_cross, ["down", "mid", "up"][li])
    avg = st.get_average_attention()
    for key, maps in avg.items():
        assert len(maps) == int(ctrl[f"store_{key}_n"])
        for i, m in enumerate(maps):
            assert np.array_equal(m.numpy(), ctrl[f"store_{key}_{i}"])
    st.reset()
    assert st.cur_step == 0 and st.attention_store == {}


def test_local_blend_matches_reference(tok, ctrl):
    a, b = PROMPT_PAIRS[0]
    lb = ptp_utils.LocalBlend(tok, [a, b], [["house"], ["fall"]], device=CPU)
    assert np.array_equal(lb.alpha_layers.numpy(), ctrl["localblend_alpha_layers"])
    store = {"down_cross": [softmax_maps(3000 + i, 4, 256, 77) for i in range(4)],
             "up_cross": [softmax_maps(3100 + i, 4, 256, 77) for i in range(3)]}
    x_t = torch.randn(2, 4, 64, 64, generator=torch.Generator().manual_seed(7))
    assert np.allclose(lb(x_t, store).numpy(), ctrl["localblend_out"], atol=1e-6)


# ------------------------------------------------------------------------------------ G6 scheduler
def test_scheduler_constants_and_reverse(ddim):
    s = DDIMScheduler()
    s.set_timesteps(50)
    assert np.array_equal(s.timesteps.numpy(), ddim["ddim_timesteps"])
    assert s.timesteps[0] == 981 and s.timesteps[-1] == 1
    assert np.array_equal(s.alphas_cumprod.numpy(), ddim["alphas_cumprod"])
    ref = p2p_ref.DDIMRef(50)
    assert np.array_equal(ref.timesteps.numpy(), ddim["ddim_timesteps"])
    x, e = torch.from_numpy(ddim["ddim_x"]), torch.from_numpy(ddim["ddim_eps"])
    for t in s.timesteps.tolist():
        want = ddim[f"ddim_reverse_t{t}"]
        assert np.array_equal(ref.reverse(e, t, x).numpy(), want), t        # oracle: operation-for-operation
        a_c, a_n = s.reverse_coeffs(t)                                       # product: same constants (kernel checked on GPU)
        x0 = (x - (1 - a_c) ** 0.5 * e) / a_c ** 0.5
        assert np.allclose((a_n ** 0.5 * x0 + (1 - a_n) ** 0.5 * e).numpy(), want, rtol=0, atol=2e-6), t


def test_scheduler_step_coeffs_edges():
    s = DDIMScheduler()
    s.set_timesteps(50)
    a_t, a_p = s.step_coeffs(1)
    assert a_p == float(s.final_alpha_cumprod) == float(s.alphas_cumprod[0])   # prev = -19 -> final_alpha_cumprod
    a_c, a_n = s.reverse_coeffs(1)
    assert a_c == float(s.final_alpha_cumprod) and a_n == float(s.alphas_cumprod[1])
    with pytest.raises(ValueError):
        s.step(torch.zeros(1), 1, torch.zeros(1), eta=0.5)


# ------------------------------------------------------------------------------------ G8 NTI loop (oracle)
def test_oracle_nti_loop_matches_reference(ddim):
    import types
    w1, w2 = torch.from_numpy(ddim["nti_w1"]), torch.from_numpy(ddim["nti_w2"])

    def toy(sd, cfg, x, t, ctx, **kw):
        c = (ctx @ w2).mean(1)
        return torch.tanh(torch.einsum("bchw,cd->bdhw", x, w1) + c[:, :, None, None] + float(t) * 1e-3)

    sched = p2p_ref.DDIMRef(5)
    assert np.array_equal(sched.timesteps.numpy(), ddim["nti_timesteps"])
    lat = [torch.from_numpy(a) for a in ddim["nti_latents"]]
    ctx = torch.from_numpy(ddim["nti_ctx"])
    saved = p2p_ref.unet_ref.unet_forward
    p2p_ref.unet_ref.unet_forward = toy
    try:
        out = p2p_ref.null_optimization(None, None, lat, ctx, sched, 10, 1e-5, 7.5)
    finally:
        p2p_ref.unet_ref.unet_forward = saved
    got = np.stack([u.numpy() for u in out])
    assert got.shape == ddim["nti_uncond"].shape
    assert np.allclose(got, ddim["nti_uncond"], atol=1e-6)


# ------------------------------------------------------------------------------------ G7 MasaCtrl
def test_masactrl_attention_base_matches_reference(golden_dir):
    from ief_amd.masactrl.model.attention_base import AttentionBase
    z = np.load(os.path.join(golden_dir, "masactrl.npz"))
    q, k, v = (torch.from_numpy(z[n]) for n in "qkv")
    heads, d = 4, 8
    sim = torch.bmm(q, k.transpose(1, 2)) * d ** -0.5
    e = AttentionBase()
    e.num_att_layers = 3
    outs = [e(q, k, v, sim, sim.softmax(-1), False, "down", heads, scale=d ** -0.5) for _ in range(4)]
    assert np.allclose(outs[0].numpy(), z["base_out"], atol=1e-6)
    assert [e.cur_step, e.cur_att_layer] == list(z["counters"])


def test_masactrl_attention_store_matches_reference(golden_dir):
    """fixture G14 (`tests/golden/make_golden_masa_store.py`, made by the reference's `AttentionStore`,
    `/root/reference/masactrl/model/attention_base.py:33-66`): outputs of every call, counters, and the store's list states
    after every step -- including the upstream aliasing quirk (the running store is emptied with the step lists)"""
    import importlib.util
    from ief_amd.masactrl.model.attention_base import AttentionStore
    spec = importlib.util.spec_from_file_location("mk_store", os.path.join(golden_dir, "make_golden_masa_store.py"))
    z = np.load(os.path.join(golden_dir, "masactrl_store.npz"))

    def calls(seed):             # same generator as the fixture script (which cannot be imported here: it imports the reference)
        g = torch.Generator().manual_seed(seed)
        out = []
        for n, l, cross, place in ((16, 16, False, "down"), (16, 7, True, "down"), (64, 64, False, "up"), (64, 7, True, "up")):
            heads, d, b = 2, 8, 2
            q, k, v = (torch.randn(b * heads, m, d, generator=g) for m in (n, l, l))
            sim = q @ k.transpose(1, 2) * d ** -0.5
            out.append((q, k, v, sim, sim.softmax(-1), cross, place, heads))
        return out

    assert spec is not None
    ed = AttentionStore(res=[32], min_step=1, max_step=4)
    ed.num_att_layers = 4
    for step in range(4):
        for j, c in enumerate(calls(100 + step)):
            assert np.allclose(ed(*c).numpy(), z[f"out_{step}_{j}"], atol=1e-6)
        state = [ed.cur_step, ed.cur_att_layer, ed.valid_steps, len(ed.self_attns), len(ed.cross_attns),
                 len(ed.self_attns_step), len(ed.cross_attns_step)]
        assert state == list(z[f"state_{step}"]), (step, state)
    ed.reset()
    assert [ed.cur_step, ed.cur_att_layer, ed.valid_steps] == list(z["state_reset"])


def test_masactrl_mutual_rule_mirror_vs_einops_vs_oracle():
    """attention_control.py cannot be imported (torchvision); its formulas (:37-66) are restated here with einops"""
    from einops import rearrange
    from ief_amd.masactrl.model.attention_control import MutualSelfAttentionControl
    from oracle.masactrl_ref import MasaCtrlRef
    g = torch.Generator().manual_seed(4)
    heads, B, n, d = 2, 4, 12, 8
    q, k, v = (torch.randn(B * heads, n, d, generator=g) for _ in range(3))
    scale = d ** -0.5

    def attn_batch(q, k, v):      # reference :37-50
        b = q.shape[0] // heads
        q = rearrange(q, "(b h) n d -> h (b n) d", h=heads)
        k = rearrange(k, "(b h) n d -> h (b n) d", h=heads)
        v = rearrange(v, "(b h) n d -> h (b n) d", h=heads)
        a = (torch.einsum("h i d, h j d -> h i j", q, k) * scale).softmax(-1)
        return rearrange(torch.einsum("h i j, h j d -> h i d", a, v), "h (b n) d -> b n (h d)", b=b)

    qu, qc = q.chunk(2); ku, kc = k.chunk(2); vu, vc = v.chunk(2)
    want = torch.cat([attn_batch(qu, ku[:heads], vu[:heads]), attn_batch(qc, kc[:heads], vc[:heads])])
    c = MutualSelfAttentionControl(start_step=0, start_layer=0, total_steps=5)
    c.num_att_layers = 4
    sim = torch.bmm(q, k.transpose(1, 2)) * scale
    got = c(q, k, v, sim, sim.softmax(-1), False, "up", heads, scale=scale)
    assert torch.allclose(got, want, atol=1e-6)
    # oracle hook: same result through plain attention on the swapped k, v
    r = MasaCtrlRef(step_idx=[0], layer_idx=[0], num_att_layers=4)
    q2, k2, v2 = r(q, k, v, False, "up", heads)
    o = torch.bmm((torch.bmm(q2, k2.transpose(1, 2)) * scale).softmax(-1), v2)
    o = o.reshape(B, heads, n, d).permute(0, 2, 1, 3).reshape(B, n, heads * d)
    assert torch.allclose(o, want, atol=1e-6)
    # inactive layer / cross attention: plain attention
    c2 = MutualSelfAttentionControl(start_step=3, start_layer=0, total_steps=5)
    c2.num_att_layers = 4
    plain = c2(q, k, v, sim, sim.softmax(-1), False, "up", heads, scale=scale)
    ref = torch.bmm(sim.softmax(-1), v).reshape(B, heads, n, d).permute(0, 2, 1, 3).reshape(B, n, heads * d)
    assert torch.allclose(plain, ref, atol=1e-6)


# ------------------------------------------------------------------------------------ G9 / G10: SDXL-family inversion loops
def test_oracle_xl_inversion_and_nti_match_reference(golden_dir):
    """`ddim_inversion_xl.ddim_inversion_loop` and `NTI_XL.null_optimization` of the reference
    (`/root/reference/pix2pix-zero/inversion/{ddim,nti}.py:60-96`, imported by `tests/golden/make_golden_xl.py`) on a toy
    UNet that reads `added_cond_kwargs`: the oracle's loops with `added_cond / added_uncond`, lr = 5e-2 and `restart`
    must reproduce them — which kwargs go to which call, the restart from the (zero) negative embedding at every
    timestep, the lr schedule and the early stop"""
    z = np.load(os.path.join(golden_dir, "nti_xl.npz"))
    w1, w2, w3 = (torch.from_numpy(z[k]) for k in ("w1", "w2", "w3"))

    def toy(sd, cfg, x, t, ctx, added_cond_kwargs=None, **kw):
        c = (ctx @ w2).mean(1) + added_cond_kwargs["text_embeds"] @ w3 + 1e-4 * added_cond_kwargs["time_ids"].sum(-1, keepdim=True)
        return torch.tanh(torch.einsum("bchw,cd->bdhw", x, w1) + c[:, :, None, None] + float(t) * 1e-3)

    sched = p2p_ref.DDIMRef(5)
    assert np.array_equal(sched.timesteps.numpy(), z["timesteps"])
    emb, pooled = torch.from_numpy(z["emb"]), torch.from_numpy(z["pooled"])
    ids = torch.tensor([[64.0, 64.0, 0.0, 0.0, 64.0, 64.0]])
    a_c = {"text_embeds": pooled, "time_ids": ids}
    a_u = {"text_embeds": torch.zeros_like(pooled), "time_ids": ids}
    saved = p2p_ref.unet_ref.unet_forward
    p2p_ref.unet_ref.unet_forward = toy
    try:
        lat = p2p_ref.ddim_inversion_loop(None, None, emb, torch.from_numpy(z["x0"]), sched, added_cond_kwargs=a_c)
        got_lat = np.stack([l.numpy() for l in lat])
        assert got_lat.shape == z["inv_latents"].shape and np.allclose(got_lat, z["inv_latents"], atol=1e-6)
        ctx = torch.cat([torch.zeros_like(emb), emb])
        out = p2p_ref.null_optimization(None, None, [torch.from_numpy(a) for a in z["inv_latents"]], ctx, sched, 10, 1e-5, 7.5,
                                        added_cond=a_c, added_uncond=a_u, lr=5e-2, restart=True)
        got = np.stack([u.numpy() for u in out])
        assert got.shape == z["nti_uncond"].shape and np.allclose(got, z["nti_uncond"], atol=1e-6)
        # G11: the P2P folder's copy of NTI_XL: lr = 0.5 (1 - i / 500)
        out2 = p2p_ref.null_optimization(None, None, [torch.from_numpy(a) for a in z["inv_latents"]], ctx, sched, 10, 1e-5, 7.5,
                                         added_cond=a_c, added_uncond=a_u, lr=0.5, lr_decay=500.0, restart=True)
        assert np.allclose(np.stack([u.numpy() for u in out2]), z["nti_uncond_p2p"], atol=1e-5)
        assert not np.allclose(z["nti_uncond_p2p"], z["nti_uncond"], atol=1e-3)
        # the knobs matter: without the restart, or with the SD1.x learning rate, the trajectory is a different one
        for kw in (dict(lr=5e-2, restart=False), dict(lr=1e-2, restart=True)):
            other = p2p_ref.null_optimization(None, None, [torch.from_numpy(a) for a in z["inv_latents"]], ctx, sched, 10, 1e-5,
                                              7.5, added_cond=a_c, added_uncond=a_u, **kw)
            assert not np.allclose(np.stack([u.numpy() for u in other]), z["nti_uncond"], atol=1e-4)
    finally:
        p2p_ref.unet_ref.unet_forward = saved


# ------------------------------------------------------------------------------------ G12: MasaCtrl hooked attention
def test_oracle_attention_matches_reference_masactrl_hook(golden_dir):
    """the reference's `masactrl/model/register.py` hooked forward + `AttentionBase.forward` on a toy tree of Attention
    modules (`tests/golden/make_golden_masa.py`): the oracle's `unet_ref.attention` — head split, scale, softmax, output
    projection, self and cross — reproduces every module's output; the layer count and the editor's counters are the ones
    the product's `AttentionBase` keeps"""
    from oracle import unet_ref
    from ief_amd.masactrl.model.attention_base import AttentionBase
    z = np.load(os.path.join(golden_dir, "masactrl_register.npz"))
    x, ctx = torch.from_numpy(z["x"]), torch.from_numpy(z["ctx"])
    assert int(z["num_att_layers"]) == 8 and int(z["cur_step_after_8_calls"]) == 1 and int(z["cur_att_layer_after_8_calls"]) == 0
    worst = 0.0
    for name in ("down0", "down1", "mid", "up0"):
        for kind in ("attn1", "attn2"):
            p = f"{name}.{kind}"
            sd = {f"{p}.{w}": torch.from_numpy(z[f"{p}.{w}"]) for w in ("to_q.weight", "to_k.weight", "to_v.weight",
                                                                         "to_out.0.weight", "to_out.0.bias")}
            got = unet_ref.attention(sd, p, x, ctx if kind == "attn2" else None, int(z[f"{p}.heads"]), None, name)
            worst = max(worst, (got - torch.from_numpy(z[f"{p}.out"])).abs().max().item())
    assert worst < 1e-5, worst
    # the product's editor base class counts like the reference's (8 calls of an 8-layer net = one step)
    e = AttentionBase()
    e.num_att_layers = 8
    q = torch.randn(4, 16, 32)
    attn = torch.softmax(torch.randn(4, 16, 16), -1)
    for _ in range(8):
        e(q, q, q, None, attn, False, "down", 2)
    assert e.cur_step == 1 and e.cur_att_layer == 0
