"""Split-operand mode (`precision="f16x3"`, csrc/split_x3.hip) on a real MI355X.

fp32 storage as in the reference-precision mode (tests/test_gpu_exact.py), but every contraction runs on THREE fp16 MFMAs
over hi + lo halves of its fp32 operands (Ah Bh + Al Bh + Ah Bl, fp32 accumulate): ~21 operand bits at a third of the fp16
matrix rate, where the fp32-input MFMA gives 24 bits at 1/16 of it.  The reference computes in fp32
(`/root/reference/p2p/edit_syn.py:38`); north_star's bound is 1e-3 max-abs on the edited images.

Stated tolerances (relative to max |reference| unless said otherwise; every test prints what it measured):
    single contractions vs an fp64 reference on the host     <= 4e-6     (fp32-MFMA mode: 2e-5 bound, ~1e-6 measured)
    one UNet forward (eps) vs the fp32 oracle                 <= 1e-4
    10-step edit: latents                                      <= 3e-4
    10-step edit: decoded images in [0, 1]                     max |diff| <= 1e-3, uint8 within 1 level
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from ief_amd import hip  # noqa: E402
from ief_amd.pipeline import StableDiffusionPipeline  # noqa: E402
from ief_amd.p2p.model.attention_control import AttentionRefine, AttentionReplace  # noqa: E402
from ief_amd.p2p.model.register import register_attention_control, unregister_attention_control  # noqa: E402
from ief_amd.p2p.model.sd_utils import P2P, _encode_prompts  # noqa: E402
from oracle import p2p_ref, unet_ref, vae_ref  # noqa: E402

DEV = torch.device("cuda:0")
PROMPTS = ["a photo of a house on a mountain", "a photo of a house on a mountain at fall"]
PROMPTS_EQ = ["a gray horse in the field", "a whie horse in the field"]
XTOL = 4e-6


def f32(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def dev(t):
    return None if t is None else t.cuda()


def rel_err(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all()
    return ((got - ref).abs().max() / ref.abs().max()).item()


@pytest.fixture(autouse=True)
def _x3():
    with hip.f32_contraction("x3"):
        yield


# ----------------------------------------------------------------------------------------------- kernels
def test_gemm_x3_identity_asymmetric():
    """A = I with an asymmetric W of fp16-exact values catches a swapped row / column fragment map and a dropped term"""
    a = torch.eye(128)
    w = torch.arange(96)[:, None] * 0.25 + torch.arange(128)[None, :] * 1.0 + 1.0 / 1024     # hi + lo exactly
    out = hip.gemm(dev(a), dev(w))
    assert out.dtype == torch.float32 and torch.equal(out.cpu(), w.t().contiguous())


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 200, 72), (16384, 320, 320), (77, 1280, 768), (1024, 40, 1024),
                                   (4, 1280, 320), (515, 64, 4), (130, 2560, 5120), (4096, 160, 64), (1000, 480, 96)])
def test_gemm_x3(M, N, K):
    a, w = f32(M, K, seed=1), f32(N, K, seed=2, scale=K ** -0.5)
    bias, res = f32(N, seed=3, scale=0.1), f32(M, N, seed=4)
    out = hip.gemm(dev(a), dev(w), bias=dev(bias), residual=dev(res), out_scale=0.5)
    ref = (a.double() @ w.double().t() + bias.double() + res.double()) * 0.5
    e = rel_err(out, ref.float())
    with hip.f32_contraction("f32"):
        e32 = rel_err(hip.gemm(dev(a), dev(w), bias=dev(bias), residual=dev(res), out_scale=0.5), ref.float())
    print(f"gemm x3 {M}x{N}x{K}: {e:.2e} (fp32 MFMA: {e32:.2e})")
    assert e < XTOL
    rv = f32(2, N, seed=5)
    if M % 2 == 0:
        out = hip.gemm(dev(a), dev(w), rowvec=dev(rv), rows_per_batch=M // 2)
        ref = a.double() @ w.double().t() + rv.double().repeat_interleave(M // 2, 0)
        assert rel_err(out, ref.float()) < XTOL
    wide, owide = f32(M, K + 8, seed=6), torch.zeros(M, N + 4)
    od = dev(owide)
    hip.gemm(dev(wide)[:, 4:4 + K], dev(w), out=od[:, :N])
    assert rel_err(od[:, :N], (wide[:, 4:4 + K].double() @ w.double().t()).float()) < XTOL and od[:, N:].abs().max() == 0


def test_gemm_x3_presplit_weight_planes_equal_on_the_fly_split(monkeypatch):
    """a weight operand is split once per tensor into cached fp16 planes (`hip.x3_weight_planes`) instead of once per launch
    and workgroup; same hi / lo values, same products: bit-identical outputs, linear and 3x3 (fast and general loaders: K
    not a multiple of 32 ends in the general one)"""
    a, w = f32(700, 328, seed=1), f32(320, 328, seed=2, scale=0.05)
    x, wc = f32(2, 16, 16, 64, seed=3), f32(160, 3, 3, 64, seed=4, scale=0.04)
    outs = {}
    for pre in (True, False):
        monkeypatch.setattr(hip, "X3_PRESPLIT", pre)
        hip.profile_begin()
        outs[pre] = (hip.gemm(dev(a), dev(w)), hip.conv3x3(dev(x), dev(wc)))
        hip.profile_end()
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    wd = dev(w)
    p1 = hip.x3_weight_planes(wd)
    assert p1 is hip.x3_weight_planes(wd) and p1.shape == (2, 320, 328) and p1.dtype == torch.float16        # cached
    hi, lo = p1[0].float(), p1[1].float()
    assert ((hi + lo) / hip.X3_SCALE_W - wd).abs().max().item() <= 2 ** -21 * w.abs().max().item()
    wd.mul_(2.0)                                                                                            # a changed tensor is re-split
    assert hip.x3_weight_planes(wd) is not p1


@pytest.mark.parametrize("M,Ch,K", [(4096, 1280, 320), (300, 640, 72), (64, 40, 32), (333, 160, 64)])
def test_gemm_x3_fused_geglu(M, Ch, K, monkeypatch):
    """FeedForward.net[0] + GEGLU in ONE launch: weight rows interleaved [8 hidden | 8 gate] (the layout `unet.GEGLU` packs), the
    epilogue pairs a hidden lane with its gate lane and writes hidden * gelu(gate) -- against fp64 and against the two-launch
    form (GEMM, then ief_geglu_il_f32)"""
    a, w, bias = f32(M, K, seed=1), f32(2 * Ch, K, seed=2, scale=K ** -0.5), f32(2 * Ch, seed=3, scale=0.1)
    out = hip.gemm(dev(a), dev(w), bias=dev(bias), geglu=True)
    pre = (a.double() @ w.double().t() + bias.double()).reshape(M, Ch // 8, 2, 8)
    ref = (pre[:, :, 0] * F.gelu(pre[:, :, 1])).reshape(M, Ch)
    e = rel_err(out, ref.float())
    monkeypatch.setattr(hip, "X3_FUSE_GEGLU", False)
    two = hip.gemm(dev(a), dev(w), bias=dev(bias), geglu=True)
    print(f"fused GEGLU {M}x{2 * Ch}x{K}: {e:.2e} vs fp64; vs the two-launch form {rel_err(out, two):.2e}")
    assert out.shape == (M, Ch) and e < XTOL and rel_err(out, two) < 1e-6


def test_gemm_x3_wide_tile_equals_the_narrow_tile():
    """the launches that take the 128 x 160 tile on 8 waves (`ief_gemm_x3_bn_k`) run the same MFMA sequence per output
    element as on the 128 x 80 tile: bit-identical results, linear (+ fused GEGLU) and 3x3 with concat and 1x1 extra sources"""
    import ctypes
    lib = hip.load()
    lib.ief_gemm_x3_set_variant.argtypes = [ctypes.c_int]
    assert lib.ief_gemm_x3_bn_k(0, 320, 256) == 160 and lib.ief_gemm_x3_bn_k(1, 160, 960) == 160 and lib.ief_gemm_x3_bn_k(0, 200, 64) == 64
    a, w, bias = f32(1000, 256, seed=1), f32(320, 256, seed=2, scale=0.06), f32(320, seed=3, scale=0.1)
    x, x2 = f32(2, 16, 16, 64, seed=4), f32(2, 16, 16, 32, seed=5)
    wc = f32(160, 9 * 96 + 96, seed=6, scale=0.03)
    outs = []
    try:
        for wide in (1, 0):
            lib.ief_gemm_x3_set_variant(wide)
            outs.append((hip.gemm(dev(a), dev(w), bias=dev(bias)), hip.gemm(dev(a), dev(w), bias=dev(bias), geglu=True),
                         hip.conv3x3(dev(x), dev(wc), x2=dev(x2), extra=(dev(x), dev(x2)))))
    finally:
        lib.ief_gemm_x3_set_variant(1)
    for got, ref in zip(*outs):
        assert torch.equal(got, ref)


@pytest.mark.parametrize("sa,sw", [(1e-4, 1e-3), (1e-2, 1e-5), (300.0, 1.0), (1.0, 30.0)])
def test_gemm_x3_operand_magnitudes(sa, sw):
    """operands far from unit scale: the lo halves of small elements fall into the fp16 subnormal range (whether the MFMA
    keeps or flushes them decides the error), large ones approach the end of the fp16 range (65504 / scale: activations
    16376, weights 255 -- beyond it the output is NaN, as on any fp16 path, not silently wrong)"""
    M, N, K = 256, 128, 640
    a, w = f32(M, K, seed=1) * sa, f32(N, K, seed=2) * sw
    e = rel_err(hip.gemm(dev(a), dev(w)), (a.double() @ w.double().t()).float())
    print(f"gemm x3 operand scales {sa:g} x {sw:g}: rel err {e:.2e}")
    assert e < 2e-4          # fp16 operands: ~3e-4; the stated bound of the mode holds at unit scale (test_gemm_x3)


@pytest.mark.parametrize("M,N,K", [(256, 40, 77), (4096, 64, 4096), (100, 160, 130), (64, 512, 64)])
def test_gemm_nt_x3(M, N, K):
    a, b = f32(M, K, seed=1, scale=K ** -0.5), f32(K, N, seed=2)
    e = rel_err(hip.gemm_nt(dev(a), dev(b)), (a.double() @ b.double()).float())
    print(f"gemm_nt x3 {M}x{N}x{K}: {e:.2e}")
    assert e < XTOL


@pytest.mark.parametrize("B,H,C1,C2,Cout,stride,ups,extra,hi", [
    (2, 16, 64, 0, 128, 1, False, False, False), (1, 32, 320, 0, 320, 1, False, False, False),
    (2, 16, 128, 64, 128, 1, False, True, False), (2, 16, 64, 0, 64, 2, False, False, False),
    (2, 8, 128, 0, 128, 1, True, False, False), (1, 16, 128, 0, 128, 2, False, False, True),
    (4, 8, 64, 64, 64, 1, False, False, False), (1, 8, 32, 0, 36, 1, False, False, False),
    # Cout % 160 == 0: the 128 x 160 tile on 8 waves -- concat + 1x1 extra sources, stride 2, one-sided padding
    (2, 16, 128, 64, 160, 1, False, True, False), (2, 16, 64, 0, 160, 2, False, False, False),
    (1, 16, 128, 0, 320, 2, False, False, True), (4, 8, 64, 64, 160, 1, False, False, False),
    (1, 12, 32, 0, 160, 1, False, False, False)])
def test_conv3x3_x3(B, H, C1, C2, Cout, stride, ups, extra, hi):
    x, x2 = f32(B, H, H, C1, seed=1), (f32(B, H, H, C2, seed=2) if C2 else None)
    Ct = C1 + C2
    w = f32(Cout, 3, 3, Ct, seed=3, scale=(9 * Ct) ** -0.5)
    bias, rv = f32(Cout, seed=4, scale=0.1), f32(B, Cout, seed=5, scale=0.2)
    xin = x if x2 is None else torch.cat([x, x2], -1)
    xn = xin.permute(0, 3, 1, 2).double()
    if ups:
        xn = F.interpolate(xn, scale_factor=2.0, mode="nearest")
    if hi:
        xn = F.pad(xn, (0, 1, 0, 1))
    ref = F.conv2d(xn, w.permute(0, 3, 1, 2).double(), bias.double(), stride=stride, padding=0 if hi else 1)
    ref = ref + rv.double()[:, :, None, None]
    if extra:
        ws = f32(Cout, Ct, seed=6, scale=Ct ** -0.5)
        ref = ref + F.conv2d(xin.permute(0, 3, 1, 2).double(), ws.double()[:, :, None, None])
        wf = torch.cat([w.reshape(Cout, 9 * Ct), ws], 1)
        out = hip.conv3x3(dev(x), dev(wf), dev(bias), x2=dev(x2), rowvec=dev(rv), extra=(dev(x), dev(x2)))
    else:
        res = f32(*ref.permute(0, 2, 3, 1).shape, seed=7)
        ref = ref + res.permute(0, 3, 1, 2).double()
        out = hip.conv3x3(dev(x), dev(w), dev(bias), x2=dev(x2), stride=stride, upsample=ups, rowvec=dev(rv),
                          residual=dev(res), pad_hi_only=hi)
    assert out.dtype == torch.float32
    e = rel_err(out, ref.permute(0, 2, 3, 1).float())
    print(f"conv x3: {e:.2e}")
    assert e < XTOL


def _attn_ref(q, k, v, heads, scale, qs=None, ks=None, vs=None):
    B, N, C = q.shape
    d = C // heads
    idx = lambda t, s: t if s is None else t[torch.as_tensor(s).long()]
    q, k, v = idx(q.double(), qs), idx(k.double(), ks), idx(v.double(), vs)
    L = k.shape[1]
    qh = q.reshape(B, N, heads, d).permute(0, 2, 1, 3)
    kh = k.reshape(B, L, heads, d).permute(0, 2, 1, 3)
    vh = v.reshape(B, L, heads, d).permute(0, 2, 1, 3)
    p = torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1)
    return (p @ vh).permute(0, 2, 1, 3).reshape(B, N, C).float(), p.float()


@pytest.mark.parametrize("B,heads,N,L,d", [(2, 8, 1024, 1024, 40), (4, 8, 256, 256, 80), (2, 2, 200, 144, 64),
                                           (2, 4, 100, 77, 160), (1, 3, 96, 77, 32), (1, 2, 130, 33, 40)])
def test_attention_x3(B, heads, N, L, d):
    """fused kernel and the materialised pipeline (scores -> softmax -> P.V) on split operands, with the batch-row
    indirection of P2P self-replace / MasaCtrl"""
    C = heads * d
    qkv, kv = f32(B, N, 3 * C, seed=1), f32(B, L, 2 * C, seed=2)
    qd, kd = dev(qkv), dev(kv)
    q, k, v = qkv[..., :C], kv[..., :C], kv[..., C:]
    scale = d ** -0.5
    ref, refp = _attn_ref(q, k, v, heads, scale)
    out = hip.attn_flash(qd[..., :C], kd[..., :C], kd[..., C:], heads, scale)
    mat = hip._attn_apply_f32(hip._attn_scores_f32(qd[..., :C], kd[..., :C], heads, scale), kd[..., C:], heads)
    e_f, e_m = rel_err(out, ref), rel_err(mat, ref)
    probs = hip.attn_probs(qd[..., :C], kd[..., :C], heads, scale)
    e_p = (probs.cpu() - refp.reshape(B * heads, N, L)).abs().max().item()
    print(f"attention x3 N={N} L={L} d={d}: fused {e_f:.2e} materialised {e_m:.2e} maps (abs) {e_p:.2e}")
    assert out.dtype == torch.float32 and e_f < XTOL and e_m < XTOL and e_p < 2e-6
    if B >= 2:
        src = torch.tensor([0] + [0] * (B - 1), dtype=torch.int32)
        keep = torch.arange(B, dtype=torch.int32)
        out = hip.attn_flash(qd[..., :C], kd[..., :C], kd[..., C:], heads, scale, q_src=dev(src), k_src=dev(src), v_src=dev(keep))
        assert rel_err(out, _attn_ref(q, k, v, heads, scale, src, src, keep)[0]) < XTOL
        out = hip.attn_flash(qd[..., :C], kd[..., :C], kd[..., C:], heads, scale, k_src=dev(src), v_src=dev(src))
        assert rel_err(out, _attn_ref(q, k, v, heads, scale, None, src, src)[0]) < XTOL


def test_attention_x3_peaky_rows_force_the_rescale():
    B, heads, N, L, d = 1, 2, 160, 512, 64
    q, k, v = f32(B, N, heads * d, seed=1), f32(B, L, heads * d, seed=2), f32(B, L, heads * d, seed=3)
    for t in range(L // 32):
        k[0, 32 * t + 7, :] = q[0, 5, :] * (0.5 + 0.25 * t)
    out = hip.attn_flash(dev(q), dev(k), dev(v), heads, d ** -0.5 * 4.0)
    e = rel_err(out, _attn_ref(q, k, v, heads, d ** -0.5 * 4.0)[0])
    print(f"attention x3 peaky rows: {e:.2e}")
    assert e < XTOL


# ----------------------------------------------------------------------------------------------- whole path
@pytest.fixture(scope="module")
def tinyx3():
    return StableDiffusionPipeline.from_pretrained("synthetic:tiny", keep_state_dict=True, precision="f16x3")


@pytest.fixture(scope="module")
def smallx3():
    return StableDiffusionPipeline.from_pretrained("synthetic:small", keep_state_dict=True, precision="f16x3")


def _inputs(cfg, B, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, 4, cfg.sample_size, cfg.sample_size, generator=g),
            torch.randn(B, 77, cfg.cross_attention_dim, generator=g))


@pytest.mark.parametrize("name,B", [("tiny", 2), ("small", 4)])
def test_unet_forward_x3_vs_oracle(name, B, tinyx3, smallx3):
    pipe = {"tiny": tinyx3, "small": smallx3}[name]
    assert pipe.unet.dtype == torch.float32 and pipe.unet.contract == "x3"
    x, ctx = _inputs(pipe.cfg, B)
    for t in (981, 1):
        hip.profile_begin()
        eps = pipe.unet(x.to(DEV), t, encoder_hidden_states=ctx.to(DEV))["sample"]
        names = {r[0] for r in hip.profile_end()}
        assert any(n.startswith("igemm_x3_kernel") for n in names) and not any(n.startswith("igemm_f32_kernel") for n in names), \
            f"the split-operand kernels must serve every contraction of the forward: {sorted(names)}"
        ref = unet_ref.unet_forward(pipe._state_dict, pipe.cfg, x, torch.tensor(t), ctx)
        e = rel_err(eps, ref)
        print(f"x3 {name} B={B} t={t}: rel err {e:.3e}")
        assert e < 1e-4


@pytest.mark.parametrize("kind,step", [("refine", 0), ("refine", 25), ("replace", 3)])
def test_p2p_controlled_forward_x3(kind, step, smallx3):
    pipe = smallx3
    cfg = pipe.cfg
    x1, ctx = _inputs(cfg, 4, seed=3)
    x = torch.cat([x1[:1], 0.8 * x1[:1] + 0.6 * x1[1:2]] * 2)
    make = (lambda: AttentionRefine(PROMPTS, pipe.tokenizer, 50, 0.8, 0.4, device=DEV)) if kind == "refine" else \
        (lambda: AttentionReplace(PROMPTS_EQ, pipe.tokenizer, 50, 0.8, 0.4, device=DEV))
    outs = {}
    for fused in (True, False):
        c = make()
        register_attention_control(pipe, c, fused=fused)
        c.cur_step = step
        outs[fused] = pipe.unet(x.to(DEV), 501, encoder_hidden_states=ctx.to(DEV))["sample"].cpu()
        assert c.cur_step == step + 1 and c.cur_att_layer == 0
        unregister_attention_control(pipe, c)
    c = make()
    rc = p2p_ref.P2PControlRef(mode=kind, num_prompts=2, cross_alpha=c.cross_replace_alpha.float().cpu(),
                               num_self_replace=c.num_self_replace, mapper=c.mapper.cpu(),
                               alphas=c.alphas.float().cpu() if hasattr(c, "alphas") else None)
    rc.num_att_layers = unet_ref.count_attention_layers(cfg)
    rc.cur_step = step
    ref = unet_ref.unet_forward(pipe._state_dict, cfg, x, torch.tensor(501), ctx, hook=rc)
    plain = unet_ref.unet_forward(pipe._state_dict, cfg, x, torch.tensor(501), ctx)
    e_f, e_g, effect = rel_err(outs[True], ref), rel_err(outs[False], ref), rel_err(plain, ref)
    print(f"x3 {kind} step {step}: fused {e_f:.3e} generic {e_g:.3e} (the edit moves eps by {effect:.3e})")
    assert e_f < 1e-4 and e_g < 1e-4 and effect > 100 * e_f


def test_full_edit_images_x3_within_1e3(tinyx3):
    """north_star's bound on the split-operand mode: `P2P.text2image_ldm_stable` end to end (text encode -> 10-step
    AttentionRefine edit in the captured step graph -> AutoencoderKL decode): decoded images in [0, 1] within 1e-3 max-abs
    of the fp32 oracle's, uint8 images within one grey level"""
    from ief_amd.vae import synthetic_vae_state_dict
    pipe = tinyx3
    cfg = pipe.cfg
    n = 10
    editor = P2P(pipe, n)
    x_T = torch.randn(1, 4, cfg.sample_size, cfg.sample_size, generator=torch.Generator().manual_seed(8888))
    c = AttentionRefine(PROMPTS, pipe.tokenizer, n, 0.8, 0.4, device=DEV)
    images, _ = editor.text2image_ldm_stable(pipe, PROMPTS, c, num_inference_steps=n, guidance_scale=7.5, latent=x_T.to(DEV))
    assert pipe.unet._plan is not None and pipe.unet._plan.kind == "p2p" and c.cur_step == n
    c2 = AttentionRefine(PROMPTS, pipe.tokenizer, n, 0.8, 0.4, device=DEV)
    lat, _ = editor.text2image_ldm_stable(pipe, PROMPTS, c2, num_inference_steps=n, guidance_scale=7.5, latent=x_T.to(DEV),
                                          return_latents=True)
    unregister_attention_control(pipe, c2)
    with torch.no_grad():
        u, cnd = _encode_prompts(pipe, PROMPTS)
    c3 = AttentionRefine(PROMPTS, pipe.tokenizer, n, 0.8, 0.4, device=DEV)
    rc = p2p_ref.P2PControlRef(mode="refine", num_prompts=2, cross_alpha=c3.cross_replace_alpha.float().cpu(),
                               num_self_replace=c3.num_self_replace, mapper=c3.mapper.cpu(), alphas=c3.alphas.float().cpu())
    ref_lat = p2p_ref.edit_loop(pipe._state_dict, cfg, torch.cat([u, cnd]).float().cpu(), x_T, rc, p2p_ref.DDIMRef(n), 7.5)
    vsd = synthetic_vae_state_dict(pipe.vae.cfg, 2)
    ref_dec = vae_ref.decode(vsd, pipe.vae.cfg, ref_lat / pipe.vae.cfg.scaling_factor)
    got_dec = pipe.vae.decode(lat / pipe.vae.cfg.scaling_factor)["sample"].cpu()
    e_lat = rel_err(lat, ref_lat)
    d_img = ((got_dec / 2 + 0.5).clamp(0, 1) - (ref_dec / 2 + 0.5).clamp(0, 1)).abs().max().item()
    diff = abs(images.astype(int) - p2p_ref.latent_to_uint8(ref_dec).astype(int))
    print(f"x3 10-step edit: latents rel err {e_lat:.3e}; images in [0,1] max |diff| {d_img:.3e}; uint8 max diff {diff.max()}, "
          f"identical pixels {(diff == 0).mean():.4f}")
    assert e_lat < 3e-4 and d_img <= 1e-3 and diff.max() <= 1


def test_pooled_loop_rerun_is_bit_identical_to_its_first_run(tinyx3):
    """a captured loop re-pointed at the next image (`denoise.acquire` -> `rebind`) refreshes the cross-attention K / V the
    graph reads; they must come from the SAME contraction kernels the forward used when the loop was captured ("x3" here, not
    the library default), else an image depends on whether it was the first of its shape in the process -- which made the
    two-rank PIE driver differ from the one-rank run by a grey level"""
    from ief_amd.denoise import acquire, drop_pool
    pipe = tinyx3
    hw = pipe.unet.config.sample_size
    pipe.scheduler.set_timesteps(50)
    g = torch.Generator().manual_seed(3)
    ctx = dev(torch.randn(1, 77, pipe.unet.config.cross_attention_dim, generator=g) * 0.3)
    x0 = dev(torch.randn(1, 4, hw, hw, generator=g))
    drop_pool()
    outs = []
    for _ in range(3):
        loop = acquire(pipe, ctx, 1, (hw, hw), None, mode="invert")
        outs.append(loop.run(x0, num_steps=4).cpu())
        loop.release()
    drop_pool()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("heads,N,L,d", [(8, 256, 77, 40), (2, 200, 77, 80), (2, 64, 77, 160), (4, 130, 40, 64), (1, 1024, 96, 40)])
def test_cross_attention_p2p_edit_fused_x3(heads, N, L, d, monkeypatch):
    """`ief_attn_cross_p2p_f32`: scores, softmax, P' = c1 (P_src M) + c2 P_tgt (`/root/reference/p2p/model/attention_base.py:118-121`)
    and P'.V of an edited cross-attention layer in ONE launch with the maps in registers -- against fp64 and against the
    four-launch materialised form; two edited rows with different tables, a table entry that is not an fp16 number, query
    counts that are not multiples of 128, 40 / 77 / 96 keys"""
    B, C = 4, heads * d
    q, k, v = f32(B, N, C, seed=1), f32(B, L, C, seed=2, scale=1.5), f32(B, L, C, seed=3)
    g = torch.Generator().manual_seed(0)
    mt, coef = torch.zeros(2, 96, 96), torch.zeros(2, 2, 96)
    Ms, cs = [], []
    for s in range(2):
        mapper = torch.randint(-1, L, (L,), generator=g)
        a = (mapper != -1).float()
        M = torch.zeros(L, L)
        M[mapper % L, torch.arange(L)] = 1.0
        M[5, 5], M[5, 6] = 1.0 / 3.0, 2.0 / 3.0
        gate = (torch.rand(L, generator=g) > 0.3).float() * (0.25 + 0.75 * torch.rand(L, generator=g))
        c1, c2 = gate * a, 1 - gate * a
        mt[s, :L, :L] = M.t()
        coef[s, 0, :L], coef[s, 1, :L] = c1, c2
        Ms.append(M.double()), cs.append((c1.double(), c2.double()))
    es, sl = torch.tensor([-1, 0, -1, 2], dtype=torch.int32), torch.tensor([0, 1, 0, 0], dtype=torch.int32)
    qh = q.double().reshape(B, N, heads, d).permute(0, 2, 1, 3)
    kh = k.double().reshape(B, L, heads, d).permute(0, 2, 1, 3)
    vh = v.double().reshape(B, L, heads, d).permute(0, 2, 1, 3)
    P = torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1)
    Pe = P.clone()
    for b in range(B):
        if es[b] >= 0:
            c1, c2 = cs[sl[b]]
            Pe[b] = c1 * (P[es[b]] @ Ms[sl[b]]) + c2 * P[b]
    ref = (Pe @ vh).permute(0, 2, 1, 3).reshape(B, N, C).float()
    args = (dev(q), dev(k), dev(v), heads, d ** -0.5, dev(es), dev(sl), dev(mt), dev(coef))
    hip.profile_begin()
    out = hip.attn_cross_p2p(*args)
    names = [n for n, _, _ in hip.profile_end()]
    assert names == [f"attn_cross_p2p_x3_kernel<{d}>"], names
    monkeypatch.setattr(hip, "X3_FUSE_CROSS", False)
    mat = hip.attn_cross_p2p(*args)
    e, em = rel_err(out, ref), rel_err(mat, ref)
    print(f"cross-attention + P2P edit fused N={N} L={L} d={d}: {e:.2e} vs fp64 (materialised form {em:.2e})")
    assert out.dtype == torch.float32 and e < XTOL and em < XTOL


@pytest.mark.parametrize("equalizer", [5.0, 10.0, 40.0])
def test_cross_attention_reweight_large_equalizer_x3(equalizer, monkeypatch):
    """AttentionReweight lowers to c1 = alpha * equalizer (`/root/reference/p2p/model/attention_control.py:42-46`): the edited map
    P' = c1 T + c2 P reaches `equalizer` on a peaky source row.  With the fixed map scale 2^14 the hi half of P' overflowed
    fp16 at P' >= 4 (inf - inf = NaN for the whole image); the scale now comes from the plan's coefficient bound
    (`hip.map_split_scale`), in the fused kernel and in the materialised form."""
    heads, N, L, d, B = 2, 256, 77, 40, 4
    C = heads * d
    q, k, v = f32(B, N, C, seed=1), f32(B, L, C, seed=2, scale=1.5), f32(B, L, C, seed=3)
    k[0, 7] = q[0, :, :].mean(0) * 0 + 3.0 * torch.sign(f32(C, seed=9))          # a key that dominates many source rows: maps near 1
    q[0] = q[0] + 2.0 * torch.sign(f32(C, seed=9))
    mt, coef = torch.zeros(1, 96, 96), torch.zeros(1, 2, 96)
    mt[0, :L, :L] = torch.eye(L)
    coef[0, 0, :L] = 1.0
    coef[0, 0, 7] = equalizer
    es, sl = torch.tensor([-1, 0, -1, 0], dtype=torch.int32), torch.zeros(B, dtype=torch.int32)
    qh = q.double().reshape(B, N, heads, d).permute(0, 2, 1, 3)
    kh = k.double().reshape(B, L, heads, d).permute(0, 2, 1, 3)
    vh = v.double().reshape(B, L, heads, d).permute(0, 2, 1, 3)
    P = torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1)
    assert P[0, :, :, 7].max() > 0.8, "the source map must be peaky for the test to mean anything"
    Pe = P.clone()
    for b in (1, 3):
        Pe[b] = coef[0, 0, :L].double() * P[0]
    ref = (Pe @ vh).permute(0, 2, 1, 3).reshape(B, N, C).float()
    bound = float((coef[:, 0].abs() + coef[:, 1].abs()).max())
    args = (dev(q), dev(k), dev(v), heads, d ** -0.5, dev(es), dev(sl), dev(mt), dev(coef))
    out = hip.attn_cross_p2p(*args, coef_bound=bound)
    monkeypatch.setattr(hip, "X3_FUSE_CROSS", False)
    mat = hip.attn_cross_p2p(*args, coef_bound=bound)
    e, em = rel_err(out, ref), rel_err(mat, ref)
    print(f"reweight equalizer {equalizer}: max P' {Pe.max():.2f}, map split scale {hip.map_split_scale(bound):.0f}: fused {e:.2e}, "
          f"materialised {em:.2e} vs fp64")
    assert e < XTOL and em < XTOL


def test_x3_activations_of_2e4_stay_finite_and_exact():
    """the activation scale of the split is 1 (the fp16 range itself, 65504; it was 4: NaN beyond 16376): operands of ~2e4 go
    through the in-kernel split, the planes GEMM and the fused attention and stay as exact as O(1) operands"""
    from ief_amd import planes
    a, w = f32(300, 320, seed=1, scale=6e3), f32(160, 320, seed=2, scale=0.05)
    assert a.abs().max() > 2e4
    ref = a.double() @ w.double().t()
    e_old = rel_err(hip.gemm(dev(a), dev(w)), ref.float())
    e_new = rel_err(planes.gemm(planes.split(dev(a)), dev(w)), ref.float())
    print(f"operands up to {a.abs().max():.3g}: in-kernel split {e_old:.2e}, planes {e_new:.2e}")
    assert e_old < XTOL and e_new < XTOL
    B, heads, N, d = 1, 2, 256, 40
    q, k, v = f32(B, N, heads * d, seed=3, scale=0.5), f32(B, N, heads * d, seed=4, scale=0.5), f32(B, N, heads * d, seed=5, scale=6e3)
    out = hip.attn_flash(dev(q), dev(k), dev(v), heads, d ** -0.5)
    qh, kh, vh = (t.double().reshape(B, N, heads, d).permute(0, 2, 1, 3) for t in (q, k, v))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1) @ vh).permute(0, 2, 1, 3).reshape(B, N, heads * d)
    assert rel_err(out, ref.float()) < XTOL
