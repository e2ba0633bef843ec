"""Activation-gradient kernels and the null-text-inversion loop on a real MI355X.

Checkers: torch autograd in fp32 on the same (fp16-rounded) inputs for single kernels, and the fp32 CPU
oracle (`oracle/unet_ref.py` under autograd, `oracle/p2p_ref.null_optimization`) for the whole UNet
gradient and the NTI loop.  Stated tolerances (relative to max |reference|):
    single adjoint kernels                         <= 1e-2   (fp16 storage of gradients)
    attention backward                             <= 2e-2
    d loss / d encoder_hidden_states, whole UNet   <= 5e-2   (61 norms + 32 attentions deep, fp16 gradients)
    Adam step vs torch.optim.Adam                  <= 1e-6 absolute on the parameters
Measured values are printed (`-s`) and recorded in DESIGN.md.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from ief_amd import hip  # noqa: E402
from ief_amd.grad import UNetAdjoint  # noqa: E402
from ief_amd.nti import NullTextOptimizer  # noqa: E402
from ief_amd.pipeline import StableDiffusionPipeline  # noqa: E402
from ief_amd.p2p.inversion.nti import NTI  # noqa: E402
from oracle import p2p_ref, unet_ref  # noqa: E402

DEV = torch.device("cuda:0")


def rel_err(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert torch.isfinite(got).all()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def h16(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half().to(DEV)


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("d,heads,N,L", [(40, 2, 320, 320), (40, 2, 200, 77), (80, 2, 256, 256), (160, 1, 64, 64),
                                         (160, 1, 64, 77), (64, 2, 16, 16), (32, 4, 4, 4), (32, 2, 130, 77),
                                         (64, 1, 1024, 1024)])
def test_attn_bwd_vs_autograd(d, heads, N, L):
    B, C = 2, heads * d
    q, k, v, do = h16(B, N, C, seed=1), h16(B, L, C, seed=2), h16(B, L, C, seed=3), h16(B, N, C, seed=4, scale=0.05)
    scale = d ** -0.5
    lse = torch.empty(B, heads, N, dtype=torch.float32, device=DEV)
    o = hip.attn_flash(q, k, v, heads, scale, lse=lse)
    dq, dk, dv = hip.attn_bwd(q, k, v, o, do, lse, heads, scale)
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    sp = lambda t, n: t.reshape(B, n, heads, d).transpose(1, 2)
    ref = (torch.softmax(sp(qf, N) @ sp(kf, L).transpose(-1, -2) * scale, -1) @ sp(vf, L)).transpose(1, 2).reshape(B, N, C)
    ref.backward(do.float())
    # the stored log-sum-exp itself
    s2 = (sp(q.float(), N) @ sp(k.float(), L).transpose(-1, -2)) * scale * math.log2(math.e)
    lse_ref = torch.logsumexp(s2 * math.log(2.0), -1) / math.log(2.0)
    assert (lse - lse_ref).abs().max().item() < 2e-2
    errs = [rel_err(dq, qf.grad), rel_err(dk, kf.grad), rel_err(dv, vf.grad)]
    print(f"attn_bwd d={d} N={N} L={L}: dq {errs[0]:.2e} dk {errs[1]:.2e} dv {errs[2]:.2e}")
    assert max(errs) < 2e-2


def test_attn_bwd_strided_outputs():
    """dK/dV written into column slices of a wide buffer (how the cross-attention layers share one GEMM)."""
    B, heads, d, N, L = 1, 2, 40, 256, 77
    C = heads * d
    q, do = h16(B, N, C, seed=1), h16(B, N, C, seed=4, scale=0.05)
    kv = h16(B, L, 4 * C, seed=2)
    k, v = kv[..., C:2 * C], kv[..., 3 * C:]
    scale = d ** -0.5
    lse = torch.empty(B, heads, N, dtype=torch.float32, device=DEV)
    o = hip.attn_flash(q, k, v, heads, scale, lse=lse)
    wide = torch.zeros(B, L, 4 * C, dtype=torch.float16, device=DEV)
    hip.attn_bwd(q, k, v, o, do, lse, heads, scale, dk=wide[..., C:2 * C], dv=wide[..., 3 * C:], want_dq=False)
    _, dk, dv = hip.attn_bwd(q, k.contiguous(), v.contiguous(), o, do, lse, heads, scale, want_dq=False)
    assert torch.equal(wide[..., C:2 * C], dk) and torch.equal(wide[..., 3 * C:], dv)
    assert wide[..., :C].abs().max() == 0 and wide[..., 2 * C:3 * C].abs().max() == 0


# ------------------------------------------------------------------------------------------------ norms, GEGLU
@pytest.mark.parametrize("C1,C2,HW,silu", [(320, 0, 1024, True), (640, 320, 256, True), (1280, 640, 64, True),
                                           (64, 0, 256, False), (128, 64, 16, True), (320, 0, 4096, False),
                                           (640, 320, 4096, True), (640, 640, 1024, True)])
def test_groupnorm_bwd_vs_autograd(C1, C2, HW, silu):
    B, G, C = 2, 32, C1 + C2
    x, x2 = h16(B, HW, C1, seed=1), (h16(B, HW, C2, seed=2) if C2 else None)
    dy, add = h16(B, HW, C, seed=3, scale=0.1), h16(B, HW, C, seed=4, scale=0.1)
    g = torch.Generator().manual_seed(5)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(C, generator=g)).to(DEV)
    xin = (torch.cat([x, x2], -1) if C2 else x).float().requires_grad_(True)
    y = F.group_norm(xin.transpose(1, 2), G, gamma, beta, 1e-5).transpose(1, 2)
    (F.silu(y) if silu else y).backward(dy.float())
    ref = xin.grad + add.float()
    # the forward's saved statistics: what the split two-launch path (large slabs) consumes
    _, st = hip.groupnorm(x, gamma, beta, G, 1e-5, silu=silu, x2=x2, return_stats=True)
    xg = xin.detach().reshape(B, HW, G, C // G).permute(0, 2, 1, 3).reshape(B, G, -1)
    assert rel_err(st[..., 0], xg.mean(-1)) < 1e-3 or xg.mean(-1).abs().max() < 1e-2
    assert rel_err(st[..., 1], (xg.var(-1, unbiased=False) + 1e-5).rsqrt()) < 1e-3
    for stats in (None, st):
        got = hip.groupnorm_bwd(x, dy, gamma, beta, G, 1e-5, silu=silu, x2=x2, add=add, stats=stats)
        if C2:
            e = max(rel_err(got[0], ref[..., :C1]), rel_err(got[1], ref[..., C1:]))
        else:
            e = rel_err(got, ref)
        print(f"groupnorm_bwd C={C1}+{C2} HW={HW} silu={silu} stats={'saved' if stats is not None else 'recomputed'}: {e:.2e}")
        assert e < 1e-2


@pytest.mark.parametrize("C", [64, 320, 1280])
def test_layernorm_bwd_vs_autograd(C):
    x, dy, add = h16(3, 100, C, seed=1), h16(3, 100, C, seed=2, scale=0.1), h16(3, 100, C, seed=3, scale=0.1)
    gamma = (1 + 0.1 * torch.randn(C, generator=torch.Generator().manual_seed(5))).to(DEV)
    got = hip.layernorm_bwd(x, dy, gamma, 1e-5, add=add)
    xin = x.float().requires_grad_(True)
    F.layer_norm(xin, (C,), gamma, torch.zeros_like(gamma), 1e-5).backward(dy.float())
    e = rel_err(got, xin.grad + add.float())
    print(f"layernorm_bwd C={C}: {e:.2e}")
    assert e < 1e-2


def test_geglu_interleaved_fwd_bwd():
    rows, Ch = 300, 1280
    pre, dy = h16(rows, 2 * Ch, seed=1), h16(rows, Ch, seed=2, scale=0.1)
    out, dpre = hip.geglu_il(pre), hip.geglu_il_bwd(pre, dy)
    p = pre.float().reshape(rows, Ch // 8, 2, 8).requires_grad_(True)
    ref = (p[:, :, 0] * F.gelu(p[:, :, 1])).reshape(rows, Ch)
    ref.backward(dy.float())
    assert rel_err(out, ref.detach()) < 2e-3
    assert rel_err(dpre, p.grad.reshape(rows, 2 * Ch)) < 2e-3


# ------------------------------------------------------------------------------------------------ conv adjoints
def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("mode", ["plain", "stride2", "upsample"])
def test_conv_data_gradient_vs_autograd(mode):
    B, Cin, Cout, H = 2, 128, 64, 16
    x = h16(B, Cin, H, H, seed=1)
    w = h16(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    xin = x.float().requires_grad_(True)
    if mode == "plain":
        y = F.conv2d(xin, w.float(), padding=1)
    elif mode == "stride2":
        y = F.conv2d(xin, w.float(), padding=1, stride=2)
    else:
        y = F.conv2d(F.interpolate(xin, scale_factor=2.0, mode="nearest"), w.float(), padding=1)
    dy = h16(*y.shape, seed=3, scale=0.1)
    y.backward(dy.float())
    adj = UNetAdjoint.__new__(UNetAdjoint)
    adj._wt = {}
    wt = adj.wt_conv(_nhwc(w))                     # forward packing is [Cout, 3, 3, Cin]
    d = _nhwc(dy)
    if mode == "plain":
        got = hip.conv3x3(d, wt)
    elif mode == "stride2":
        got = hip.conv3x3(hip.zero_insert2x(d), wt)
    else:
        got = hip.pool2x2_sum(hip.conv3x3(d, wt))
    e = rel_err(got.permute(0, 3, 1, 2), xin.grad)
    print(f"conv data gradient [{mode}]: {e:.2e}")
    assert e < 5e-3


def test_conv_out_bwd_vs_autograd():
    B, C, H = 2, 320, 16
    x = h16(B, C, H, H, seed=1)
    w = h16(4, C, 3, 3, seed=2, scale=(9 * C) ** -0.5)
    xin = x.float().requires_grad_(True)
    de = torch.randn(B, 4, H, H, generator=torch.Generator().manual_seed(3)).to(DEV)
    F.conv2d(xin, w.float(), padding=1).backward(de)
    got = hip.conv_out_bwd(de.contiguous(), _nhwc(w))
    e = rel_err(got.permute(0, 3, 1, 2), xin.grad)
    print(f"conv_out_bwd: {e:.2e}")
    assert e < 2e-3


# ------------------------------------------------------------------------------------------------ objective, Adam
def test_nti_loss_grad_and_adam_vs_torch():
    g = torch.Generator().manual_seed(0)
    n = 4 * 16 * 16
    eu, ec, x, tgt = (torch.randn(n, generator=g).to(DEV) for _ in range(4))
    a_f, a_t, gs = 0.31, 0.42, 7.5
    coef = torch.tensor([a_f, a_t, gs, 0.0], device=DEV)
    d_eps, stats = torch.empty(n, device=DEV), torch.zeros(2, device=DEV)
    hip.nti_loss_grad(eu, ec, x, tgt, coef, d_eps, stats, 4.0)
    eur = eu.clone().requires_grad_(True)
    e = eur + gs * (ec - eur)
    rec = math.sqrt(a_t) * (x - math.sqrt(1 - a_f) * e) / math.sqrt(a_f) + math.sqrt(1 - a_t) * e
    loss = F.mse_loss(rec, tgt)
    loss.backward()
    assert abs(stats[0].item() - loss.item()) <= 1e-5 * loss.item()
    assert abs(d_eps.abs().max().item() - 4.0) < 1e-5
    assert rel_err(d_eps * stats[1], eur.grad) < 1e-5

    # Adam: five steps against torch.optim.Adam fed the same (fp16-stored, rescaled) gradients
    n = 77 * 64
    p0 = torch.randn(n, generator=g).to(DEV)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=7e-3)
    param, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    p16 = torch.empty(n, dtype=torch.float16, device=DEV)
    hyper = torch.tensor([7e-3, 0.9, 0.999, 1e-8], device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    st = torch.tensor([0.0, 3e-4], device=DEV)
    for it in range(5):
        g16 = (torch.randn(n, generator=g) * 2.0).half().to(DEV)
        hip.nti_adam(param, m, v, g16, st, hyper, step, p16)
        p_ref.grad = g16.float() * 3e-4
        opt.step()
    assert step.item() == 5
    assert (param - p_ref.detach()).abs().max().item() < 1e-6
    assert torch.equal(p16, param.half())


# ------------------------------------------------------------------------------------------------ whole UNet
@pytest.fixture(scope="module")
def tiny():
    return StableDiffusionPipeline.from_pretrained("synthetic:tiny", keep_state_dict=True)


@pytest.fixture(scope="module")
def small():
    return StableDiffusionPipeline.from_pretrained("synthetic:small", keep_state_dict=True)


def _ctx_grad_case(pipe, B, seed=0):
    cfg = pipe.cfg
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4, cfg.sample_size, cfg.sample_size, generator=g)
    ctx = torch.randn(B, 77, cfg.cross_attention_dim, generator=g) * 0.1
    de = torch.randn(B, 4, cfg.sample_size, cfg.sample_size, generator=g)
    de = de / de.abs().max()
    t = 601
    ctx_ref = ctx.half().float().requires_grad_(True)      # the HIP path sees the fp16-rounded context
    eps_ref = unet_ref.unet_forward(pipe._state_dict, cfg, x, t, ctx_ref)
    (eps_ref * de).sum().backward()
    unet = pipe.unet
    adj = UNetAdjoint(unet)
    adj.taps = {}
    temb = unet.time_rows(torch.tensor([float(t)], device=DEV))
    eps = adj.forward(x.to(DEV), temb, ctx.half().to(DEV))
    got = adj.backward(de.to(DEV).contiguous())
    torch.cuda.synchronize()
    return rel_err(eps, eps_ref.detach()), rel_err(got, ctx_ref.grad), adj.taps, ctx_ref.grad.abs().max().item()


@pytest.mark.parametrize("B", [1, 2])
def test_unet_context_gradient_tiny(tiny, B):
    e_fwd, e_grad, taps, gmax = _ctx_grad_case(tiny, B)
    print(f"tiny B={B}: forward(tape) {e_fwd:.2e}; d/d ctx {e_grad:.2e}; max|grad| {gmax:.2e}; taps {taps}")
    assert e_fwd < 2e-2
    assert e_grad < 5e-2


def test_unet_context_gradient_small(small):
    e_fwd, e_grad, taps, gmax = _ctx_grad_case(small, 1, seed=3)
    print(f"small: forward(tape) {e_fwd:.2e}; d/d ctx {e_grad:.2e}; max|grad| {gmax:.2e}; taps {taps}")
    assert e_fwd < 2e-2
    assert e_grad < 5e-2


# ------------------------------------------------------------------------------------------------ the NTI loop
def _nti_case(pipe, steps, inner, seed=0):
    cfg = pipe.cfg
    pipe.scheduler.set_timesteps(steps)
    g = torch.Generator().manual_seed(seed)
    ctx = torch.randn(2, 77, cfg.cross_attention_dim, generator=g) * 0.1
    x0 = torch.randn(1, 4, cfg.sample_size, cfg.sample_size, generator=g)
    sched = p2p_ref.DDIMRef(num_inference_steps=steps)
    lat_ref = p2p_ref.ddim_inversion_loop(pipe._state_dict, cfg, ctx[1:], x0, sched)
    return ctx, lat_ref, sched


def test_nti_loop_vs_oracle(tiny):
    """epsilon = 0 disables the early stop, so both sides take exactly `inner` Adam steps per timestep.  Every timestep is
    judged from IDENTICAL starting points: the oracle's loop body (`oracle.p2p_ref.null_optimization(start=i, cur0=...)`) is
    entered with the product's own latent and embedding of that moment, so nothing accumulates from timestep to timestep and
    the same strict criterion holds at all of them (round 2 applied it to the first timestep only and a bulk percentile to
    the others).

    Adam moves every element by ~lr per step whatever |grad| is, so an element whose gradient sits at the fp16 noise floor
    (the fp16 pass resolves the gradient to 3-5e-3 of its largest element, `test_unet_context_gradient_*`) can legitimately
    take the other sign; elements with a SIGNIFICANT gradient cannot.  The oracle reports its gradients, and each timestep
    is judged on the elements whose first gradient is >= 5 % of the largest: after the three Adam steps >= 99 % of them must
    have moved the oracle's way and >= 98 % must sit within 10 % of the movement -- a wrong-sign or mis-scaled gradient
    flips half or all of them; no element may be further than a full step the other way (2.1 movements).  The fp32-storage
    modes meet an element-by-element bound of 1e-2 instead (tests/test_gpu_grad_f32.py)."""
    steps, inner, outer, gs = 4, 3, 3, 7.5
    ctx, lat_ref, sched = _nti_case(tiny, steps, inner)
    opt = NullTextOptimizer(tiny, ctx[1:], gs, tuple(lat_ref[-1].shape[-2:]))
    opt.begin([l.to(DEV) for l in lat_ref], ctx[:1])
    for i in range(outer):
        lat_i, u_i = opt.lat.clone().cpu(), opt.param.clone().cpu()
        opt.outer_begin(i)
        for j in range(inner):
            opt.inner_step()
            opt.inner_loss()
        opt.outer_end()
        a = opt.out[-1].cpu()
        trace = []
        b = p2p_ref.null_optimization(tiny._state_dict, tiny.cfg, lat_ref, torch.cat([u_i, ctx[1:]]), sched, num_inner_steps=inner,
                                      epsilon=0.0, guidance_scale=gs, num_outer=1, start=i, cur0=lat_i, grad_trace=trace)[0]
        g0 = trace[0][2][:1]
        strong = g0.abs() >= 0.05 * g0.abs().max()
        moved = (b - u_i).abs().max().item()
        diff = (a - b).abs()
        same_way = (torch.sign(a - u_i)[strong] == torch.sign(b - u_i)[strong]).float().mean().item()
        near = (diff[strong] <= 0.1 * moved).float().mean().item()
        frac_close = (diff <= 0.1 * moved).float().mean().item()
        print(f"NTI timestep {i}: moved {moved:.3e}, max diff {diff.max().item():.3e}; {int(strong.sum())} elements with a significant "
              f"first gradient: {same_way:.4f} moved the oracle's way, {near:.4f} within 10 % of the movement; all elements within "
              f"10 %: {frac_close:.4f}")
        assert strong.sum() > 50
        assert same_way >= 0.99 and near >= 0.98
        assert diff.max().item() <= 2.1 * moved
    opt.release()
    assert opt.inner_steps_run == [inner] * outer


def test_nti_reduces_reconstruction_error(tiny):
    """The purpose of the optimisation (nti.py:26): the CFG step with the optimised embedding lands closer to the
    inversion trajectory than with the original one.  Also exercises the class the CLI uses, early stop active."""
    steps, inner, gs = 4, 5, 7.5
    ctx, lat_ref, sched = _nti_case(tiny, steps, inner, seed=1)
    inv = NTI()
    lats = [l.to(DEV) for l in lat_ref]
    emb = inv.null_optimization(tiny, lats, ctx.to(DEV), inner, 1e-5, gs)
    assert len(emb) == steps and all(tuple(e.shape) == (1, 77, tiny.cfg.cross_attention_dim) for e in emb)
    assert all(1 <= n <= inner for n in inv.inner_steps_run)

    def replay(uncond_list):
        cur = lat_ref[-1]
        err = []
        for i in range(steps):
            t = sched.timesteps[i]
            u = uncond_list[i]
            eps = unet_ref.unet_forward(tiny._state_dict, tiny.cfg, torch.cat([cur] * 2), t, torch.cat([u, ctx[1:]]))
            e_u, e_c = eps.chunk(2)
            cur = sched.step(e_u + gs * (e_c - e_u), int(t), cur)
            err.append(F.mse_loss(cur, lat_ref[len(lat_ref) - i - 2]).item())
        return err

    with torch.no_grad():
        base = replay([ctx[:1]] * steps)
        tuned = replay([e.cpu() for e in emb])
    print(f"reconstruction mse per step: plain {base}, null-text optimised {tuned}")
    assert tuned[-1] < base[-1]


def test_nti_many_in_flight_matches_single(tiny):
    """two images optimised concurrently (streams, interleaved graph replays) = each optimised alone, bit for bit"""
    from ief_amd.nti import run_many
    steps, inner, gs = 4, 4, 7.5
    cases = [_nti_case(tiny, steps, inner, seed=s) for s in (1, 2)]
    singles = []
    for ctx, lat_ref, _ in cases:
        opt = NullTextOptimizer(tiny, ctx[1:], gs, tuple(lat_ref[-1].shape[-2:]))
        singles.append((opt.run([l.to(DEV) for l in lat_ref], ctx[:1], inner, 1e-5), list(opt.inner_steps_run)))
        opt.release()
    opts = [NullTextOptimizer(tiny, ctx[1:], gs, tuple(lat_ref[-1].shape[-2:])) for ctx, lat_ref, _ in cases]
    outs = run_many(opts, [[l.to(DEV) for l in lat_ref] for _, lat_ref, _ in cases], [ctx[:1] for ctx, _, _ in cases], inner, 1e-5)
    for k, o in enumerate(opts):
        assert o.inner_steps_run == singles[k][1]
        for a, b in zip(outs[k], singles[k][0]):
            assert torch.equal(a, b)
        o.release()
