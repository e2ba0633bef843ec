"""Split-operand contractions on PRE-SPLIT PLANES (`csrc/gemm_x3p.hip`, `ief_amd.planes`) on a real MI355X.

The f16x3 mode's activations travel as two fp16 planes (hi = fp16(x), lo = fp16(x - hi)) written by their producers; the GEMM /
implicit-GEMM convolution stages both operands by LDS-DMA and runs three fp16 MFMAs per fragment pair.  The reference computes
these layers in fp32 (`/root/reference/p2p/model/register.py:33-54`, `/root/reference/pnp/model/register.py:139-175`).

Stated tolerances (every test prints what it measured):
    planes written by any producer            == split of the fp32 value, bit for bit
    single contractions vs fp64 on the host   <= 4e-6 of max |reference|  (the bound of tests/test_gpu_x3.py, unchanged)
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from ief_amd import hip, planes  # noqa: E402
from ief_amd.planes import Planes  # noqa: E402

XTOL = 4e-6


def f32(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def dev(t):
    return None if t is None else t.cuda()


def rel_err(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all()
    return ((got - ref).abs().max() / ref.abs().max()).item()


def ref_split(x):
    """the definition of the planes: two round-to-nearest conversions"""
    hi = x.half()
    lo = (x - hi.float()).half()
    return hi, lo


def assert_planes_equal_split(pl, x32):
    hi, lo = ref_split(x32.float().cpu())
    assert torch.equal(pl.hi.cpu(), hi), "hi plane differs from fp16(x)"
    assert torch.equal(pl.lo.cpu(), lo), "lo plane differs from fp16(x - hi)"


@pytest.fixture(autouse=True)
def _x3():
    with hip.f32_contraction("x3"):
        yield


def test_split_act_is_the_definition():
    x = torch.cat([f32(64, 320, seed=1), f32(64, 320, seed=2, scale=1e-4), f32(64, 320, seed=3, scale=1e4)])        # max |x| ~ 4.5e4 < 65504
    pl = planes.split(dev(x))
    assert_planes_equal_split(pl, x)
    assert (pl.to_f32().cpu() - x).abs().max() <= 2 ** -21 * x.abs().max()
    # a strided destination (column slice of a wider planes tensor)
    wide = Planes.empty(192, 960, device="cuda")
    wide.t.zero_()
    planes.split(dev(x), out=wide[:, 320:640])
    assert_planes_equal_split(wide[:, 320:640], x)
    assert wide.t[:, :, :320].abs().max() == 0 and wide.t[:, :, 640:].abs().max() == 0


def test_gemm_x3p_identity_asymmetric():
    """A = I with an asymmetric W of fp16-exact values catches a swapped row / column fragment map, a wrong swizzle and a dropped term"""
    a = torch.eye(160)
    w = torch.arange(320)[:, None] * 0.25 + torch.arange(160)[None, :] * 1.0 + 1.0 / 1024
    for tile in (1, 2, 3, 4, 5):
        out = planes.gemm(planes.split(dev(a)), dev(w), tile=tile)
        assert out.dtype == torch.float32 and torch.equal(out.cpu(), w.t().contiguous()), f"tile {tile}"


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 7, 8])
@pytest.mark.parametrize("M,N,K", [(128, 160, 32), (300, 320, 96), (16384, 320, 320), (77, 1280, 768), (4, 1280, 320),
                                   (1000, 480, 96), (4096, 2560, 320), (256, 1280, 5120)])
def test_gemm_x3p(M, N, K, tile):
    a, w = f32(M, K, seed=1), f32(N, K, seed=2, scale=K ** -0.5)
    bias, res = f32(N, seed=3, scale=0.1), f32(M, N, seed=4)
    ap = planes.split(dev(a))
    out, op = planes.gemm(ap, dev(w), bias=dev(bias), residual=dev(res), out_scale=0.5, out=True, out_planes=True, tile=tile)
    ref = (a.double() @ w.double().t() + bias.double() + res.double()) * 0.5
    e = rel_err(out, ref)
    old = hip.gemm(dev(a), dev(w), bias=dev(bias), residual=dev(res), out_scale=0.5)       # the in-kernel split (scale 4)
    print(f"gemm x3p t{tile} {M}x{N}x{K}: {e:.2e} vs fp64 (in-kernel split: {rel_err(old, ref):.2e})")
    assert e < XTOL
    assert_planes_equal_split(op, out)                       # the epilogue's planes ARE the split of its fp32 output
    # split-K: slabs summed in slab order by the reducer launch
    if K >= 256:
        o2 = planes.gemm(ap, dev(w), bias=dev(bias), residual=dev(res), out_scale=0.5, tile=tile, splits=4)
        assert rel_err(o2, ref) < XTOL
    # planes only; row vector; strided A and output views
    rv = f32(2, N, seed=5)
    if M % 2 == 0:
        o3 = planes.gemm(ap, dev(w), rowvec=dev(rv), rows_per_batch=M // 2, out=False, out_planes=True, tile=tile)
        ref3 = a.double() @ w.double().t() + rv.double().repeat_interleave(M // 2, 0)
        assert isinstance(o3, Planes) and rel_err(o3.to_f32(), ref3) < XTOL
    wide = planes.split(dev(f32(M, K + 64, seed=6)))
    owide = torch.zeros(M, N + 4, device="cuda")
    planes.gemm(wide[:, 32:32 + K], dev(w), out=owide[:, :N], tile=tile)
    assert rel_err(owide[:, :N], wide[:, 32:32 + K].to_f32().double().cpu() @ w.double().t()) < XTOL and owide[:, N:].abs().max() == 0


@pytest.mark.parametrize("tile", [0, 3, 8])           # 0: the plan table's choice; 3: 128 x 80 (one pair + an odd block per wave); 8: 256 x 320 (five pairs)
@pytest.mark.parametrize("M,Ch,K", [(4096, 1280, 320), (300, 640, 96), (333, 160, 64)])
def test_gemm_x3p_fused_geglu(M, Ch, K, tile):
    a, w, bias = f32(M, K, seed=1), f32(2 * Ch, K, seed=2, scale=K ** -0.5), f32(2 * Ch, seed=3, scale=0.1)
    out, op = planes.gemm(planes.split(dev(a)), dev(w), bias=dev(bias), geglu=True, out=True, out_planes=True, tile=tile)
    pre = (a.double() @ w.double().t() + bias.double()).reshape(M, Ch // 8, 2, 8)
    ref = (pre[:, :, 0] * F.gelu(pre[:, :, 1])).reshape(M, Ch)
    e = rel_err(out, ref)
    print(f"x3p fused GEGLU {M}x{2 * Ch}x{K}: {e:.2e}")
    assert e < XTOL
    assert_planes_equal_split(op, out)


def _conv_ref(x, w, bias, x2=None, stride=1, upsample=False, pad_hi_only=False):
    xin = x if x2 is None else torch.cat([x, x2], -1)
    xin = xin.permute(0, 3, 1, 2).double()
    if upsample:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    wt = w.permute(0, 3, 1, 2).double()
    if pad_hi_only:
        xin = F.pad(xin, (0, 1, 0, 1))
        y = F.conv2d(xin, wt, None if bias is None else bias.double(), stride=stride)
    else:
        y = F.conv2d(xin, wt, None if bias is None else bias.double(), stride=stride, padding=1)
    return y.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 11])
@pytest.mark.parametrize("case", ["plain", "concat", "stride2", "upsample", "pad_hi", "odd"])
def test_conv_x3p(case, tile):
    """tile 11 = the halo form (super-tile resident across the nine taps; 12 with the fused nearest-2x): plain / concat / odd sizes"""
    if tile == 11 and case in ("stride2", "pad_hi"):
        pytest.skip("the halo form covers stride 1 / pad 1 only")
    B, H, W, C1, Cout = 2, 16, 16, 64, 160
    kw, C2 = {}, 0
    if tile == 11 and case == "upsample":
        tile = 12
    if case == "concat":
        C2 = 32
    elif case == "stride2":
        kw["stride"] = 2
    elif case == "upsample":
        kw["upsample"] = True
        H = W = 8
    elif case == "pad_hi":
        kw.update(stride=2, pad_hi_only=True)
    elif case == "odd":
        B, H, W = 3, 9, 7
    x, w, bias = f32(B, H, W, C1, seed=1), f32(Cout, 3, 3, C1 + C2, seed=2, scale=(9 * (C1 + C2)) ** -0.5), f32(Cout, seed=3, scale=0.1)
    x2 = f32(B, H, W, C2, seed=4) if C2 else None
    ref = _conv_ref(x, w, bias, x2, **kw)
    rv = f32(B, Cout, seed=5)
    res = f32(*ref.shape, seed=6)
    out, op = planes.conv3x3(planes.split(dev(x)), dev(w), dev(bias), x2=None if x2 is None else planes.split(dev(x2)),
                             rowvec=dev(rv), residual=dev(res), out=True, out_planes=True, tile=tile, **kw)
    full = ref + rv.double()[:, None, None, :] + res.double()
    e = rel_err(out, full)
    print(f"conv x3p {case} t{tile}: {e:.2e}")
    assert e < XTOL
    assert_planes_equal_split(op, out)
    if case in ("plain", "concat"):
        o2 = planes.conv3x3(planes.split(dev(x)), dev(w), dev(bias), x2=None if x2 is None else planes.split(dev(x2)), tile=tile, splits=3)
        assert rel_err(o2, ref) < XTOL


@pytest.mark.parametrize("tile", [1, 3])
def test_conv_x3p_fused_shortcut(tile):
    """ResnetBlock2D tail with a channel-changing shortcut in ONE launch: conv3x3(h) + conv1x1([x | skip])"""
    B, H, W, Cout, Cx, Cs = 2, 16, 16, 160, 64, 96
    h, x, skip = f32(B, H, W, Cout, seed=1), f32(B, H, W, Cx, seed=2), f32(B, H, W, Cs, seed=3)
    w2, ws, bias = f32(Cout, 3, 3, Cout, seed=4, scale=0.03), f32(Cout, Cx + Cs, seed=5, scale=0.08), f32(Cout, seed=6, scale=0.1)
    wf = torch.cat([w2.reshape(Cout, -1), ws], 1).contiguous()
    out = planes.conv3x3(planes.split(dev(h)), dev(wf), dev(bias), extra=(planes.split(dev(x)), planes.split(dev(skip))), tile=tile)
    ref = _conv_ref(h, w2, bias) + torch.cat([x, skip], -1).double() @ ws.double().t()
    e = rel_err(out, ref)
    print(f"conv x3p + fused 1x1 shortcut t{tile}: {e:.2e}")
    assert e < XTOL
    out1 = planes.conv3x3(planes.split(dev(h)), dev(torch.cat([w2.reshape(Cout, -1), ws[:, :Cx]], 1).contiguous()), dev(bias),
                          extra=(planes.split(dev(x)), None), tile=tile)
    assert rel_err(out1, _conv_ref(h, w2, bias) + x.double() @ ws[:, :Cx].double().t()) < XTOL


@pytest.mark.parametrize("M,N,K", [(256, 64, 64), (1000, 192, 128), (4096, 1024, 128), (77, 256, 64)])
def test_gemm_x3p_widths_of_64(M, N, K):
    """tile 6 (128 x 64): the widths of the TINY / SMALLXL test geometries and of the VAE (multiples of 64, not of 80)"""
    a, w, bias = f32(M, K, seed=1), f32(N, K, seed=2, scale=K ** -0.5), f32(N, seed=3, scale=0.1)
    out = planes.gemm(planes.split(dev(a)), dev(w), bias=dev(bias))
    e = rel_err(out, a.double() @ w.double().t() + bias.double())
    print(f"gemm x3p (auto tile) {M}x{N}x{K}: {e:.2e}")
    assert e < XTOL
    x, wc = f32(2, 16, 16, 64, seed=4), f32(128, 3, 3, 64, seed=5, scale=0.04)
    assert rel_err(planes.conv3x3(planes.split(dev(x)), dev(wc)), _conv_ref(x, wc, None)) < XTOL


@pytest.mark.parametrize("B,H,W,C1,C2,Cout,ups,splits", [(4, 64, 64, 320, 0, 320, False, 1), (2, 32, 32, 640, 320, 640, False, 2),
                                                          (4, 16, 16, 1280, 0, 1280, False, 4), (3, 8, 8, 1280, 0, 1280, False, 8),
                                                          (2, 24, 40, 64, 32, 80, False, 1), (2, 32, 32, 640, 0, 640, True, 1),
                                                          (4, 8, 8, 1280, 0, 1280, True, 4), (1, 64, 64, 320, 0, 320, True, 1)])
def test_conv_halo_x3p_real_shapes(B, H, W, C1, C2, Cout, ups, splits):
    """the halo form at the UNet's real geometries (tile edges inside images, rows of 8 .. 128 pixels, split-K over channel blocks)
    against the implicit GEMM of the same mode (both <= 4e-6 of fp64 on the small cases above) and fp64 on one image"""
    x, w = f32(B, H, W, C1, seed=1), f32(Cout, 3, 3, C1 + C2, seed=2, scale=(9 * (C1 + C2)) ** -0.5)
    x2 = f32(B, H, W, C2, seed=3) if C2 else None
    bias = f32(Cout, seed=4, scale=0.1)
    xp, x2p = planes.split(dev(x)), (planes.split(dev(x2)) if C2 else None)
    halo = planes.conv3x3(xp, dev(w), dev(bias), x2=x2p, upsample=ups, tile=12 if ups else 11, splits=splits)
    ig = planes.conv3x3(xp, dev(w), dev(bias), x2=x2p, upsample=ups, tile=1)
    e = rel_err(halo, ig)
    ref = _conv_ref(x[:1], w, bias, None if x2 is None else x2[:1], upsample=ups)
    e64 = rel_err(halo[:1], ref)
    print(f"halo x3p {B}x{H}x{W} {C1}+{C2}->{Cout} ups={ups} s{splits}: {e:.2e} vs implicit GEMM, {e64:.2e} vs fp64")
    assert e < XTOL and e64 < XTOL


@pytest.mark.parametrize("B,heads,N,L,d", [(2, 8, 1024, 1024, 40), (4, 8, 256, 256, 80), (2, 2, 200, 144, 64), (1, 3, 77, 333, 40),
                                            (1, 5, 4096, 4096, 64), (2, 8, 4096, 4096, 40), (1, 1, 130, 64, 80)])
def test_attention_planes_in(B, heads, N, L, d):
    """`attn_flash_x3p_kernel`: q / k / v as operand planes (column slices of one q|k|v planes tensor), K / V tiles by LDS-DMA,
    unpadded 80 / 128 / 160-byte LDS rows (swizzled for d = 64), key counts that are not multiples of the tile, batch-row
    indirection (P2P self-replace / MasaCtrl / PnP sources) -- against fp64; the planes it writes == split of its fp32 output"""
    C = heads * d
    q, k, v = f32(B, N, C, seed=1), f32(B, L, C, seed=2, scale=1.5), f32(B, L, C, seed=3)
    if N == L:
        qkv = planes.split(dev(torch.cat([q, k, v], -1)))
        qp, kp, vp = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    else:
        qp, kp, vp = planes.split(dev(q)), planes.split(dev(k)), planes.split(dev(v))
    src = torch.arange(B, dtype=torch.int32)
    ks = src.clone()
    ks[-1] = 0
    out = planes.attn_flash(qp, kp, vp, heads, d ** -0.5, k_src=dev(ks), v_src=dev(ks), out_planes=False)
    op = planes.attn_flash(qp, kp, vp, heads, d ** -0.5, k_src=dev(ks), v_src=dev(ks))
    sub = slice(0, min(N, 512))
    qh = q.double()[:, sub].reshape(B, -1, heads, d).permute(0, 2, 1, 3)
    kh = k.double()[ks.long()].reshape(B, L, heads, d).permute(0, 2, 1, 3)
    vh = v.double()[ks.long()].reshape(B, L, heads, d).permute(0, 2, 1, 3)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1) @ vh).permute(0, 2, 1, 3).reshape(B, -1, C)
    e = rel_err(out[:, sub], ref)
    print(f"attention planes-in B={B} h={heads} N={N} L={L} d={d}: {e:.2e}")
    assert e < XTOL
    assert_planes_equal_split(op, out)


def test_sd15_shapes_x3p_full_size():
    """the step's largest layers at their real sizes (batch 4): K = C projection, FeedForward.net[0], the 64x64 convolution"""
    for (M, N, K) in [(16384, 320, 320), (16384, 2560, 320), (4096, 640, 2560)]:
        a, w = f32(M, K, seed=1), f32(N, K, seed=2, scale=K ** -0.5)
        out = planes.gemm(planes.split(dev(a)), dev(w))
        idx = torch.randint(0, M, (256,), generator=torch.Generator().manual_seed(3))
        ref = a[idx].double() @ w.double().t()
        e = ((out[idx.cuda()].double().cpu() - ref).abs().max() / ref.abs().max()).item()
        print(f"x3p {M}x{N}x{K}: {e:.2e}")
        assert e < XTOL
    x, w = f32(4, 64, 64, 320, seed=1), f32(320, 3, 3, 320, seed=2, scale=2880 ** -0.5)
    out = planes.conv3x3(planes.split(dev(x)), dev(w))
    ref = _conv_ref(x[:1], w, None)
    e = rel_err(out[:1], ref)
    print(f"x3p conv 64x64x320: {e:.2e}")
    assert e < XTOL


@pytest.mark.parametrize("M,C,N", [(4096, 320, 960), (1000, 640, 640), (300, 1280, 2560)])
def test_layernorm_folded_into_the_consumer_gemm(M, C, N):
    """BasicTransformerBlock's norm -> linear pair without the LayerNorm launch: the PRODUCER of the residual stream leaves (mean,
    M2) of every row per 80-column slice (Chan-mergeable: no E[x^2] - mean^2), the consumer runs on the planes of the RAW stream
    with W gamma and computes rstd (acc - mean colsum) + (b + W beta).  Against fp64 LayerNorm + linear, with a row offset 20x the
    row's spread (the cancellation case), and against the unfolded pair of launches; GEGLU epilogue included"""
    x, res = f32(M, C, seed=1) + 20.0 * f32(M, 1, seed=7), f32(M, C, seed=2)
    w0, b0 = f32(C, C, seed=3, scale=C ** -0.5), f32(C, seed=4, scale=0.1)
    gamma, beta = 1.0 + 0.3 * f32(C, seed=5), 0.2 * f32(C, seed=6)
    w1, b1 = f32(N, C, seed=8, scale=C ** -0.5), f32(N, seed=9, scale=0.1)
    h, hp, st = planes.gemm(planes.split(dev(x)), dev(w0), bias=dev(b0), residual=dev(res), out=True, out_planes=True, row_stats=True)
    assert_planes_equal_split(hp, h)
    hd = h.double().cpu()
    mean, var = hd.mean(1), hd.var(1, unbiased=False)
    got_mean = (st.buf[:, :, 0].double().cpu()).mean(1)
    assert (got_mean - mean).abs().max() < 1e-5 * hd.abs().max()
    ln = ((hd - mean[:, None]) / torch.sqrt(var[:, None] + 1e-5)) * gamma.double() + beta.double()
    ref = ln @ w1.double().t() + b1.double()
    f = planes.FoldedLN(dev(w1), dev(b1), dev(gamma), dev(beta), 1e-5)
    out = planes.gemm(hp, f.w, bias=f.bias, ln=(st, f.colsum, f.eps))
    two = planes.gemm(planes.layernorm(h, dev(gamma), dev(beta), 1e-5), dev(w1), bias=dev(b1))
    e, e2 = rel_err(out, ref), rel_err(two, ref)
    print(f"LayerNorm folded {M}x{C}->{N}: {e:.2e} vs fp64 (LayerNorm launch + GEMM: {e2:.2e})")
    assert e < 2e-5 and e2 < 2e-5
    if N % 16 == 0:
        pre = ref.reshape(M, N // 16, 2, 8)
        gref = (pre[:, :, 0] * F.gelu(pre[:, :, 1])).reshape(M, N // 2)
        g = planes.gemm(hp, f.w, bias=f.bias, geglu=True, ln=(st, f.colsum, f.eps))
        assert rel_err(g, gref) < 2e-5


# ----------------------------------------------------------------------------------------------- GroupNorm as a producer of planes
@pytest.mark.parametrize("B,H,W,C1,C2,groups,silu", [
    (4, 64, 64, 320, 0, 32, True),          # three row-streaming launches (the 64 x 64 level of the SD1.5 step)
    (2, 32, 32, 640, 640, 32, True),        # channel concat
    (2, 16, 16, 1280, 640, 32, True),       # one launch (small levels); 60 channels per group: group 21 straddles the two sources
    (1, 8, 8, 1280, 1280, 32, False),
    (3, 5, 7, 64, 0, 32, True),             # 2 channels per group, ragged image
])
def test_groupnorm_writes_planes(B, H, W, C1, C2, groups, silu):
    """`planes.groupnorm` (the three row-streaming launches, and whatever form is the default of the size) against fp64 on activations
    with a 20-sigma offset (the centred second moment must not cancel); the planes are the split of the fp32 output bit for bit.
    (Round 4 also built a cooperative one-launch form whose workgroups meet through arrival counters -- parity-green and 4x
    SLOWER: DESIGN.md section 3e; the slab-in-registers form is tested below.)"""
    x = f32(B, H, W, C1, seed=1) + 20.0
    x2 = f32(B, H, W, C2, seed=2) * 3.0 - 20.0 if C2 else None
    gamma, beta = f32(C1 + C2, seed=3) * 0.5 + 1.0, f32(C1 + C2, seed=4) * 0.3
    xin = x if x2 is None else torch.cat([x, x2], -1)
    ref = F.group_norm(xin.permute(0, 3, 1, 2).double(), groups, gamma.double(), beta.double(), 1e-5).permute(0, 2, 3, 1)
    if silu:
        ref = F.silu(ref)
    keep = planes.GN_REG_MAX_HW
    try:
        planes.GN_REG_MAX_HW = 0                                      # the three row-streaming launches
        pl, o32 = planes.groupnorm(dev(x), dev(gamma), dev(beta), groups, 1e-5, silu=silu, x2=dev(x2), out32=True)
        pl2 = planes.groupnorm(dev(x), dev(gamma), dev(beta), groups, 1e-5, silu=silu, x2=dev(x2))
    finally:
        planes.GN_REG_MAX_HW = keep
    pl3 = planes.groupnorm(dev(x), dev(gamma), dev(beta), groups, 1e-5, silu=silu, x2=dev(x2))      # the default form of this size
    e = rel_err(o32, ref)
    assert_planes_equal_split(pl, o32)
    e2, e3 = rel_err(pl2.hi.float() + pl2.lo.float(), ref), rel_err(pl3.hi.float() + pl3.lo.float(), ref)
    print(f"GroupNorm -> planes B={B} {H}x{W} C={C1}+{C2}: {e:.2e} (fp32 + planes), {e2:.2e} (planes only), {e3:.2e} (default form) vs fp64")
    assert e < 2e-6 and e2 < 2e-6 and e3 < 2e-6


@pytest.mark.parametrize("B,H,W,C1,C2,groups,silu", [
    (1, 64, 64, 320, 0, 32, True),          # batch 1 (DDIM inversion, null-text): 40 floats per thread
    (2, 32, 32, 640, 640, 32, True),        # channel concat
    (2, 16, 16, 1280, 640, 32, True),       # 60 channels per group: group 21 straddles the two sources
    (3, 5, 7, 64, 0, 32, True),             # 2 channels per group, ragged image
    (1, 24, 24, 1280, 0, 32, False),        # SD2.1 at 768 px
])
def test_groupnorm_slab_in_registers(B, H, W, C1, C2, groups, silu):
    """`ief_groupnorm_silu_reg` (one launch, one workgroup per (image, group), the input read once) against fp64 on activations
    with a 20-sigma offset; the planes are the split of the fp32 output bit for bit; the row-streaming / KS forms it replaces on
    small tensors give the same values to fp32 rounding"""
    lib = hip.load()
    assert lib.ief_groupnorm_reg_fits(C1, C2, H * W, groups) == 1
    assert lib.ief_groupnorm_reg_fits(640, 320, 64 * 64, 32) == 0     # 30 channels per group at 64 x 64: 122880 floats, does not fit
    x = f32(B, H, W, C1, seed=1) + 20.0
    x2 = f32(B, H, W, C2, seed=2) * 3.0 - 20.0 if C2 else None
    gamma, beta = f32(C1 + C2, seed=3) * 0.5 + 1.0, f32(C1 + C2, seed=4) * 0.3
    xin = x if x2 is None else torch.cat([x, x2], -1)
    ref = F.group_norm(xin.permute(0, 3, 1, 2).double(), groups, gamma.double(), beta.double(), 1e-5).permute(0, 2, 3, 1)
    if silu:
        ref = F.silu(ref)
    keep = planes.GN_REG_MAX_HW
    try:
        planes.GN_REG_MAX_HW = 1 << 30
        pl, o32 = planes.groupnorm(dev(x), dev(gamma), dev(beta), groups, 1e-5, silu=silu, x2=dev(x2), out32=True)
        pl2 = planes.groupnorm(dev(x), dev(gamma), dev(beta), groups, 1e-5, silu=silu, x2=dev(x2))
        planes.GN_REG_MAX_HW = 0
        pl3 = planes.groupnorm(dev(x), dev(gamma), dev(beta), groups, 1e-5, silu=silu, x2=dev(x2))
    finally:
        planes.GN_REG_MAX_HW = keep
    e, e_old = rel_err(o32, ref), rel_err(pl3.hi.float() + pl3.lo.float(), ref)
    assert_planes_equal_split(pl, o32)
    assert torch.equal(pl2.t, pl.t)
    print(f"GroupNorm, slab in registers B={B} {H}x{W} C={C1}+{C2}: {e:.2e} vs fp64 (the forms it replaces: {e_old:.2e})")
    assert e < 2e-6 and e_old < 2e-6
