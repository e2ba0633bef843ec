"""Pre-split operand planes of the f16x3 mode and the launches that consume / produce them (`csrc/gemm_x3p.hip`).

An fp32 activation x of the split-operand mode travels between kernels as TWO fp16 planes, `hi = fp16(x)`,
`lo = fp16(x - hi)` (4 bytes per element, like fp32): its producer writes them, so the GEMM that consumes it stages both
operands by LDS-DMA and spends no vector instruction on the split (include/ief_hip.h, ABI 4).  `Planes` is the host-side
handle of such a pair: one fp16 tensor `[2, *shape]` (index 0 = hi, 1 = lo), last dimension contiguous; column slices and
reshapes stay views.  There is no CPU path: `Planes.to_f32()` exists for tests and debugging only.
"""
import os
from ctypes import byref

import torch

from . import hip
from .hip import IefGemmX3pParams, _check, _dev32, _ptr, _stream, _Timed, _zeros

ACT_SCALE = 1.0          # activations: hi = fp16(x) — the fp16 range itself (65504), as on the fp16-storage path
W_SCALE = hip.X3_SCALE_W


class Planes:
    __slots__ = ("t",)

    def __init__(self, t):
        if not (isinstance(t, torch.Tensor) and t.dtype == torch.float16 and t.dim() >= 2 and t.shape[0] == 2 and t.stride(-1) == 1):
            raise TypeError("Planes: expected an fp16 tensor [2, ...] with a contiguous last dimension")
        if not t.is_cuda:
            raise TypeError("Planes: device tensor expected (no CPU path)")
        self.t = t

    @staticmethod
    def empty(*shape, device):
        return Planes(torch.empty(2, *shape, dtype=torch.float16, device=device))

    @property
    def shape(self):
        return self.t.shape[1:]

    @property
    def device(self):
        return self.t.device

    @property
    def hi(self):
        return self.t[0]

    @property
    def lo(self):
        return self.t[1]

    @property
    def plane(self):
        """element offset from a hi element to its lo element"""
        return self.t.stride(0)

    def dim(self):
        return self.t.dim() - 1

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        return Planes(self.t[(slice(None),) + idx])

    def reshape(self, *shape):
        return Planes(self.t.reshape(2, *shape))

    def is_contiguous(self):
        return self.t[0].is_contiguous()

    def to_f32(self):
        return self.t[0].float() + self.t[1].float()

    def rows_ld(self, name="planes"):
        """(rows, cols, ld) with the leading dims collapsing to rows of one stride"""
        h = self.t[0]
        cols = h.shape[-1]
        ld = h.stride(-2) if h.dim() >= 2 else cols
        rows = 1
        for s in h.shape[:-1]:
            rows *= s
        for d in range(h.dim() - 2):
            if h.shape[d] > 1 and h.stride(d) != h.stride(d + 1) * h.shape[d + 1]:
                raise ValueError(f"{name}: leading dimensions must collapse to rows of one stride")
        return rows, cols, ld


def split(x, out=None, scale=ACT_SCALE):
    """fp32 [..., C] (rows of one stride) -> Planes (the standalone splitter: producers normally write planes themselves)"""
    lib = hip.load()
    rows, C, ldx = hip._rows_ld32(x, "x")
    if out is None:
        out = Planes.empty(*x.shape, device=x.device)
    _, _, ldp = out.rows_ld("out")
    _check(lib.ief_x3_split_act(x.data_ptr(), out.t.data_ptr(), out.plane, rows, C, ldx, ldp, float(scale), _stream()), "ief_x3_split_act")
    return out


def weight_planes(w):
    """Planes view [N, K] of an fp32 weight's cached pre-split planes (scale 2^8)"""
    t = hip.x3_weight_planes(w, W_SCALE)
    if t is None:
        raise RuntimeError("weight planes missing while capturing: run one eager forward first")
    return t


# ---- plan: tile id and split-K per (M, N, K); tuned table first (tuned_plans_x3.json), heuristic otherwise
_PLAN_FILE = os.environ.get("IEF_PLAN_FILE_X3") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_plans_x3.json")
_plans = None
TILE_FORCE = int(os.environ.get("IEF_X3P_TILE", "0"))        # A/B runs: force one tile id


def _plan_table():
    global _plans
    if _plans is None:
        _plans = {}
        if os.path.exists(_PLAN_FILE) and os.environ.get("IEF_NO_PLAN_TABLE", "0") != "1":
            import json
            with open(_PLAN_FILE) as f:
                _plans = {k: tuple(v) for k, v in json.load(f).items()}
    return _plans


def save_plans(path=None):
    import json
    with open(path or _PLAN_FILE, "w") as f:
        json.dump({k: list(v) for k, v in sorted(_plan_table().items())}, f, indent=0)


_TILES = {1: (128, 160), 2: (128, 160), 3: (128, 80), 4: (256, 160), 5: (64, 160), 6: (128, 64), 7: (128, 160), 8: (256, 320),
          11: (256, 80), 12: (256, 80)}


def heuristic_plan(M, N, K, conv=False):
    nk = K // 32
    if N % 80 and N % 64 == 0:
        tile = 6
    elif N % 160:
        tile = 3
    elif M <= 64:
        tile = 5
    else:
        t1 = -(-M // 128) * (N // 160)
        tile = 4 if t1 >= 1024 and M % 256 == 0 else 1
    bm, bn = _TILES[tile]
    tiles = -(-M // bm) * -(-N // bn)
    splits = 1
    if tiles < 160 and nk >= 16:
        splits = max(1, min(-(-256 // tiles), nk // 8, 16))
    return tile, splits


def pick_plan(M, N, K, conv=False, variant=""):
    hit = _plan_table().get(f"{'conv' if conv else 'gemm'}|{M}|{N}|{K}{variant}")
    if hit is None:
        hit = heuristic_plan(M, N, K, conv)
    if TILE_FORCE and (TILE_FORCE < 10 or conv):
        return TILE_FORCE, hit[1]
    return hit[0], hit[1]


AUTOTUNE = os.environ.get("IEF_AUTOTUNE_X3", "0") == "1"      # tune unseen shapes on first (eager) use: tests/tune_plans_x3.py


def candidate_plans(M, N, K, conv=False, halo_ok=False, ncb=0, geglu=False):
    """(tile, splits) worth timing for one shape"""
    nk = K // 32
    out = []
    for t, (bm, bn) in _TILES.items():
        if t in (11, 12):
            continue
        if N % bn or (t == 6 and N % 80 == 0):
            continue
        tiles = -(-M // bm) * (N // bn)
        for sp in (1, 2, 3, 4, 6, 8, 12, 16):
            if sp > 1 and (geglu or nk // sp < 4 or tiles * sp > 2048 or tiles >= 512):
                continue
            if tiles * sp < 40 and sp < 16:
                continue
            out.append((t, sp))
    if conv and halo_ok:
        tiles = -(-M // 256) * (N // 80)
        for sp in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20):
            if sp > 1 and (sp > ncb // 2 or tiles * sp > 1024 or tiles >= 384):
                continue
            if tiles * sp < 40 and sp < 16:
                continue
            out.append((11, sp))
    return out


def _autotune(key, cands, run, w):
    """time run(tile, splits, weight copy) for every candidate (hipGraph of 20 launches over cache-cold weight copies)"""
    wc = hip._cold_copies(w)
    for x in wc:
        weight_planes(x)
    best = None
    for t, sp in cands:
        try:
            us = hip._time_graph(lambda i: run(t, sp, wc[i % len(wc)]))
        except (RuntimeError, ValueError):
            continue
        if best is None or us < best[0]:
            best = (us, t, sp)
    for x in wc[1:]:
        hip._x3_planes.pop((x.data_ptr(), tuple(x.shape), float(W_SCALE)), None)
    if best is not None:
        _plan_table()[key] = (best[1], best[2])
    return best


def supported(N, K):
    """shapes the planes kernels take (everything else stays on the in-kernel split of csrc/split_x3.hip)"""
    return (N % 80 == 0 or N % 64 == 0) and K % 32 == 0


def _out_args(p, M, No, out, out_planes, device, lead_shape):
    """fill Out / OutP of a launch; returns (fp32 tensor | None, Planes | None)"""
    o32 = op = None
    if out is True or isinstance(out, torch.Tensor) or (out is None and out_planes in (None, False)):
        o32 = out if isinstance(out, torch.Tensor) else torch.empty(*lead_shape, No, dtype=torch.float32, device=device)
        Mo, Nn, ldo = hip._rows_ld32(o32, "out")
        if (Mo, Nn) != (M, No):
            raise ValueError("x3p: out shape mismatch")
        p.Out, p.ldo = o32.data_ptr(), ldo
    if out_planes not in (None, False):
        op = out_planes if isinstance(out_planes, Planes) else Planes.empty(*lead_shape, No, device=device)
        Mp, Np, ldp = op.rows_ld("out_planes")
        if (Mp, Np) != (M, No):
            raise ValueError("x3p: out_planes shape mismatch")
        p.OutP, p.planeO, p.ldp = op.t.data_ptr(), op.plane, ldp
    return o32, op


def _ret(o32, op):
    if o32 is not None and op is not None:
        return o32, op
    return o32 if op is None else op


class RowStats:
    """(mean, M2) of every output row over each column slice of ONE launch (IefGemmX3pParams.rstat_out): what the next GEMM
    needs to fold the LayerNorm between them into its epilogue"""
    __slots__ = ("buf", "slots", "cnt")

    def __init__(self, buf, slots, cnt):
        self.buf, self.slots, self.cnt = buf, slots, cnt


class FoldedLN:
    """a linear with the LayerNorm that feeds it folded in (`/root/reference` computes norm(x) W^T + b with diffusers'
    BasicTransformerBlock): weight W gamma, bias b + W beta, and colsum[n] = sum_k (W gamma)[n][k] AS THE OPERAND PLANES HOLD IT, so
    that  LN(x) W^T + b = rstd (x (W gamma)^T - mean colsum) + (b + W beta)  is exact up to the products' own rounding"""
    __slots__ = ("w", "bias", "colsum", "eps")

    def __init__(self, w, bias, gamma, beta, eps):
        wf = (w * gamma[None, :]).contiguous()
        self.w = wf
        b = w @ beta
        self.bias = (b if bias is None else b + bias).contiguous()
        pl = weight_planes(wf)
        self.colsum = ((pl[0].float().sum(1) + pl[1].float().sum(1)) / W_SCALE).contiguous()
        self.eps = eps


LN_FOLD = os.environ.get("IEF_X3P_FOLD_LN", "0") == "1"      # 1: fold the transformer LayerNorms into their consumer GEMMs.  Measured
# on the SD1.5 batch-4 step (same box): 14.86 ms folded vs 14.16 ms with the 48 LayerNorm launches -- every producer of the residual
# stream then writes fp32 AND planes AND row statistics, its consumers lose their split-K plans: the launches saved cost less


def gemm(a, w, bias=None, residual=None, rowvec=None, rows_per_batch=0, out=None, out_planes=None, out_scale=1.0,
         geglu=False, tile=0, splits=0, row_stats=False, ln=None):
    """(a . w^T + bias + rowvec + residual) * out_scale with a: Planes [..., K], w: fp32 weight [N, K] (its planes are cached).
    out: an fp32 tensor to write / True (allocate one) / None (allocate one unless planes are asked for) / False (none);
    out_planes: True / a Planes to receive the result's planes.  Returns the fp32 tensor, the Planes, or (fp32, Planes).
    row_stats=True: a RowStats of the output rows is appended to the result; ln=(RowStats of a's rows, colsum, eps): the launch
    folds the LayerNorm of its input (w / bias are then the folded ones, `FoldedLN`)."""
    lib = hip.load()
    if not isinstance(a, Planes):
        raise TypeError("planes.gemm: a must be Planes")
    M, K, lda = a.rows_ld("a")
    N = w.shape[0]
    if w.dim() != 2 or w.shape[1] != K:
        raise ValueError(f"planes.gemm: K mismatch {tuple(w.shape)} vs {K}")
    wp = weight_planes(w)
    p = IefGemmX3pParams()
    p.A, p.planeA, p.lda = a.t.data_ptr(), a.plane, lda
    p.W, p.planeW, p.ldw = wp.data_ptr(), wp.stride(0), K
    No = N // 2 if geglu else N
    o32, op = _out_args(p, M, No, out, out_planes, a.device, a.shape[:-1])
    p.bias = _ptr(_dev32(bias, "bias")) if bias is not None else None
    if rowvec is not None:
        p.rowvec, p.rows_per_batch = _dev32(rowvec, "rowvec").data_ptr(), rows_per_batch
    if residual is not None:
        Mr, Nr, ldr = hip._rows_ld32(residual, "residual")
        if (Mr, Nr) != (M, N):
            raise ValueError("planes.gemm: residual shape mismatch")
        p.residual, p.ldr = residual.data_ptr(), ldr
    p.M, p.N, p.K = M, N, K
    p.out_scale, p.inv_scale = out_scale, 1.0 / (ACT_SCALE * W_SCALE)
    p.geglu, p.zeros = 1 if geglu else 0, _zeros(a.device)
    if tile:
        p.tile, p.splits = tile, max(1, splits)
    else:
        key = f"gemm|{M}|{N}|{K}" + ("|g" if geglu else "")
        if AUTOTUNE and key not in _plan_table() and not hip._capturing() and hip._prof is None:
            _autotune(key, candidate_plans(M, N, K, geglu=geglu),
                      lambda t, sp, wi: gemm(a, wi, bias=bias, residual=residual, rowvec=rowvec, rows_per_batch=rows_per_batch, out=out,
                                             out_planes=out_planes, out_scale=out_scale, geglu=geglu, tile=t, splits=sp), w)
        p.tile, p.splits = pick_plan(M, N, K, variant="|g" if geglu else "")
    if geglu or row_stats or ln is not None:
        p.splits = 1
    rs = None
    if row_stats:
        wn = lib.ief_gemm_x3p_tile_wn(p.tile)
        if N % wn:
            raise ValueError("planes.gemm: row statistics need N to be a multiple of the tile's wave width")
        rs = RowStats(torch.empty(M, N // wn, 2, dtype=torch.float32, device=a.device), N // wn, wn)
        p.rstat_out = rs.buf.data_ptr()
    if ln is not None:
        st, colsum, eps = ln
        if st.buf.shape[0] != M or st.slots * st.cnt != K or _dev32(colsum, "colsum").numel() != N:
            raise ValueError("planes.gemm: ln statistics / colsum do not match this launch")
        p.rstat_in, p.rstat_slots, p.rstat_cnt, p.colsum, p.ln_eps = st.buf.data_ptr(), st.slots, st.cnt, colsum.data_ptr(), eps
    ws = None
    if p.splits > 1:
        ws = torch.empty(p.splits * M * N, dtype=torch.float32, device=a.device)       # noqa: F841 (alive until queued)
        p.ws = ws.data_ptr()
    nbytes = 4.0 * (M * K + N * K + M * No * ((1 if o32 is not None else 0) + (1 if op is not None else 0)) + (M * N if residual is not None else 0))
    kn = f"igemm_x3p_kernel<false> t{p.tile}" + (f" {M}x{N}x{K} s{p.splits}" if hip.PROF_SHAPES else "")
    bm, bn = _TILES[p.tile]
    hip.note_staged(kn, 4.0 * K * (bm + bn) * -(-M // bm) * -(-N // bn))       # every output tile stages (bm + bn) rows x K x two fp16 planes
    with _Timed(kn, 2.0 * M * N * K, nbytes):
        _check(lib.ief_gemm_x3p(byref(p), _stream()), "ief_gemm_x3p")
    r = _ret(o32, op)
    if row_stats:
        return (r + (rs,)) if isinstance(r, tuple) else (r, rs)
    return r


def conv3x3(x, w, bias=None, x2=None, stride=1, upsample=False, rowvec=None, residual=None, out=None, out_planes=None,
            extra=None, pad_hi_only=False, tile=0, splits=0):
    """3x3 / pad 1 convolution over NHWC Planes x [B,H,W,C1] (+ x2 channel concat); w fp32 [Cout,3,3,C1+C2] or fused
    [Cout, 9 (C1+C2) + CE1 + CE2] with extra=(Planes e1, Planes e2 | None); residual fp32; outputs as `gemm`"""
    lib = hip.load()
    if not isinstance(x, Planes) or (x2 is not None and not isinstance(x2, Planes)):
        raise TypeError("planes.conv3x3: x / x2 must be Planes")
    if not x.is_contiguous() or (x2 is not None and not x2.is_contiguous()) or not w.is_contiguous():
        raise ValueError("planes.conv3x3: x, x2, w must be contiguous")
    B, Hp, Wp, C1 = x.shape
    C2 = 0 if x2 is None else x2.shape[-1]
    Cout = w.shape[0]
    e1 = e2 = None
    CE1 = CE2 = 0
    if extra is not None:
        e1, e2 = extra
        CE1 = e1.shape[-1]
        CE2 = 0 if e2 is None else e2.shape[-1]
        for e in (e1, e2):
            if e is not None and (not isinstance(e, Planes) or not e.is_contiguous() or tuple(e.shape[:3]) != (B, Hp, Wp)):
                raise ValueError("planes.conv3x3: extra sources must be contiguous NHWC Planes at the output resolution")
    K = 9 * (C1 + C2) + CE1 + CE2
    if w.numel() != Cout * K:
        raise ValueError(f"planes.conv3x3: weight {tuple(w.shape)} does not match K = {K}")
    H, Wd = (Hp * 2, Wp * 2) if upsample else (Hp, Wp)
    pad_total = 1 if pad_hi_only else 2
    Ho, Wo = (H + pad_total - 3) // stride + 1, (Wd + pad_total - 3) // stride + 1
    M = B * Ho * Wo
    wp = weight_planes(w)
    p = IefGemmX3pParams()
    p.A, p.planeA = x.t.data_ptr(), x.plane
    if x2 is not None:
        p.A2, p.planeA2 = x2.t.data_ptr(), x2.plane
    if e1 is not None:
        p.E1, p.planeE1 = e1.t.data_ptr(), e1.plane
    if e2 is not None:
        p.E2, p.planeE2 = e2.t.data_ptr(), e2.plane
    p.W, p.planeW, p.ldw = wp.data_ptr(), wp.stride(0), K
    o32, op = _out_args(p, M, Cout, out, out_planes, x.device, (B, Ho, Wo))
    p.bias = _ptr(_dev32(bias, "bias")) if bias is not None else None
    if rowvec is not None:
        _dev32(rowvec, "rowvec")
        if rowvec.dim() != 2 or rowvec.shape[1] != Cout or rowvec.shape[0] not in (1, B):
            raise ValueError("planes.conv3x3: rowvec must be [B, Cout] or [1, Cout]")
        p.rowvec = rowvec.data_ptr()
        p.rows_per_batch = Ho * Wo if rowvec.shape[0] == B and B > 1 else B * Ho * Wo
    if residual is not None:
        if tuple(hip._act32(residual, "residual").shape) != (B, Ho, Wo, Cout) or not residual.is_contiguous():
            raise ValueError("planes.conv3x3: residual must be contiguous fp32 [B, Ho, Wo, Cout]")
        p.residual, p.ldr = residual.data_ptr(), Cout
    p.M, p.N, p.K = M, Cout, K
    p.conv, p.H, p.Wd, p.C1, p.C2, p.Ho, p.Wo = 1, H, Wd, C1, C2, Ho, Wo
    p.stride, p.ups, p.batch_images, p.pad_hi_only = stride, 1 if upsample else 0, B, 1 if pad_hi_only else 0
    p.CE1, p.CE2 = CE1, CE2
    p.out_scale, p.inv_scale, p.zeros = 1.0, 1.0 / (ACT_SCALE * W_SCALE), _zeros(x.device)
    # the halo form (input super-tile resident in LDS across the nine taps): plain 3x3 / stride 1 / pad 1 on rows of <= 64
    # pixels, or the fused nearest-2x on rows of <= 128 pixels with whole output rows per 256-pixel tile
    halo_ok = HALO and stride == 1 and not pad_hi_only and extra is None and Hp >= 2 and Cout % 80 == 0 and (
        (not upsample and 2 <= Wd <= 64) or (upsample and Wd <= 128 and (H * Wd) % 256 == 0 and 256 % Wd == 0))
    if tile:
        p.tile, p.splits = tile, max(1, splits)
    else:
        variant = ("|u" if upsample else "") + (f"|s{stride}" if stride != 1 else "") + ("|e" if extra is not None else "")
        key = f"conv|{M}|{Cout}|{K}{variant}"
        if AUTOTUNE and key not in _plan_table() and not hip._capturing() and hip._prof is None:
            _autotune(key, candidate_plans(M, Cout, K, conv=True, halo_ok=halo_ok, ncb=(C1 + C2) // 32),
                      lambda t, sp, wi: conv3x3(x, wi, bias, x2=x2, stride=stride, upsample=upsample, rowvec=rowvec, residual=residual,
                                                out=out, out_planes=out_planes, extra=extra, pad_hi_only=pad_hi_only, tile=t, splits=sp), w)
        hit = _plan_table().get(key)
        if hit is not None and (hit[0] not in (11, 12) or halo_ok):
            p.tile, p.splits = hit[0], hit[1]
        elif halo_ok:
            tiles, ncb = -(-M // 256) * (Cout // 80), (C1 + C2) // 32
            sp = 1
            while tiles * sp < 192 and sp * 2 <= ncb // 2 and sp < 16:       # cut K (whole channel blocks) until ~256 workgroups exist
                sp *= 2
            p.tile, p.splits = 11, sp
        else:
            p.tile, p.splits = heuristic_plan(M, Cout, K, conv=True)        # (the table is keyed WITH the variant: no plain-key fallback)
            if TILE_FORCE and TILE_FORCE < 10:
                p.tile = TILE_FORCE
    if p.tile in (11, 12):
        if not halo_ok:
            raise ValueError("planes.conv3x3: this geometry has no halo form (tile 11 / 12)")
        p.tile = 12 if upsample else 11
        p.splits = min(p.splits, (C1 + C2) // 32)
    ws = None
    if p.splits > 1:
        ws = torch.empty(p.splits * M * Cout, dtype=torch.float32, device=x.device)     # noqa: F841
        p.ws = ws.data_ptr()
    nbytes = 4.0 * (B * Hp * Wp * (C1 + C2) + M * (CE1 + CE2) + Cout * K + M * Cout * ((1 if o32 is not None else 0) + (1 if op is not None else 0) + (1 if residual is not None else 0)))
    kn = f"conv3x3_halo_x3p_kernel<{'true' if upsample else 'false'}>" if p.tile in (11, 12) else f"igemm_x3p_kernel<true> t{p.tile}"
    kn += f" {M}x{Cout}x{K} s{p.splits}" if hip.PROF_SHAPES else ""
    if p.tile in (11, 12):      # halo form: per (256-pixel, 80-column) workgroup the weights of nine taps + ONE input super-tile per channel block
        hip.note_staged(kn, 4.0 * (C1 + C2) * (9 * 80 + 256 + 2 * (Wd + 1)) * -(-M // 256) * -(-Cout // 80))
    else:
        bm, bn = _TILES[p.tile]
        hip.note_staged(kn, 4.0 * K * (bm + bn) * -(-M // bm) * -(-Cout // bn))
    with _Timed(kn, 2.0 * M * Cout * K, nbytes):
        _check(lib.ief_gemm_x3p(byref(p), _stream()), "ief_gemm_x3p (conv)")
    return _ret(o32, op)


# ---- producers: the normalisations and the attention kernels write operand planes for the GEMM that follows them
GN_SMALL_ELEMS = int(os.environ.get("IEF_GN_SMALL_ELEMS", str(3 << 20)))      # GroupNorm inputs up to this many elements take one launch
GN_REG_MAX_HW = int(os.environ.get("IEF_GN_REG_MAX_HW", "1024"))      # pixels per image up to which the one-launch GroupNorm forms run (0: always the three row-streaming launches)
HALO = os.environ.get("IEF_X3P_HALO", "1") == "1"      # 0: every convolution on the implicit GEMM (A/B runs)
ENABLED = os.environ.get("IEF_X3P", "1") == "1"        # 0: the f16x3 model keeps the in-kernel split everywhere (A/B runs)


def groupnorm(x, gamma, beta, groups, eps, silu=False, x2=None, out32=False):
    """GroupNorm (+SiLU) over fp32 NHWC / tokens-major x (+ channel-concat x2) -> Planes [..., C1 + C2] (out32: also fp32)"""
    lib = hip.load()
    if not hip._act32(x, "x").is_contiguous() or (x2 is not None and not hip._act32(x2, "x2").is_contiguous()):
        raise ValueError("planes.groupnorm: inputs must be contiguous")
    B, C1 = x.shape[0], x.shape[-1]
    C2 = 0 if x2 is None else x2.shape[-1]
    HW = x.numel() // (B * C1)
    out = Planes.empty(*x.shape[:-1], C1 + C2, device=x.device)
    cpg = (C1 + C2) // groups
    if HW <= GN_REG_MAX_HW and lib.ief_groupnorm_reg_fits(C1, C2, HW, groups):
        # one workgroup per (image, group) keeps its slab in registers -- ONE launch that reads the input once.  Measured per call
        # (tests/bench_gn.py, batches 1 / 2 / 4): the fastest form wherever the image has at most 1024 pixels (batch 4: 32x32x640
        # 21.7 / 22.9 -> 14.9 us against the three row-streaming launches / the KS-workgroup launch, 32x32x1280 33.8 / 36.5 -> 21.6,
        # 16x16x2560 32.1 / 17.5 -> 13.0; batch 1: 32x32x640 17.2 / 14.0 -> 12.1); at 64x64 the three row-streaming launches win at every
        # batch (64x64x320: 16.8 / 18.2 / 24.2 us at batch 1 / 2 / 4 against 28.5 / 29.1 / 35.0: a workgroup per group reads 40-byte
        # slices of 1280-byte pixel rows), and the KS-workgroup launch never does there (38 / 45 / 69 us)
        o32 = torch.empty(*x.shape[:-1], C1 + C2, dtype=torch.float32, device=x.device) if out32 else None
        with _Timed("groupnorm_reg_kernel", 0.0, (12.0 if out32 else 8.0) * (x.numel() + (0 if x2 is None else x2.numel()))):
            _check(lib.ief_groupnorm_silu_reg(x.data_ptr(), _ptr(x2), C1, C2, _ptr(o32), out.t.data_ptr(), out.plane,
                                              _dev32(gamma, "gamma").data_ptr(), _dev32(beta, "beta").data_ptr(), B, HW, groups,
                                              eps, 1 if silu else 0, _stream()), "ief_groupnorm_silu_reg")
        return (out, o32) if out32 else out
    if not out32 and HW <= GN_REG_MAX_HW and B * HW * (C1 + C2) <= GN_SMALL_ELEMS and cpg % 2 == 0 and C1 % 2 == 0 and cpg // 2 <= 256:
        # small tensors (the 16x16 / 8x8 levels): ONE launch instead of three dispatch latencies
        with _Timed("groupnorm_f32_kernel", 0.0, 12.0 * (x.numel() + (0 if x2 is None else x2.numel()))):
            _check(lib.ief_groupnorm_silu_x3p_small(x.data_ptr(), _ptr(x2), C1, C2, out.t.data_ptr(), out.plane,
                                                    _dev32(gamma, "gamma").data_ptr(), _dev32(beta, "beta").data_ptr(), B, HW, groups,
                                                    eps, 1 if silu else 0, _stream()), "ief_groupnorm_silu_x3p_small")
        return out
    o32 = torch.empty(*x.shape[:-1], C1 + C2, dtype=torch.float32, device=x.device) if out32 else None
    nws = lib.ief_groupnorm_f32_ws_floats(B, HW, C1 + C2)
    ws = torch.empty(nws, dtype=torch.float32, device=x.device)
    with _Timed("gn3_f32 (stats + finalize + apply)", 0.0, 12.0 * (x.numel() + (0 if x2 is None else x2.numel()))):
        _check(lib.ief_groupnorm_silu_x3p_ws(x.data_ptr(), _ptr(x2), C1, C2, _ptr(o32), out.t.data_ptr(), out.plane,
                                             _dev32(gamma, "gamma").data_ptr(), _dev32(beta, "beta").data_ptr(), B, HW, groups, eps,
                                             1 if silu else 0, ws.data_ptr(), nws, _stream()), "ief_groupnorm_silu_x3p_ws")
    return (out, o32) if out32 else out


def layernorm(x, gamma, beta, eps=1e-5):
    """LayerNorm over the last dim of contiguous fp32 x -> Planes"""
    lib = hip.load()
    if not hip._act32(x, "x").is_contiguous():
        raise ValueError("planes.layernorm: x must be contiguous")
    C = x.shape[-1]
    out = Planes.empty(*x.shape, device=x.device)
    with _Timed("layernorm_f32_vec_kernel", 0.0, 8.0 * x.numel()):
        _check(lib.ief_layernorm_x3p(x.data_ptr(), out.t.data_ptr(), out.plane, _dev32(gamma, "gamma").data_ptr(),
                                     _dev32(beta, "beta").data_ptr(), x.numel() // C, C, eps, _stream()), "ief_layernorm_x3p")
    return out


def attn_out_args(p, B, N, C, device, out32=False):
    """fill OutP (and Out) of an IefAttnF32Params for an attention launch whose consumer is to_out's planes GEMM"""
    op = Planes.empty(B, N, C, device=device)
    p.OutP, p.planeO, p.sOPb, p.ldp = op.t.data_ptr(), op.plane, N * C, C
    o32 = None
    if out32:
        o32 = torch.empty(B, N, C, dtype=torch.float32, device=device)
        p.Out, p.sOb, p.ldo = o32.data_ptr(), N * C, C
    return op, o32


FLASH_PLANES = os.environ.get("IEF_X3P_FLASH", "1") == "1"       # 0: the fused attention always takes fp32 q / k / v (A/B runs)
FLASH_PLANES_DIMS = (40, 64, 80)


def attn_flash(q, k, v, heads, scale, q_src=None, k_src=None, v_src=None, out=None, out_planes=True):
    """fused attention on operand planes: q [B,N,h*d], k / v [B,L,h*d] as Planes (column slices of the q|k|v GEMM's output are
    fine); K / V tiles are staged by LDS-DMA, nothing is split in the kernel (`attn_flash_x3p_kernel`).  Returns Planes (for
    to_out's GEMM) or, with out_planes=False, fp32."""
    lib = hip.load()
    for t, nm in ((q, "q"), (k, "k"), (v, "v")):
        if not isinstance(t, Planes) or t.dim() != 3:
            raise TypeError(f"planes.attn_flash: {nm} must be Planes [B, rows, heads*d]")
    B, N, C = q.shape
    L, d = k.shape[1], C // heads
    if d not in FLASH_PLANES_DIMS:
        raise ValueError(f"planes.attn_flash: head dim {d} has no planes instantiation")
    p = hip.IefAttnF32Params()
    for t, nm in ((q, "Q"), (k, "K"), (v, "V")):
        h_ = t.hi
        if h_.stride(2) != 1 or (h_.shape[0] > 1 and h_.stride(0) != h_.shape[1] * h_.stride(1)):
            raise ValueError("planes.attn_flash: operands must be [B, rows, heads*d] with batch stride rows * ld")
        setattr(p, nm + "p", h_.data_ptr())
        setattr(p, "plane" + nm, t.plane)
    p.ldq, p.ldk, p.ldv = q.hi.stride(1), k.hi.stride(1), v.hi.stride(1)
    p.sQb, p.sKb, p.sVb = q.hi.stride(0), k.hi.stride(0), v.hi.stride(0)
    p.B, p.heads, p.N, p.L, p.d, p.scale = B, heads, N, L, d, scale
    p.q_src, p.k_src, p.v_src = _ptr(hip._devi32(q_src, "q_src")), _ptr(hip._devi32(k_src, "k_src")), _ptr(hip._devi32(v_src, "v_src"))
    p.x3, p.zeros = 1, _zeros(q.device)
    op = None
    if out_planes:
        op, _ = attn_out_args(p, B, N, C, q.device)
    else:
        if out is None:
            out = torch.empty(B, N, C, dtype=torch.float32, device=q.device)
        p.Out, p.sOb, p.ldo = hip._act32(out, "out").data_ptr(), out.stride(0), out.stride(1)
    with _Timed(f"attn_flash_x3p_kernel<{d}>", 4.0 * B * heads * N * L * d, 4.0 * B * heads * d * (2 * N + 2 * L)):
        _check(lib.ief_attn_flash_f32(byref(p), _stream()), "ief_attn_flash_f32 (planes)")
    return op if out_planes else out
