"""Kernel-level A/B on one MI355X: the planes GEMM / convolution (csrc/gemm_x3p.hip) against the in-kernel split
(csrc/split_x3.hip) on the SD1.5 batch-4 step's layer shapes.  Each timing = one hipGraph of 20 launches over cache-cold weight
copies (hip._time_graph), us per launch and algorithmic TFLOP/s (peak of the mode: 2500 / 3 = 833).

    python tests/bench_x3p.py [--tiles 1,2,3,4] [--conv] [--lin]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ief_amd  # noqa: E402,F401
from ief_amd import hip, planes  # noqa: E402

LIN = [(16384, 320, 320), (16384, 960, 320), (16384, 2560, 320), (16384, 320, 1280), (4096, 640, 640), (4096, 1920, 640),
       (4096, 5120, 640), (4096, 640, 2560), (1024, 1280, 1280), (1024, 3840, 1280), (1024, 10240, 1280), (1024, 1280, 5120),
       (256, 1280, 1280), (256, 10240, 1280)]
# (B, H, W, C1, C2, Cout, stride, upsample, CE1, CE2)
CONV = [(4, 64, 64, 320, 0, 320, 1, False, 0, 0), (4, 64, 64, 320, 320, 320, 1, False, 0, 0), (4, 64, 64, 640, 320, 320, 1, False, 0, 0),
        (4, 64, 64, 320, 0, 320, 1, False, 320, 320), (4, 32, 32, 640, 0, 640, 1, False, 0, 0), (4, 32, 32, 320, 0, 640, 1, False, 0, 0),
        (4, 32, 32, 640, 640, 640, 1, False, 0, 0), (4, 16, 16, 1280, 0, 1280, 1, False, 0, 0), (4, 16, 16, 1280, 1280, 1280, 1, False, 0, 0),
        (4, 8, 8, 1280, 0, 1280, 1, False, 0, 0), (4, 64, 64, 320, 0, 320, 2, False, 0, 0), (4, 32, 32, 640, 0, 640, 1, True, 0, 0)]


def rnd(*shape, scale=1.0):
    return torch.randn(*shape, device="cuda") * scale


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", default="1,2,3,4")
    ap.add_argument("--conv", action="store_true")
    ap.add_argument("--lin", action="store_true")
    ap.add_argument("--splits", default="1")
    ap.add_argument("--attn", action="store_true", help="the fused split-operand attention at the step's sequence lengths")
    ap.add_argument("--geglu", action="store_true", help="FeedForward.net[0] as the step runs it: bias + GEGLU epilogue, planes out only")
    args = ap.parse_args()
    tiles = [int(t) for t in args.tiles.split(",")]
    splits = [int(s) for s in args.splits.split(",")]
    only = args.conv or args.attn or args.lin or args.geglu
    do_lin, do_conv = args.lin or not only, args.conv or not only
    with hip.f32_contraction("x3"):
        if args.attn:
            for B, heads, N, d in [(4, 8, 4096, 40), (1, 8, 4096, 40), (2, 8, 4096, 40), (4, 8, 1024, 80), (4, 8, 256, 160), (4, 8, 16384, 40), (4, 5, 9216, 64), (1, 10, 4096, 64)]:
                qkv = rnd(B, N, 3 * heads * d)
                C = heads * d
                fl = 4.0 * B * heads * N * N * d
                us = hip._time_graph(lambda i: hip.attn_flash(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, d ** -0.5), iters=5)
                usp = hip._time_graph(lambda i: hip.attn_flash(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, d ** -0.5, out_planes=True), iters=5)
                qp = planes.split(qkv)
                uspp = hip._time_graph(lambda i: planes.attn_flash(qp[..., :C], qp[..., C:2 * C], qp[..., 2 * C:], heads, d ** -0.5), iters=5) if d in planes.FLASH_PLANES_DIMS else float("nan")
                print(f"attn_flash_x3 B={B} h={heads} N={N:6d} d={d:3d}: {us:9.1f} us {fl / us / 1e6:6.1f} TF  (planes out: {usp:9.1f} us; planes in + out: "
                      f"{uspp:9.1f} us {fl / uspp / 1e6:6.1f} TF)", flush=True)
        if args.geglu:
            shapes = [(16384, 2560, 320), (4096, 5120, 640), (1024, 10240, 1280), (256, 10240, 1280)]
            if os.environ.get("IEF_BENCH_FF1_ALL") == "1":        # every FeedForward.net[0] shape of the plan table's configurations
                shapes = [(8192, 2560, 320), (32768, 2560, 320), (65536, 2560, 320), (9216, 2560, 320), (18432, 2560, 320), (36864, 2560, 320),
                          (2048, 5120, 640), (8192, 5120, 640), (16384, 5120, 640), (2304, 5120, 640), (4608, 5120, 640), (9216, 5120, 640),
                          (2048, 10240, 1280), (4096, 10240, 1280), (2304, 10240, 1280)]
            for M, N, K in shapes:
                a, w, bias = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
                wc = hip._cold_copies(w)
                for x in wc:
                    planes.weight_planes(x)
                ap_ = planes.split(a)
                outp = planes.Planes.empty(M, N // 2, device="cuda")
                fl = 2.0 * M * N * K
                line = f"ff1+geglu {M:6d}x{N:5d}x{K:5d} |"
                for t in tiles:
                    bm, bn = planes._TILES[t]
                    if N % bn:
                        continue
                    us = hip._time_graph(lambda i: planes.gemm(ap_, wc[i % len(wc)], bias=bias, geglu=True, out=False, out_planes=outp, tile=t, splits=1))
                    us0 = hip._time_graph(lambda i: planes.gemm(ap_, wc[i % len(wc)], bias=bias, out=False, out_planes=True, tile=t, splits=1))
                    line += f" t{t}: {us:7.1f} us {fl / us / 1e6:6.1f} TF (no geglu, planes out: {us0:7.1f}) |"
                print(line, flush=True)
                del wc
        if do_lin:
            lin = LIN if os.environ.get("IEF_BENCH_LIN_B1") != "1" else [(4096, 320, 320), (4096, 960, 320), (4096, 320, 1280), (1024, 640, 640), (1024, 1920, 640), (1024, 640, 2560), (256, 1280, 1280), (256, 3840, 1280), (256, 1280, 5120), (64, 1280, 1280)]
            for M, N, K in lin:
                a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
                wc = hip._cold_copies(w)
                for x in wc:
                    planes.weight_planes(x)
                ap_ = planes.split(a)
                out = torch.empty(M, N, device="cuda")
                fl = 2.0 * M * N * K
                us_old = hip._time_graph(lambda i: hip.gemm(a, wc[i % len(wc)], out=out))
                line = f"lin {M:6d}x{N:5d}x{K:5d}  old {us_old:7.1f} us {fl / us_old / 1e6:6.1f} TF |"
                for t in tiles:
                    bm, bn = planes._TILES[t]
                    if N % bn:
                        continue
                    best = None
                    for sp in splits:
                        if sp > 1 and K // 32 // sp < 4:
                            continue
                        us = hip._time_graph(lambda i: planes.gemm(ap_, wc[i % len(wc)], out=out, tile=t, splits=sp))
                        if best is None or us < best[0]:
                            best = (us, sp)
                    line += f" t{t}: {best[0]:7.1f} us {fl / best[0] / 1e6:6.1f} TF s{best[1]} |"
                print(line, flush=True)
                del wc
        if do_conv:
            for B, H, W, C1, C2, Cout, stride, ups, CE1, CE2 in CONV:
                x, x2 = rnd(B, H, W, C1), (rnd(B, H, W, C2) if C2 else None)
                K = 9 * (C1 + C2) + CE1 + CE2
                w = rnd(Cout, K, scale=K ** -0.5) if CE1 else rnd(Cout, 3, 3, C1 + C2, scale=K ** -0.5)
                Ho, Wo = (H * (2 if ups else 1)) // stride, (W * (2 if ups else 1)) // stride
                e1 = rnd(B, Ho, Wo, CE1) if CE1 else None
                e2 = rnd(B, Ho, Wo, CE2) if CE2 else None
                extra = (e1, e2) if CE1 else None
                wc = hip._cold_copies(w)
                for t_ in wc:
                    planes.weight_planes(t_)
                xp, x2p = planes.split(x), (planes.split(x2) if C2 else None)
                extrap = (planes.split(e1), planes.split(e2) if CE2 else None) if CE1 else None
                M = B * Ho * Wo
                fl = 2.0 * M * Cout * K
                us_old = hip._time_graph(lambda i: hip.conv3x3(x, wc[i % len(wc)], x2=x2, stride=stride, upsample=ups, extra=extra))
                line = f"conv {H:3d}x{W:3d} {C1:4d}+{C2:4d}->{Cout:4d} s{stride} u{int(ups)} e{CE1 + CE2:4d}  old {us_old:7.1f} us {fl / us_old / 1e6:6.1f} TF |"
                for t in tiles:
                    if t in (11, 12):
                        if stride != 1 or CE1 or (ups and (H * 2) * (W * 2) % 256) or (not ups and W > 64):
                            continue
                        t = 12 if ups else 11
                    bm, bn = planes._TILES[t]
                    if Cout % bn:
                        continue
                    best = None
                    for sp in splits:
                        if sp > 1 and K // 32 // sp < 4:
                            continue
                        try:
                            us = hip._time_graph(lambda i: planes.conv3x3(xp, wc[i % len(wc)], x2=x2p, stride=stride, upsample=ups,
                                                                          extra=extrap, tile=t, splits=sp))
                        except RuntimeError:
                            continue
                        if best is None or us < best[0]:
                            best = (us, sp)
                    if best:
                        line += f" t{t}: {best[0]:7.1f} us {fl / best[0] / 1e6:6.1f} TF s{best[1]} |"
                print(line, flush=True)
                del wc


if __name__ == "__main__":
    main()
