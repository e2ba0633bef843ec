"""Kernel microbenchmarks on the SD1.5 (B=4, 512x512) layer shapes.  Tuning aid, not a test.

    python tests/bench_kernels.py [gemm|conv|attn|norm|all] [--iters 30]
Each shape: warm-up, then `iters` back-to-back launches between two HIP events (queue stays full, so
the figure is device time).  Operands are random (zero-filled operands read high: guide rule 25).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ief_amd  # noqa: F401
from ief_amd import hip

DEV = torch.device("cuda:0")


def timeit(fn, iters):
    """device time per call: `iters` launches captured in one hipGraph (no host launch cost in the figure)"""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def h(*shape, scale=1.0):
    return (torch.randn(*shape, device=DEV) * scale).half()


GEMMS = [  # (M, N, K, count per step)
    (16384, 320, 320, 25), (4096, 640, 640, 25), (1024, 1280, 1280, 25), (16384, 960, 320, 5), (16384, 2560, 320, 5),
    (16384, 320, 1280, 5), (4096, 1920, 640, 5), (4096, 5120, 640, 5), (4096, 640, 2560, 5), (1024, 3840, 1280, 5),
    (1024, 10240, 1280, 5), (1024, 1280, 5120, 5), (256, 1280, 1280, 5), (256, 10240, 1280, 1), (256, 1280, 5120, 1),
]
CONVS = [  # (B, H, W, C1, C2, Cout, stride, ups, count)
    (4, 64, 64, 320, 0, 320, 1, False, 8), (4, 64, 64, 320, 320, 320, 1, False, 2), (4, 64, 64, 640, 320, 320, 1, False, 1),
    (4, 64, 64, 640, 0, 640, 1, True, 1), (4, 32, 32, 640, 0, 640, 1, False, 7), (4, 32, 32, 320, 0, 640, 1, False, 1),
    (4, 32, 32, 1280, 640, 640, 1, False, 1), (4, 32, 32, 1280, 0, 1280, 1, True, 1), (4, 16, 16, 1280, 0, 1280, 1, False, 8),
    (4, 16, 16, 1280, 1280, 1280, 1, False, 2), (4, 8, 8, 1280, 0, 1280, 1, False, 9), (4, 8, 8, 1280, 1280, 1280, 1, False, 3),
    (4, 64, 64, 320, 0, 320, 2, False, 1),
]
ATTN = [(4, 8, 4096, 40, 5), (4, 8, 1024, 80, 5), (4, 8, 256, 160, 5), (4, 8, 64, 160, 1)]


def sweep(fn_of_plan, M, N, K, iters, conv=False):
    """(us, tile, splits, stages) sorted, over the tile family x split-K x LDS ring depth"""
    res = []
    for t, s, st in hip.candidate_plans(M, N, K, conv=conv):
        try:
            res.append((timeit(lambda: fn_of_plan(t, s, st), iters), t, s, st))
        except RuntimeError:
            pass
    res.sort()
    return res


def run_gemm(iters, do_sweep=False):
    tot = tot_best = 0.0
    print(f"{'us':>8} {'TF/s':>7} {'us*n':>8}  gemm (M,N,K) plan")
    for M, N, K, n in GEMMS:
        a, w, bias, res = h(M, K), h(N, K, scale=K ** -0.5), torch.randn(N, device=DEV), h(M, N)
        out = torch.empty(M, N, dtype=torch.float16, device=DEV)
        us = timeit(lambda: hip.gemm(a, w, bias=bias, residual=res, out=out), iters)
        tot += us * n
        line = f"{us:8.1f} {2.0 * M * N * K / us / 1e6:7.1f} {us * n:8.1f}  ({M},{N},{K}) {hip.pick_plan(M, N, K)}"
        if do_sweep:
            r = sweep(lambda t, s, st: hip.gemm(a, w, bias=bias, residual=res, out=out, tile_hint=t, splits=s, stages=st), M, N, K, iters)
            tot_best += r[0][0] * n
            line += "  best: " + " ".join(f"{u:.1f}us@t{t}s{s}r{st}" for u, t, s, st in r[:4])
        print(line, flush=True)
    print(f"gemm total per step: {tot / 1e3:.3f} ms" + (f"  (best-of-sweep {tot_best / 1e3:.3f} ms)" if do_sweep else ""))


def run_conv(iters, do_sweep=False):
    tot = tot_best = 0.0
    print(f"{'us':>8} {'TF/s':>7} {'us*n':>8}  conv (B,H,W,C1,C2,Cout,stride,ups) plan")
    for B, H, W, C1, C2, Cout, s, ups, n in CONVS:
        x = h(B, H // (2 if ups else 1), W // (2 if ups else 1), C1)
        x2 = h(B, H, W, C2) if C2 else None
        w = h(Cout, 3, 3, C1 + C2, scale=(9 * (C1 + C2)) ** -0.5)
        bias = torch.randn(Cout, device=DEV)
        fn = lambda: hip.conv3x3(x, w, bias, x2=x2, stride=s, upsample=ups)
        us = timeit(fn, iters)
        Ho = H // s
        M, K = B * Ho * Ho, 9 * (C1 + C2)
        tot += us * n
        line = f"{us:8.1f} {2.0 * M * Cout * K / us / 1e6:7.1f} {us * n:8.1f}  {(B, H, W, C1, C2, Cout, s, ups)} {hip.pick_plan(M, Cout, K, conv=True)}"
        if do_sweep:
            r = sweep(lambda t, sp, st: hip.conv3x3(x, w, bias, x2=x2, stride=s, upsample=ups, tile_hint=t, splits=sp, stages=st), M, Cout, K, iters, conv=True)
            tot_best += r[0][0] * n
            line += "  best: " + " ".join(f"{u:.1f}us@t{t}s{sp}r{st}" for u, t, sp, st in r[:4])
        print(line, flush=True)
    print(f"conv total per step: {tot / 1e3:.3f} ms" + (f"  (best-of-sweep {tot_best / 1e3:.3f} ms)" if do_sweep else ""))


def run_attn(iters):
    print(f"{'us':>8} {'TF/s':>7}  attention (B,heads,N,d)")
    for B, heads, N, d, n in ATTN:
        qkv = h(B, N, 3 * heads * d)
        C = heads * d
        us = timeit(lambda: hip.attn_flash(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, d ** -0.5), iters)
        print(f"{us:8.1f} {4.0 * B * heads * N * N * d / us / 1e6:7.1f}  flash {(B, heads, N, d)} x{n}")
        q, kv = h(B, N, C), h(B, 77, 2 * C)
        us = timeit(lambda: hip.attn_cross_p2p(q, kv[..., :C], kv[..., C:], heads, d ** -0.5), iters)
        print(f"{us:8.1f} {4.0 * B * heads * N * 77 * d / us / 1e6:7.1f}  cross {(B, heads, N, d)}")


def run_norm(iters):
    print(f"{'us':>8} {'GB/s':>7}  norm")
    for HW, C in [(4096, 320), (4096, 640), (4096, 960), (1024, 640), (1024, 1920), (256, 1280), (256, 2560), (64, 1280), (64, 2560)]:
        x = h(4, HW, C)
        g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        us = timeit(lambda: hip.groupnorm(x, g, b, 32, 1e-5, silu=True), iters)
        print(f"{us:8.1f} {3.0 * x.numel() * 2 / us / 1e3:7.1f}  groupnorm B=4 HW={HW} C={C} (3 passes of the tensor)")
    for rows, C in [(16384, 320), (4096, 640), (1024, 1280)]:
        x = h(rows, C)
        g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        us = timeit(lambda: hip.layernorm(x, g, b), iters)
        print(f"{us:8.1f} {2.0 * x.numel() * 2 / us / 1e3:7.1f}  layernorm rows={rows} C={C}")
    for rows, Ch in [(16384, 1280), (4096, 2560), (1024, 5120)]:
        x = h(rows, 2 * Ch)
        us = timeit(lambda: hip.geglu(x), iters)
        print(f"{us:8.1f} {3.0 * rows * Ch * 2 / us / 1e3:7.1f}  geglu rows={rows} Ch={Ch}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--sweep", action="store_true", help="also try every tile x split-K plan per shape")
    a = ap.parse_args()
    for name, fn in (("gemm", run_gemm), ("conv", run_conv), ("attn", run_attn), ("norm", run_norm)):
        if a.what in (name, "all"):
            if name in ("gemm", "conv"):
                fn(a.iters, a.sweep)
            else:
                fn(a.iters)
