"""A/B of the implicit-GEMM conv (tile 7) and the halo-tile conv (tile 14) on the SD1.5 batch-4 conv shapes it covers.
    python tests/ab_halo.py [--iters 30]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ief_amd  # noqa: F401
from ief_amd import hip
from bench_kernels import timeit, h, DEV

SHAPES = [(4, 64, 64, 320, 0, 320, (1,)), (4, 64, 64, 320, 320, 320, (1,)), (4, 64, 64, 640, 320, 320, (1,)),
          (4, 32, 32, 640, 0, 640, (1, 2)), (4, 32, 32, 1280, 640, 640, (2, 4)), (4, 16, 16, 1280, 0, 1280, (2, 4, 5)),
          (4, 16, 16, 1280, 1280, 1280, (4, 8)), (4, 8, 8, 1280, 0, 1280, (10, 20))]
iters = 30
for B, H, W, C1, C2, Cout, sps in SHAPES:
    x = h(B, H, W, C1)
    x2 = h(B, H, W, C2) if C2 else None
    w = h(Cout, 3, 3, C1 + C2, scale=(9 * (C1 + C2)) ** -0.5)
    bias = torch.randn(Cout, device=DEV)
    M, K = B * H * W, 9 * (C1 + C2)
    t, sp, st = hip.pick_plan(M, Cout, K, conv=True)
    arms = [(f"t{t}s{sp}r{st}", lambda: hip.conv3x3(x, w, bias, x2=x2, tile_hint=t, splits=sp, stages=st))]
    for s_ in sps:
        for t_ in (14, 15):
            arms.append((f"t{t_} s{s_}", (lambda q, tt: (lambda: hip.conv3x3(x, w, bias, x2=x2, tile_hint=tt, splits=q, stages=4)))(s_, t_)))
    res = {n: [] for n, _ in arms}
    for rnd in range(4):                      # interleaved rounds, one process (guide rule 24)
        for n, f in arms:
            res[n].append(timeit(f, iters))
    print(f"{(B, H, W, C1, C2, Cout)}: " + " | ".join(f"{n}: {min(v):6.1f}/{sorted(v)[len(v) // 2]:6.1f}" for n, v in res.items()), flush=True)
