"""Upsample2D convolutions (nearest-2x fused): the tuned implicit-GEMM plan vs the halo kernel (tile 15), interleaved rounds"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ief_amd  # noqa: F401
from ief_amd import hip
from bench_kernels import timeit, h, DEV
for B, Hs, Ws, C, Cout, sps in [(4, 32, 32, 640, 640, (1,)), (4, 16, 16, 1280, 1280, (1, 2)), (4, 8, 8, 1280, 1280, (2, 4)), (4, 64, 64, 320, 320, (1,))]:
    x = h(B, Hs, Ws, C); w = h(Cout, 3, 3, C, scale=(9 * C) ** -0.5); bias = torch.randn(Cout, device=DEV)
    M, K = B * Hs * Ws * 4, 9 * C
    t, sp, st = hip.pick_plan(M, Cout, K, conv=True)
    if t in (14, 15):
        t, sp, st = 7, sp, 3
    arms = [(f"t{t}s{sp}r{st}", lambda: hip.conv3x3(x, w, bias, upsample=True, tile_hint=t, splits=sp, stages=st))]
    for s_ in sps:
        arms.append((f"t15 s{s_}", (lambda q: (lambda: hip.conv3x3(x, w, bias, upsample=True, tile_hint=15, splits=q, stages=4)))(s_)))
    res = {n: [] for n, _ in arms}
    for rnd in range(3):
        for n, f in arms:
            res[n].append(timeit(f, 30))
    print(f"{(B, Hs, Ws, C, Cout)} -> x2: " + " | ".join(f"{n}: {min(v):6.1f}" for n, v in res.items()), flush=True)
