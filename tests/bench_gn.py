"""GroupNorm -> planes, per call, by form (tuning aid): python tests/bench_gn.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ief_amd  # noqa: E402,F401
from ief_amd import hip, planes  # noqa: E402

SHAPES = [(64, 64, 320, 0), (32, 32, 320, 0), (32, 32, 640, 0), (32, 32, 640, 640), (16, 16, 1280, 0), (16, 16, 1280, 1280), (8, 8, 1280, 0),
          (8, 8, 1280, 1280), (16, 16, 640, 0)]
with hip.f32_contraction("x3"):
    for B in (1, 2, 4):
        for H, W, C1, C2 in SHAPES:
            x = torch.randn(B, H, W, C1, device="cuda")
            x2 = torch.randn(B, H, W, C2, device="cuda") if C2 else None
            g, b = torch.ones(C1 + C2, device="cuda"), torch.zeros(C1 + C2, device="cuda")
            line = f"B={B} {H}x{W} C={C1}+{C2}:"
            for wgs in (0, 1 << 30):
                planes.GN_REG_MAX_WGS = wgs
                us = hip._time_graph(lambda i: planes.groupnorm(x, g, b, 32, 1e-5, silu=True, x2=x2), iters=20)
                line += f"  {'slab in registers' if wgs else 'row-streaming / KS'} {us:6.1f} us"
            print(line, flush=True)
