"""GroupNorm -> planes, per call, by form (tuning aid): python tests/bench_gn.py
   three row-streaming launches | the one-launch form planes.groupnorm picks when allowed at any size (the slab-in-registers kernel
   where the slab fits, else the KS-workgroup kernel up to GN_SMALL_ELEMS elements)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ief_amd  # noqa: E402,F401
from ief_amd import hip, planes  # noqa: E402

SHAPES = [(64, 64, 320, 0), (32, 32, 320, 0), (32, 32, 640, 0), (32, 32, 640, 640), (32, 32, 1280, 640), (16, 16, 1280, 0), (16, 16, 1280, 1280),
          (8, 8, 1280, 0), (8, 8, 1280, 1280), (16, 16, 640, 0)]
with hip.f32_contraction("x3"):
    lib = hip.load()
    for B in (1, 2, 4):
        for H, W, C1, C2 in SHAPES:
            x = torch.randn(B, H, W, C1, device="cuda")
            x2 = torch.randn(B, H, W, C2, device="cuda") if C2 else None
            g, b = torch.ones(C1 + C2, device="cuda"), torch.zeros(C1 + C2, device="cuda")
            fits = bool(lib.ief_groupnorm_reg_fits(C1, C2, H * W, 32))
            line = f"B={B} {H}x{W} C={C1}+{C2}:"
            for label, max_hw in (("three row-streaming launches", 0), ("slab in registers" if fits else "KS-workgroup launch", 1 << 30)):
                planes.GN_REG_MAX_HW = max_hw
                us = hip._time_graph(lambda i: planes.groupnorm(x, g, b, 32, 1e-5, silu=True, x2=x2), iters=20)
                line += f"  {label} {us:6.1f} us |"
            print(line, flush=True)
