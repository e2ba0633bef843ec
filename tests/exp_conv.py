"""EXPERIMENT: where does the implicit-GEMM conv spend its time?  flags 0x100 = every LDS-DMA reads the zero page,
0x200 = only the first K tile is loaded (MFMA + LDS reads only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ief_amd
from ief_amd import hip
from bench_kernels import timeit, h
x, w, b = h(4, 64, 64, 320), h(320, 3, 3, 320, scale=0.02), torch.randn(320, device="cuda:0")
gf = 2 * 16384 * 320 * 2880 / 1e9
for tile, st in ((6, 2), (6, 3), (7, 2), (7, 3), (7, 4), (1, 3), (9, 3)):
    row = []
    for fl in ("0", "0x100", "0x200"):
        os.environ["IEF_DBG_FLAGS"] = fl
        try:
            us = timeit(lambda: hip.conv3x3(x, w, b, tile_hint=tile, stages=st), 20)
            row.append(f"{fl}: {us:7.1f} us {gf / us * 1e-3:6.0f} TF/s")
        except Exception as e:
            row.append(f"{fl}: fail {e}")
    print(f"tile {tile} stages {st}: " + " | ".join(row), flush=True)
