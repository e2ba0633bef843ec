import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ief_amd  # noqa: F401
from ief_amd import hip
from bench_kernels import timeit, h, DEV
x = torch.randn(4, 4, 64, 64, device=DEV)
w = h(3, 3, 4, 320, scale=1 / 6); b = torch.randn(320, device=DEV)
xa = h(4, 64, 64, 320); wo = h(4, 3, 3, 320, scale=0.02); bo = torch.randn(4, device=DEV)
for r in range(3):
    print("conv_in %.1f us   conv_out %.1f us" % (timeit(lambda: hip.conv_in(x, w, b), 30), timeit(lambda: hip.conv_out(xa, wo, bo), 30)), flush=True)
