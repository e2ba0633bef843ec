import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ief_amd  # noqa: F401
from ief_amd import hip
from bench_kernels import timeit, h, DEV
for B, heads, N, d in [(4, 8, 16384, 40), (4, 8, 4096, 40), (4, 5, 9216, 64)]:
    qkv = h(B, N, 3 * heads * d); C = heads * d
    r = [timeit(lambda: hip.attn_flash(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, d ** -0.5), 10) for _ in range(3)]
    print((B, heads, N, d), "%.1f us  %.0f TF/s" % (min(r), 4.0 * B * heads * N * N * d / min(r) / 1e6), flush=True)
