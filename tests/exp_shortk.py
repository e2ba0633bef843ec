"""where the time of the K = C square projections goes: launch of M=16384 N=320 at several K, with / without residual,
per tile plan; us per launch from a replayed hipGraph of 20 launches on cache-cold weights / activations"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ief_amd
from ief_amd import hip
dev = torch.device("cuda:0")
M, N = 16384, 320
for K in (64, 128, 320, 640, 1280):
    acts = [torch.randn(M, K, device=dev).half() for _ in range(8)]       # 8 x 10 MB at K=320: rotate so A comes from HBM
    w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    res = [torch.randn(M, N, device=dev).half() for _ in range(8)]
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    for tile, st in ((5, 2), (5, 3), (7, 2), (7, 3), (3, 2)):
        try:
            t_res = hip._time_graph(lambda i: hip.gemm(acts[i % 8], w, bias=bias, residual=res[i % 8], out=out, tile_hint=tile, stages=st))
            t_nores = hip._time_graph(lambda i: hip.gemm(acts[i % 8], w, bias=bias, out=out, tile_hint=tile, stages=st))
        except RuntimeError as e:
            print(K, tile, st, "n/a"); continue
        mb = 2 * (M * K + M * N * 2 + N * K) / 1e6
        print(f"K={K:5d} tile {tile} NS {st}: {t_res:6.2f} us with residual ({mb / t_res / 1e3 * 1e3:6.0f} GB/s), {t_nores:6.2f} us without", flush=True)
