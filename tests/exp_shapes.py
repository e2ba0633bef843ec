"""Per-layer-shape timing of one eager edit step (HIP events per launch, GPU kept busy while the host enqueues):
which GEMM / conv shapes the step's matrix time goes to and at what rate.   IEF_PROF_SHAPES=1 python tests/exp_shapes.py [cfg] [latent] [f16 | f32 | f16x3]"""
import os
import sys

os.environ["IEF_PROF_SHAPES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ief_amd  # noqa: F401
from ief_amd import hip
import bench

cfg_name = sys.argv[1] if len(sys.argv) > 1 else "sd15"
dev = torch.device("cuda:0")
precision = sys.argv[3] if len(sys.argv) > 3 else "f16"
pipe, cfg = bench.build_pipe(cfg_name, dev, 0, 1, precision=precision)
hw = int(sys.argv[2]) if len(sys.argv) > 2 else cfg.sample_size
B = int(os.environ.get("IEF_EXP_B", "4"))
x = torch.randn(B, 4, hw, hw, device=dev)
ctx = (torch.randn(B, 77, cfg.cross_attention_dim, device=dev) * 0.1)
agg = {}
with torch.no_grad():
    for it in range(6):
        if it:
            torch.cuda._sleep(int(1.2e8))
            hip.profile_begin()
        pipe.unet(x, 501, encoder_hidden_states=ctx)
        if it:
            for name, flops, ms in hip.profile_end():
                a = agg.setdefault(name, [0, 0.0, 0.0])
                a[0] += 1; a[1] += flops; a[2] += ms
tot = sum(a[2] for a in agg.values()) / 5
print(f"timed kernels: {tot:.3f} ms per forward")
for name, (n, fl, ms) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print(f"{ms / 5:7.3f} ms  x{n // 5:3d}  {ms / n * 1e3:7.1f} us  {fl / ms / 1e9 if fl else 0:7.1f} TF/s  {name}")
