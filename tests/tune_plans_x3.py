"""Autotune the (tile, split-K) plan of every planes GEMM / convolution (csrc/gemm_x3p.hip, csrc/conv_halo_x3p.hip) the f16x3
UNet launches, per layer shape, cache-cold, on the current GPU; writes the table `planes.py` loads at import.

    python tests/tune_plans_x3.py out.json [config=sd15] [latent]      batches 4 / 2 / 1 (edit, synthesis / CFG split, inversion)
    IEF_TUNE_KEEP=1: keep the committed table and tune only the shapes it lacks
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ief_amd  # noqa: F401
from ief_amd import hip, planes
import bench

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join("gpurun_out", "tuned_plans_x3.json")
cfg_name = sys.argv[2] if len(sys.argv) > 2 else "sd15"
dev = torch.device("cuda:0")
if os.environ.get("IEF_TUNE_KEEP", "0") != "1":
    planes._plans = {}
else:
    planes._plan_table()
planes.AUTOTUNE = True
pipe, cfg = bench.build_pipe(cfg_name, dev, 0, 1, precision="f16x3")
hw = int(sys.argv[3]) if len(sys.argv) > 3 else cfg.sample_size
for B in [int(b) for b in os.environ.get("IEF_TUNE_BATCHES", "4,2,1").split(",")]:
    x = torch.randn(B, 4, hw, hw, device=dev)
    ctx = torch.randn(B, 77, cfg.cross_attention_dim, device=dev) * 0.1
    added = None
    if cfg.addition_embed:
        added = {"text_embeds": torch.randn(B, cfg.pooled_text_dim, device=dev),
                 "time_ids": torch.tensor([[hw * 8.0, hw * 8.0, 0.0, 0.0, hw * 8.0, hw * 8.0]] * B, device=dev)}
    with torch.no_grad():
        pipe.unet(x, 501, encoder_hidden_states=ctx, added_cond_kwargs=added)
    torch.cuda.synchronize()
    print(f"{cfg_name} latent {hw} B={B}: {len(planes._plan_table())} shapes tuned", flush=True)
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    planes.save_plans(out)
for k, v in sorted(planes._plan_table().items()):
    print(k, v)
