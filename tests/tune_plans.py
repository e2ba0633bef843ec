"""Autotune the (tile, split-K) plan of every GEMM / conv layer shape of the SD1.5 UNet at UNet batch 4 / 2 / 1
(edit, synthesis, inversion) on the current GPU and write the table the binding loads at import.

    python tests/tune_plans.py [out.json]       (run on the MI355X box; ~1 minute)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ief_amd  # noqa: F401
from ief_amd import hip
import bench

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join("gpurun_out", "tuned_plans.json")
cfg_name = sys.argv[2] if len(sys.argv) > 2 else "sd15"
dev = torch.device("cuda:0")
if os.environ.get("IEF_TUNE_KEEP", "0") != "1":      # IEF_TUNE_KEEP=1: keep the committed table, tune only missing shapes
    os.environ["IEF_NO_PLAN_TABLE"] = "1"
    hip._plans = {}
hip.AUTOTUNE = True
pipe, cfg = bench.build_pipe(cfg_name, dev, 0, 1)
hw = int(os.environ.get('IEF_TUNE_LATENT', cfg.sample_size))   # e.g. 128: the 1024x1024 shapes of bench.py --latent 128
for B in (4, 2, 1):
    x = torch.randn(B, 4, hw, hw, device=dev)
    ctx = (torch.randn(B, 77, cfg.cross_attention_dim, device=dev) * 0.1)
    added = None
    if cfg.addition_embed:
        added = {"text_embeds": torch.randn(B, cfg.pooled_text_dim, device=dev),
                 "time_ids": torch.tensor([[hw * 8.0, hw * 8.0, 0.0, 0.0, hw * 8.0, hw * 8.0]] * B, device=dev)}
    with torch.no_grad():
        pipe.unet(x, 501, encoder_hidden_states=ctx, added_cond_kwargs=added)
    torch.cuda.synchronize()
    print(f"B={B}: {len(hip._plan_table())} shapes tuned", flush=True)
if cfg.addition_embed or os.environ.get('IEF_TUNE_FORWARD_ONLY') == '1':          # forward shapes only
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    hip.save_plans(out)
    sys.exit(0)
# null-text optimisation: the data-gradient GEMMs / convs of the reverse pass at batch 1 (grad.UNetAdjoint)
from ief_amd.grad import UNetAdjoint
adj = UNetAdjoint(pipe.unet)
x = torch.randn(1, 4, hw, hw, device=dev)
ctx16 = (torch.randn(1, 77, cfg.cross_attention_dim, device=dev) * 0.1).half()
temb = pipe.unet.time_rows(torch.tensor([501.0], device=dev))
eps = adj.forward(x, temb, ctx16)
adj.backward(torch.randn_like(eps).contiguous())
torch.cuda.synchronize()
print(f"NTI reverse pass: {len(hip._plan_table())} shapes tuned", flush=True)
# Pix2Pix-zero: the same reverse pass at batch 2, continued down to the latent (grad.UNetAdjoint mode "input")
adj2 = UNetAdjoint(pipe.unet, 1024.0, mode="input")
x2 = torch.randn(2, 4, hw, hw, device=dev)
ctx2 = (torch.randn(2, 77, cfg.cross_attention_dim, device=dev) * 0.1).half()
pipe.unet(x2, 501, encoder_hidden_states=ctx2)
cross = [m for m in pipe.unet.attention_modules() if m.is_cross]
adj2.set_reference_maps([torch.softmax(torch.randn(2 * m.heads, m.last_tokens, 77, device=dev), -1).half() for m in cross])
adj2.forward(x2, temb, ctx2)
adj2.backward(torch.zeros_like(x2))
torch.cuda.synchronize()
print(f"Pix2Pix-zero reverse pass: {len(hip._plan_table())} shapes tuned", flush=True)
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
hip.save_plans(out)
for k, v in sorted(hip._plan_table().items()):
    print(k, v)
