import sys, torch
sys.path.insert(0, ".")
import ief_amd
from ief_amd.pipeline import StableDiffusionPipeline
from oracle import unet_ref
name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pipe = StableDiffusionPipeline.from_pretrained(f"synthetic:{name}", keep_state_dict=True)
cfg = pipe.cfg
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 4, cfg.sample_size, cfg.sample_size, generator=g)
ctx = torch.randn(B, 77, cfg.cross_attention_dim, generator=g) * 0.1
taps, rtaps = {}, {}
eps = pipe.unet(x.cuda(), 981, encoder_hidden_states=ctx.cuda(), taps=taps)["sample"].cpu()
ref = unet_ref.unet_forward(pipe._state_dict, cfg, x, torch.tensor(981), ctx, taps=rtaps)
for k in rtaps:
    e = (taps[k] - rtaps[k]).abs().max() / rtaps[k].abs().max()
    print(f"{k:10s} rel err {e:.3e}  ref max {rtaps[k].abs().max():.2f}")
print("eps rel err", ((eps - ref).abs().max() / ref.abs().max()).item())
