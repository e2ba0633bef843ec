"""Per-launch table of one eager P2P edit step (SD1.5 shapes): name, shape, ms, TFLOP/s.  Debug aid."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ief_amd
from ief_amd import hip
import bench

rec_shapes = []
_orig_gemm, _orig_conv = hip.gemm, hip.conv3x3
def gemm(a, w, **kw):
    rec_shapes.append(("gemm", a.numel() // a.shape[-1], w.shape[0], w.shape[1]))
    return _orig_gemm(a, w, **kw)
def conv3x3(x, w, bias=None, **kw):
    B, H, W, C = x.shape
    ups = 2 if kw.get("upsample") else 1
    s = kw.get("stride", 1)
    rec_shapes.append(("conv", B * (H * ups // s) * (W * ups // s), w.shape[0], w.numel() // w.shape[0]))
    return _orig_conv(x, w, bias, **kw)

dev = torch.device("cuda:0")
pipe, cfg = bench.build_pipe("sd15", dev, 0, 1)
from ief_amd.denoise import FusedDenoiser
from ief_amd.p2p.model.attention_control import AttentionRefine
from ief_amd.p2p.model.register import register_attention_control
from ief_amd.p2p.model.sd_utils import _encode_prompts
pipe.scheduler.set_timesteps(50)
x_T = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(8888)).to(dev)
with torch.no_grad():
    u, c = _encode_prompts(pipe, bench.PROMPTS)
ctx = torch.cat([u, c])
ctrl = AttentionRefine(bench.PROMPTS, pipe.tokenizer, 50, 0.8, 0.4, device=dev)
register_attention_control(pipe, ctrl)
loop = FusedDenoiser(pipe, ctx, 2, (64, 64), 7.5, use_graph=False)
loop.lat.copy_(x_T.expand_as(loop.lat))
loop._set_kv_cache(True)
loop._step_body(); loop._step_body()
torch.cuda.synchronize()
hip.gemm, hip.conv3x3 = gemm, conv3x3
import ief_amd.unet as U
hip.profile_begin()
loop._step_body()
rec = hip.profile_end()
gi = 0
rows = []
for name, flops, ms in rec:
    shape = ""
    if name.startswith("igemm"):
        shape = rec_shapes[gi]; gi += 1
    rows.append((ms, name, shape, flops))
agg = {}
for ms, name, shape, flops in rows:
    k = (name, shape)
    a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += flops
print(f"{'ms':>8} {'n':>3} {'TF/s':>7}  kernel shape(M,N,K)")
for (name, shape), (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{ms:8.3f} {n:3d} {fl / ms / 1e9 if ms > 0 else 0:7.1f}  {name} {shape}")
print("total ms", sum(r[0] for r in rows))
