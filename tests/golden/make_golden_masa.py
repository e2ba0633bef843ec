"""Golden vectors for the MasaCtrl folder's hooked attention, made by IMPORTING the reference's own
`/root/reference/masactrl/model/{register,attention_base}.py` (torch + einops: they import cleanly here) and running
`regiter_attention_editor_diffusers` on a toy module tree of `Attention` modules: the hooked forward (q/k/v projections,
head split `b n (h d) -> (b h) n d`, scaled scores, softmax, `AttentionBase.forward`, output projection) for a self- and a
cross-attention call, the layer count, and the editor's step counter.  Run in the build container only:

    python tests/golden/make_golden_masa.py        ->  tests/golden/masactrl_register.npz   (G12)
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/masactrl")
from model import register as ref_register  # noqa: E402  (reference)
from model.attention_base import AttentionBase  # noqa: E402


class Attention(nn.Module):      # the class NAME is what the reference's walk looks for (register.py:55)
    def __init__(self, dim, heads, ctx_dim=None):
        super().__init__()
        self.heads, self.scale = heads, (dim // heads) ** -0.5
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(ctx_dim or dim, dim, bias=False)
        self.to_v = nn.Linear(ctx_dim or dim, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])


class Block(nn.Module):
    def __init__(self, dim, heads, ctx_dim):
        super().__init__()
        self.attn1 = Attention(dim, heads)
        self.attn2 = Attention(dim, heads, ctx_dim)


class ToyUNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.down_blocks = nn.ModuleList([Block(64, 2, 32), Block(64, 4, 32)])
        self.mid_block = Block(64, 4, 32)
        self.up_blocks = nn.ModuleList([Block(64, 2, 32)])
        self.conv_out = nn.Linear(4, 4)          # not an attention holder: must not be counted


def main():
    torch.manual_seed(3)
    unet = ToyUNet()
    model = type("M", (), {})()
    model.unet = unet
    editor = AttentionBase()
    ref_register.regiter_attention_editor_diffusers(model, editor)
    out = {"num_att_layers": np.array(editor.num_att_layers)}
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 16, 64, generator=g)
    ctx = torch.randn(2, 7, 32, generator=g)
    out["x"], out["ctx"] = x.numpy(), ctx.numpy()
    mods = {"down0": unet.down_blocks[0], "down1": unet.down_blocks[1], "mid": unet.mid_block, "up0": unet.up_blocks[0]}
    with torch.no_grad():
        for name, blk in mods.items():
            for kind, a in (("attn1", blk.attn1), ("attn2", blk.attn2)):
                y = a.forward(x, encoder_hidden_states=ctx if kind == "attn2" else None)
                out[f"{name}.{kind}.out"] = y.numpy()
                out[f"{name}.{kind}.heads"] = np.array(a.heads)
                for w in ("to_q", "to_k", "to_v"):
                    out[f"{name}.{kind}.{w}.weight"] = getattr(a, w).weight.numpy()
                out[f"{name}.{kind}.to_out.0.weight"] = a.to_out[0].weight.numpy()
                out[f"{name}.{kind}.to_out.0.bias"] = a.to_out[0].bias.numpy()
    out["cur_step_after_8_calls"] = np.array(editor.cur_step)         # 8 calls = one full sweep of the 8 hooked layers
    out["cur_att_layer_after_8_calls"] = np.array(editor.cur_att_layer)
    np.savez_compressed(os.path.join(HERE, "masactrl_register.npz"), **out)
    print("masactrl_register.npz", len(out), "arrays; layers", editor.num_att_layers, "step", editor.cur_step)


if __name__ == "__main__":
    main()
