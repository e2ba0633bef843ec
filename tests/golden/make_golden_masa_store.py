"""Fixture G14 — MasaCtrl's `AttentionStore` editor, made by IMPORTING the reference's own
`/root/reference/masactrl/model/attention_base.py` (:33-66) and driving it through four denoising steps of a four-layer toy
schedule (self / cross maps of two resolutions, one of them above the 64^2 store limit is not needed: the limit is a size
test on `attn.shape[1]`, exercised with a patched limit-sized dummy).  Recorded: every call's output, the counters, and the
state of the store's four lists after every step — including the reference's aliasing quirk (`self.self_attns =
self.self_attns_step` followed by `self.self_attns_step.clear()` empties both, :50-59).  Build container only:

    python tests/golden/make_golden_masa_store.py        ->  tests/golden/masactrl_store.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/masactrl")
from model.attention_base import AttentionStore  # noqa: E402  (reference)


def calls(seed):
    """the editor calls of one denoising step: (q, k, v, sim, attn, is_cross, place, heads)"""
    g = torch.Generator().manual_seed(seed)
    out = []
    for n, l, cross, place in ((16, 16, False, "down"), (16, 7, True, "down"), (64, 64, False, "up"), (64, 7, True, "up")):
        heads, d, b = 2, 8, 2
        q, k, v = (torch.randn(b * heads, m, d, generator=g) for m in (n, l, l))
        sim = q @ k.transpose(1, 2) * d ** -0.5
        out.append((q, k, v, sim, sim.softmax(-1), cross, place, heads))
    return out


def main():
    ed = AttentionStore(res=[32], min_step=1, max_step=4)
    ed.num_att_layers = 4
    rec = {}
    for step in range(4):
        for j, c in enumerate(calls(100 + step)):
            rec[f"out_{step}_{j}"] = ed(*c).numpy()
        rec[f"state_{step}"] = np.array([ed.cur_step, ed.cur_att_layer, ed.valid_steps, len(ed.self_attns), len(ed.cross_attns),
                                         len(ed.self_attns_step), len(ed.cross_attns_step)])
    ed.reset()
    rec["state_reset"] = np.array([ed.cur_step, ed.cur_att_layer, ed.valid_steps])
    np.savez_compressed(os.path.join(HERE, "masactrl_store.npz"), **rec)
    print({k: (v.shape if v.ndim else v) for k, v in rec.items() if k.startswith("state")}, [rec[k].tolist() for k in sorted(rec) if k.startswith("state")])


if __name__ == "__main__":
    main()
