"""Synthesize an image and edit it with Plug-and-Play — CLI of `/root/reference/pnp/edit_syn.py` (same flags and
defaults; `pnp_attn_t = pnp_f_t = 1.0`, :37-39; outputs `./exp/source.png`, `./exp/edit.png`)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "p2p"))
from _bootstrap import load_pipe, seed_everything  # noqa: E402

from ief_amd.p2p.utils.save_image import save_img  # noqa: E402
from ief_amd.pnp.model.sd_utils import PnP, PnP_XL  # noqa: E402

parser = argparse.ArgumentParser("General config")
parser.add_argument("--sd_version", type=str, default="1.5")
parser.add_argument("--device", type=int, default=0)
parser.add_argument("--seed", type=int, default=74089447)
parser.add_argument("--source_prompt", type=str, default="A crisp, juicy green apple sits perched on a wooden table, "
                    "its smooth surface glistening in the light")
parser.add_argument("--target_prompt", type=str, default="A crisp, juicy red apple sits perched on a wooden table, "
                    "its smooth surface glistening in the light")


def main(argv=None):
    args = parser.parse_args(argv)
    device = torch.device("cuda:{}".format(args.device))
    seed_everything(args.seed)
    num_inference_steps, GUIDANCE_SCALE = 50, 7.5
    pnp_attn_t, pnp_f_t = 1.0, 1.0
    out_path = "./exp"
    pipe = load_pipe(args.sd_version, device)
    xl = pipe.__class__.__name__ == "StableDiffusionXLPipeline"          # dispatch of edit_syn.py:88-91
    editor = (PnP_XL if xl else PnP)(pipe, num_inference_steps)
    os.makedirs(out_path, exist_ok=True)
    images = editor(prompt=[args.source_prompt] + [args.target_prompt], num_inference_steps=num_inference_steps,
                    guidance_scale=GUIDANCE_SCALE, pnp_attn_t=pnp_attn_t, pnp_f_t=pnp_f_t)
    save_img(images[0], os.path.join(out_path, "source.png"))
    save_img(images[1], os.path.join(out_path, "edit.png"))


if __name__ == "__main__":
    main()
