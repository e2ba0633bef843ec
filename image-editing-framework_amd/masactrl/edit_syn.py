"""Edit a synthesized image with MasaCtrl — CLI of `/root/reference/masactrl/edit_syn.py` (same flags, defaults
`STEP = 4`, `LAYPER = 10`, outputs `./exp/source.png`, `./exp/edit.png`)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "p2p"))
from _bootstrap import load_pipe, seed_everything  # noqa: E402

from ief_amd.masactrl.model.attention_base import AttentionBase  # noqa: E402
from ief_amd.masactrl.model.attention_control import MutualSelfAttentionControl  # noqa: E402
from ief_amd.masactrl.model.register import regiter_attention_editor_diffusers  # noqa: E402
from ief_amd.masactrl.model.sd_utils import MasaCtrl  # noqa: E402
from ief_amd.p2p.utils.save_image import save_img  # noqa: E402

parser = argparse.ArgumentParser("General config")
parser.add_argument("--sd_version", type=str, default="1.5")
parser.add_argument("--device", type=int, default=0)
parser.add_argument("--seed", type=int, default=8888)
parser.add_argument("--source_prompt", type=str, default="A standing dog on the grass field")
parser.add_argument("--target_prompt", type=str, default="A running dog on the grass field")


def main(argv=None):
    args = parser.parse_args(argv)
    device = torch.device("cuda:{}".format(args.device))
    seed_everything(args.seed)
    num_inference_steps, GUIDANCE_SCALE, STEP, LAYPER = 50, 7.5, 4, 10
    out_path = "./exp"
    pipe = load_pipe(args.sd_version, device)
    size = pipe.unet.config.sample_size * pipe.vae_scale_factor
    if pipe.__class__.__name__ == "StableDiffusionXLPipeline":          # dispatch of edit_syn.py:87-98
        from ief_amd.masactrl.model.sd_utils import MasaCtrl_XL
        model_type, LAYPER, editor = "SDXL", 54, MasaCtrl_XL(pipe, num_inference_steps)
    else:
        model_type, editor = "SD", MasaCtrl(pipe, num_inference_steps)
    os.makedirs(out_path, exist_ok=True)
    controller = AttentionBase()
    regiter_attention_editor_diffusers(editor.model, controller)
    image, init_latent = editor(prompt=[args.source_prompt], guidance_scale=GUIDANCE_SCALE,
                                num_inference_steps=num_inference_steps, height=size, width=size)
    save_img(image, os.path.join(out_path, "source.png"))
    init_latent = torch.cat([init_latent, init_latent])
    controller = MutualSelfAttentionControl(STEP, LAYPER, model_type=model_type)
    regiter_attention_editor_diffusers(editor.model, controller)
    image_masactrl, _ = editor(prompt=[args.source_prompt, args.target_prompt], latents=init_latent,
                               guidance_scale=GUIDANCE_SCALE, num_inference_steps=num_inference_steps, height=size,
                               width=size)
    save_img(image_masactrl[1], os.path.join(out_path, "edit.png"))


if __name__ == "__main__":
    main()
