"""Invert a real image, then edit it with Pix2Pix-zero — CLI of `/root/reference/pix2pix-zero/edit_real.py` (same flags and
defaults, `--inversion_type null-text` included): DDIM inversion under the source prompt (+ null-text optimisation),
then `P2P_Zero` / `P2P_Zero_NTI` from the inverted latent; outputs `./exp/source.png`, `./exp/inversion.png`,
`./exp/edit.png` (:118-139)."""
import argparse
import os
import sys

import torch
from PIL import Image

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "p2p"))
from _bootstrap import load_pipe, seed_everything  # noqa: E402

from ief_amd.p2p.inversion.ddim import ddim_inversion, ddim_inversion_xl  # noqa: E402
from ief_amd.p2p.inversion.nti import NTI, NTI_XL_5e2 as NTI_XL  # noqa: E402  (this folder's copy: lr 5e-2)
from ief_amd.p2p.utils.save_image import save_img  # noqa: E402
from ief_amd.pix2pix_zero.model.sd_utils import P2P_Zero, P2P_Zero_NTI, P2P_Zero_XL, P2P_Zero_XL_NTI  # noqa: E402

parser = argparse.ArgumentParser("General config")
parser.add_argument("--sd_version", type=str, default="1.5")
parser.add_argument("--device", type=int, default=0)
parser.add_argument("--seed", type=int, default=42)
parser.add_argument("--source_prompt", type=str, default="a round cake with orange frosting on a wooden plate")
parser.add_argument("--target_prompt", type=str, default="a square cake with orange frosting on a wooden plate")
parser.add_argument("--source_image", type=str, default="./test.jpg")
parser.add_argument("--inversion_type", type=str, default="null-text")


def main(argv=None):
    args = parser.parse_args(argv)
    if args.inversion_type not in ("ddim", "null-text"):
        raise ValueError("--inversion_type must be ddim or null-text")
    device = torch.device("cuda:{}".format(args.device))
    seed_everything(args.seed)
    num_inference_steps, GUIDANCE_SCALE = 50, 7.5
    num_inner_steps, early_stop_epsilon = 10, 1e-5
    only_sample = False
    out_path = "./exp"
    pipe = load_pipe(args.sd_version, device)
    size = pipe.unet.config.sample_size * pipe.vae_scale_factor
    xl = pipe.__class__.__name__ == "StableDiffusionXLPipeline"          # dispatch of edit_real.py:95-112
    if xl and args.inversion_type == "ddim":
        invertor, editor = ddim_inversion_xl(), P2P_Zero_XL(pipe, num_inference_steps)
    elif xl:
        invertor, editor = NTI_XL(), P2P_Zero_XL_NTI(pipe, num_inference_steps)
    elif args.inversion_type == "ddim":
        invertor, editor = ddim_inversion(), P2P_Zero(pipe, num_inference_steps)
    else:
        invertor, editor = NTI(), P2P_Zero_NTI(pipe, num_inference_steps)
    os.makedirs(out_path, exist_ok=True)
    original_image = Image.open(args.source_image).convert("RGB").resize((size, size))
    original_image.save(os.path.join(out_path, "source.png"))
    latent = invertor.image2latent(model=pipe, image=original_image, device=device, dtype=torch.float32)
    source_prompt, target_prompt = [args.source_prompt], [args.target_prompt]
    latents, context = invertor.ddim_inversion_loop(pipe, latent, source_prompt)
    extra = {}
    if args.inversion_type == "null-text":
        extra["uncond_embeddings_list"] = invertor.null_optimization(pipe, latents, context, num_inner_steps,
                                                                     early_stop_epsilon, GUIDANCE_SCALE)
    image_source, image_edit = editor(prompt=source_prompt + target_prompt, num_inference_steps=num_inference_steps,
                                      guidance_scale=GUIDANCE_SCALE, only_sample=only_sample, edit_dir=None,
                                      latents=latents[-1], **extra)
    save_img(image_source, os.path.join(out_path, "inversion.png"))
    save_img(image_edit, os.path.join(out_path, "edit.png"))


if __name__ == "__main__":
    main()
