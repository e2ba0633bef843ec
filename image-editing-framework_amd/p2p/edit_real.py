"""Invert a real image, then edit it with Prompt-to-Prompt — CLI of `/root/reference/p2p/edit_real.py`.

Same flags and defaults (`--inversion_type` defaults to "null-text" as in the reference, :26), same outputs:
`./exp/source.png`, `./exp/inversion.png`, `./exp/edit.png`.
"""
import argparse
import os

import torch
from PIL import Image

from _bootstrap import load_pipe, seed_everything

from ief_amd.p2p.inversion.ddim import ddim_inversion, ddim_inversion_xl
from ief_amd.p2p.inversion.nti import NTI, NTI_XL
from ief_amd.p2p.model.attention_control import AttentionRefine, AttentionReplace
from ief_amd.p2p.model.register import unregister_attention_control
from ief_amd.p2p.model.sd_utils import P2P, P2P_NTI, P2P_XL, P2P_XL_NTI
from ief_amd.p2p.utils.save_image import save_img

parser = argparse.ArgumentParser("General config")
parser.add_argument("--sd_version", type=str, default="1.5")
parser.add_argument("--device", type=int, default=0)
parser.add_argument("--seed", type=int, default=42)
parser.add_argument("--source_prompt", type=str, default="a gray horse in the field")
parser.add_argument("--target_prompt", type=str, default="a whie horse in the field")
parser.add_argument("--source_image", type=str, default="./test.jpg")
parser.add_argument("--inversion_type", type=str, default="null-text")
# not a reference flag; default = the mode that meets the reference's fp32 results to 1e-3 (see edit_syn.py); every inversion
# type runs in every mode (the null-text reverse pass in fp32 for "f16x3" / "f32")
parser.add_argument("--precision", type=str, default=os.environ.get("IEF_PRECISION", "f16x3"), choices=["f16", "f32", "f16x3"])


def edit_latent(pipe, editor, x_T, source_prompt, target_prompt, edit_type, device, extra=None, num_inference_steps=50,
                guidance_scale=7.5, cross_replace_steps=0.8, self_replace_steps=0.6):
    """Prompt-to-Prompt edit from an inverted latent x_T [1,4,h,w] -> uint8 images [2,H,W,3] (reconstruction, edit)"""
    kw = dict(prompts=source_prompt + target_prompt, tokenizer=pipe.tokenizer, num_steps=num_inference_steps,
              cross_replace_steps=cross_replace_steps, self_replace_steps=self_replace_steps, device=device)
    if edit_type == "replace":
        controller = AttentionReplace(**kw)
    elif edit_type == "refine":
        controller = AttentionRefine(**kw)
    else:
        raise ValueError("Please choose right eidt type")
    images, _ = editor.text2image_ldm_stable(pipe, source_prompt + target_prompt, controller, latent=x_T,
                                             num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                                             low_resource=False, **(extra or {}))
    controller.reset()
    unregister_attention_control(pipe, controller)
    return images


def edit_one(pipe, editor, invertor, image, source_prompt, target_prompt, inversion_type, edit_type, device,
             num_inference_steps=50, guidance_scale=7.5, cross_replace_steps=0.8, self_replace_steps=0.6,
             num_inner_steps=10, early_stop_epsilon=1e-5):
    """invert + edit one PIL image -> uint8 images [2,H,W,3] (inversion reconstruction, edit)"""
    latent = invertor.image2latent(model=pipe, image=image, device=device, dtype=torch.float32)
    latents, context = invertor.ddim_inversion_loop(pipe, latent, source_prompt)
    extra = {}
    if inversion_type == "null-text":
        extra["uncond_embeddings_list"] = invertor.null_optimization(pipe, latents, context, num_inner_steps,
                                                                     early_stop_epsilon, guidance_scale)
    elif inversion_type != "ddim":
        raise ValueError("Please choose right inversion type")
    return edit_latent(pipe, editor, latents[-1], source_prompt, target_prompt, edit_type, device, extra,
                       num_inference_steps, guidance_scale, cross_replace_steps, self_replace_steps)


def main(argv=None):
    args = parser.parse_args(argv)
    device = torch.device("cuda:{}".format(args.device))
    seed_everything(args.seed)
    out_path = "./exp"
    edit_type = "refine"  # ["refine", "replace"]
    pipe = load_pipe(args.sd_version, device, precision=args.precision)
    xl = pipe.__class__.__name__ == "StableDiffusionXLPipeline"          # dispatch of edit_real.py:97-115
    if args.inversion_type == "ddim":
        editor = (P2P_XL if xl else P2P)(model=pipe, num_inference_steps=50)
        invertor = ddim_inversion_xl() if xl else ddim_inversion()
    elif args.inversion_type == "null-text":
        editor = (P2P_XL_NTI if xl else P2P_NTI)(model=pipe, num_inference_steps=50)
        invertor = NTI_XL() if xl else NTI()
    else:
        raise ValueError("Please choose right inversion type")
    size = pipe.unet.config.sample_size * pipe.vae_scale_factor
    image = Image.open(args.source_image).convert("RGB").resize((size, size))
    os.makedirs(out_path, exist_ok=True)
    image.save(os.path.join(out_path, "source.png"))
    images = edit_one(pipe, editor, invertor, image, [args.source_prompt], [args.target_prompt], args.inversion_type,
                      edit_type, device)
    save_img(images[0], os.path.join(out_path, "inversion.png"))
    save_img(images[1], os.path.join(out_path, "edit.png"))


if __name__ == "__main__":
    main()
