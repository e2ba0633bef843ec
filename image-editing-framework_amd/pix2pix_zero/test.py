"""PIE-Bench driver for Pix2Pix-zero — `/root/reference/pix2pix-zero/test.py:84-137`, sharded over the GPUs of one node
exactly as `p2p/test.py`: rank r of W takes items i with i % W == r, no collective on the data path.

Per image: DDIM inversion under the source prompt (50 steps, UNet batch 1; `--inversion_type null-text` adds the null-text
optimisation) -> `P2P_Zero` / `P2P_Zero_NTI` from the inverted latent (reference pass: 50 steps at batch 2 recording the
cross-attention maps; edit pass: 50 x {forward + reverse pass + forward} at batch 2) -> `source.png / inversion.png /
edit.png` under `./test_exp/<relpath>`.  `--synthetic N` replaces the (unavailable) PIE download; `--invert_batch K`
inverts K images per batched DDIM loop.
"""
import argparse
import json
import os
import sys
import time

import torch
from PIL import Image

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "p2p"))
from _bootstrap import load_pipe, seed_everything  # noqa: E402

from ief_amd.p2p.dataset.pie import PIE, SyntheticPIE  # noqa: E402
from ief_amd.p2p.inversion.ddim import ddim_inversion, ddim_inversion_xl  # noqa: E402
from ief_amd.p2p.inversion.nti import NTI, NTI_XL_5e2 as NTI_XL  # noqa: E402  (this folder's copy: lr 5e-2)
from ief_amd.p2p.utils.save_image import PngWriter  # noqa: E402
from ief_amd.pix2pix_zero.model.sd_utils import P2P_Zero, P2P_Zero_NTI, P2P_Zero_XL, P2P_Zero_XL_NTI  # noqa: E402

CATEGORIES = [0, 1, 2, 3, 4, 6, 7, 8, 9]


def main(argv=None):
    ap = argparse.ArgumentParser("PIE-Bench Pix2Pix-zero")
    ap.add_argument("--sd_version", type=str, default="1.5")
    ap.add_argument("--dataset_path", type=str, default="./PIE")
    ap.add_argument("--exp_path", type=str, default="./test_exp")
    ap.add_argument("--inversion_type", type=str, default="ddim")
    ap.add_argument("--synthetic", type=int, default=0, help="use N generated images instead of ./PIE")
    ap.add_argument("--no_save", action="store_true")
    ap.add_argument("--invert_batch", type=int, default=1)
    args = ap.parse_args(argv)
    if args.inversion_type not in ("ddim", "null-text"):
        raise ValueError("--inversion_type must be ddim or null-text")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device(f"cuda:{local % max(1, torch.cuda.device_count())}")   # ranks beyond the device count share GPUs (gloo rehearsals)
    torch.cuda.set_device(device)
    if world > 1:
        from _bootstrap import init_distributed
        dist = init_distributed(device)       # rank 0 loads the weights; `load_pipe` broadcasts them (RCCL over xGMI)
    seed_everything(42)
    pipe = load_pipe(args.sd_version, device)
    num_inference_steps, guidance_scale = 50, 7.5
    num_inner_steps, early_stop_epsilon = 10, 1e-5
    nti = args.inversion_type == "null-text"
    xl = pipe.__class__.__name__ == "StableDiffusionXLPipeline"          # dispatch of test.py:84-103
    if xl:          # BASELINE.json config 5: SDXL, 1024x1024
        invertor = NTI_XL() if nti else ddim_inversion_xl()
        editor = (P2P_Zero_XL_NTI if nti else P2P_Zero_XL)(pipe, num_inference_steps)
    else:
        invertor = NTI() if nti else ddim_inversion()
        editor = (P2P_Zero_NTI if nti else P2P_Zero)(pipe, num_inference_steps)
    size = pipe.unet.config.sample_size * pipe.vae_scale_factor
    if args.synthetic > 0:
        root = os.path.join(args.exp_path, "_synthetic_inputs")
        items = list(SyntheticPIE(root, args.synthetic, size=size).items)
    else:
        items, root = [], os.path.join(args.dataset_path, "annotation_images")
        for category in CATEGORIES:
            items += PIE(args.dataset_path, None, category=category).items
    mine = list(range(rank, len(items), world))
    bs = max(1, args.invert_batch)
    # PNG encoding on host threads: the GPU loop never waits for a file; the context manager drains the pool when the loop
    # raises too, WITHOUT letting a failed write replace the loop's own exception (PngWriter.__exit__)
    with PngWriter() as writer:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c0 in range(0, len(mine), bs):
            chunk = [items[i] for i in mine[c0:c0 + bs]]
            originals = [Image.open(path).convert("RGB").resize((size, size)) for path, _, _ in chunk]
            latent = torch.cat([invertor.image2latent(model=pipe, image=im, device=device, dtype=torch.float32) for im in originals])
            latents, context = invertor.ddim_inversion_loop(pipe, latent, [src for _, src, _ in chunk])
            for j, (image_path, source_prompt, target_prompt) in enumerate(chunk):
                extra = {}
                if nti:
                    lat_j = [l[j:j + 1].clone() for l in latents]
                    if xl:      # the encoder's 4-tuple, one row per image
                        ctx_j = tuple(c[j:j + 1] for c in context)
                    else:
                        ctx_j = torch.cat([context[j:j + 1], context[len(chunk) + j:len(chunk) + j + 1]])
                    extra["uncond_embeddings_list"] = invertor.null_optimization(pipe, lat_j, ctx_j, num_inner_steps,
                                                                                 early_stop_epsilon, guidance_scale)
                image_source, image_edit = editor(prompt=[source_prompt] + [target_prompt],
                                                  num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                                                  only_sample=False, edit_dir=None, latents=latents[-1][j:j + 1].clone(), **extra)
                if not args.no_save:
                    out_path = os.path.join(args.exp_path, os.path.relpath(image_path.split(".")[0], root))
                    os.makedirs(out_path, exist_ok=True)
                    writer.save_pil(originals[j], os.path.join(out_path, "source.png"))
                    writer.save_img(image_source, os.path.join(out_path, "inversion.png"))
                    writer.save_img(image_edit, os.path.join(out_path, "edit.png"))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = torch.tensor([float(len(mine)), dt], device=device)
    if world > 1:
        cnt = n[:1].clone()
        dist.all_reduce(cnt)
        tmax = n[1:].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        n = torch.cat([cnt, tmax])
    if rank == 0:
        print(json.dumps({"images": int(n[0].item()), "seconds": round(n[1].item(), 3),
                          "images_per_sec": round(n[0].item() / max(n[1].item(), 1e-9), 4), "n_gpus": world}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
