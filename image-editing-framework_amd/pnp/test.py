"""PIE-Bench driver for Plug-and-Play — `/root/reference/pnp/test.py:112-` (ddim and null-text branches), sharded over the GPUs of
one node exactly as `p2p/test.py`: rank r of W takes items i with i % W == r, no collective on the data path.

Per image: DDIM inversion under the source prompt (50 steps, UNet batch 1) -> PnP sampler from
`latents = cat([x_T, x_T])` (50 steps, batch 4) -> `source.png / inversion.png / edit.png` under `./test_exp/<relpath>`.
`--synthetic N` replaces the (unavailable) PIE download; `--invert_batch K` inverts K images per batched DDIM loop.
Config 4 of BASELINE.json (`--sd_version 2.1`: 768x768, 96x96 latents, head dim 64, context 1024) runs the same code.
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
from ief_amd.p2p.utils.save_image import PngWriter  # noqa: E402
from ief_amd.p2p.inversion.nti import NTI, NTI_XL_5e2 as NTI_XL  # noqa: E402  (this folder's copy: lr 5e-2)
from ief_amd.pnp.model.sd_utils import PnP, PnP_NTI, PnP_XL, PnP_XL_NTI  # noqa: E402

CATEGORIES = [0, 1, 2, 3, 4, 6, 7, 8, 9]


def main(argv=None):
    ap = argparse.ArgumentParser("PIE-Bench PnP")
    ap.add_argument("--sd_version", type=str, default="1.5")
    ap.add_argument("--dataset_path", type=str, default="./PIE")
    ap.add_argument("--exp_path", type=str, default="./test_exp")
    ap.add_argument("--inversion_type", type=str, default="ddim")
    ap.add_argument("--synthetic", type=int, default=0, help="use N generated images instead of ./PIE")
    ap.add_argument("--no_save", action="store_true")
    ap.add_argument("--invert_batch", type=int, default=1)
    ap.add_argument("--in_flight", type=int, default=1, help="edits stepped concurrently on one GPU (PnP.edit_many)")
    args = ap.parse_args(argv)
    if args.inversion_type not in ("ddim", "null-text"):
        raise ValueError("--inversion_type must be ddim or null-text")
    nti = args.inversion_type == "null-text"
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device(f"cuda:{local % max(1, torch.cuda.device_count())}")   # ranks beyond the device count share GPUs (gloo rehearsals)
    torch.cuda.set_device(device)
    if world > 1:
        from _bootstrap import init_distributed
        dist = init_distributed(device)       # rank 0 loads the weights; `load_pipe` broadcasts them (RCCL over xGMI)
    seed_everything(42)
    pipe = load_pipe(args.sd_version, device)
    num_inference_steps, guidance_scale, pnp_attn_t, pnp_f_t = 50, 7.5, 1.0, 1.0
    num_inner_steps, early_stop_epsilon = 10, 1e-5
    xl = pipe.__class__.__name__ == "StableDiffusionXLPipeline"          # the XL branch of the reference's dispatch
    if xl:
        invertor = NTI_XL() if nti else ddim_inversion_xl()
        editor = (PnP_XL_NTI if nti else PnP_XL)(pipe, num_inference_steps)
    else:
        invertor = NTI() if nti else ddim_inversion()
        editor = (PnP_NTI if nti else PnP)(pipe, num_inference_steps)
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
    E = max(1, args.in_flight)
    if E > 1:
        bs = max(bs, E)
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
            if E > 1 and not nti:       # the chunk's edits in flight, E at a time; same images as the per-image calls below
                for e0 in range(0, len(chunk), E):
                    part = list(range(e0, min(e0 + E, len(chunk))))
                    jobs = []
                    for j in part:
                        x_T = latents[-1][j:j + 1].clone()
                        jobs.append(([chunk[j][1], chunk[j][2]], torch.cat([x_T, x_T])))
                    outs = editor.edit_many(jobs, num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                                            pnp_attn_t=pnp_attn_t, pnp_f_t=pnp_f_t)
                    for j, images in zip(part, outs):
                        if not args.no_save:
                            out_path = os.path.join(args.exp_path, os.path.relpath(chunk[j][0].split(".")[0], root))
                            os.makedirs(out_path, exist_ok=True)
                            writer.save_pil(originals[j], os.path.join(out_path, "source.png"))
                            writer.save_img(images[0], os.path.join(out_path, "inversion.png"))
                            writer.save_img(images[1], os.path.join(out_path, "edit.png"))
                continue
            for j, (image_path, source_prompt, target_prompt) in enumerate(chunk):
                x_T = latents[-1][j:j + 1].clone()
                extra = {}
                if nti:         # `PnP_NTI` (`/root/reference/pnp/test.py:132-`): null-text optimisation of this image first
                    lat_j = [l[j:j + 1].clone() for l in latents]
                    ctx_j = (tuple(c[j:j + 1] for c in context) if xl else
                             torch.cat([context[j:j + 1], context[len(chunk) + j:len(chunk) + j + 1]]))
                    extra["uncond_embeddings_list"] = invertor.null_optimization(pipe, lat_j, ctx_j, num_inner_steps,
                                                                                 early_stop_epsilon, guidance_scale)
                images = editor(prompt=[source_prompt] + [target_prompt], num_inference_steps=num_inference_steps,
                                guidance_scale=guidance_scale, pnp_attn_t=pnp_attn_t, pnp_f_t=pnp_f_t,
                                latents=torch.cat([x_T, x_T]), **extra)
                if not args.no_save:
                    out_path = os.path.join(args.exp_path, os.path.relpath(image_path.split(".")[0], root))
                    os.makedirs(out_path, exist_ok=True)
                    writer.save_pil(originals[j], os.path.join(out_path, "source.png"))
                    writer.save_img(images[0], os.path.join(out_path, "inversion.png"))
                    writer.save_img(images[1], os.path.join(out_path, "edit.png"))
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
