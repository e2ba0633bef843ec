"""PIE-Bench driver — `/root/reference/p2p/test.py:114-181`, sharded over the GPUs of one node.

Per image: invert (50 DDIM steps, UNet batch 1) -> edit-type rule (equal word counts => "replace", else
"refine", :120-123) -> edit (50 steps, batch 4) -> `source.png / inversion.png / edit.png` under
`./test_exp/<relpath>` -> `controller.reset()` + `unregister_attention_control` (:180-181).

Multi-GPU: images are independent, so rank r of W takes items i with i % W == r of the concatenated
category lists [0,1,2,3,4,6,7,8,9] (category 5 is skipped by the reference, :114) and writes its own
files; no collective touches the data path.  Launch with
    python -m torch.distributed.run --nproc-per-node W test.py [...]
The only collective is the final all_reduce of the image counters / time for the images/sec report.

`--synthetic N` replaces the (unavailable) PIE download by N generated images so the loop can be timed.
`--invert_batch K` inverts K images of a rank's shard in ONE batched DDIM loop (UNet batch K
instead of K loops at batch 1; images are independent, results are those of the per-image loop) before editing them one
by one: at batch 1 the UNet is bound by per-kernel latency, so K = 4 nearly quarters the inversion time per image.
`--in_flight E` keeps E images in flight on the GPU, for the edits (`P2P.edit_many`) and for the null-text
optimisations (`NTI.null_optimization_many`): a step is hundreds of dependent launches, and E independent chains fill
each other's dispatch gaps.  Both change the schedule, not the results.
"""
import argparse
import json
import os
import time

import torch
from PIL import Image

from _bootstrap import load_pipe, seed_everything

from edit_real import edit_latent, edit_one
from ief_amd.p2p.model.attention_control import AttentionRefine, AttentionReplace
from ief_amd.p2p.dataset.pie import PIE, SyntheticPIE
from ief_amd.p2p.inversion.ddim import ddim_inversion, ddim_inversion_xl
from ief_amd.p2p.inversion.nti import NTI, NTI_XL
from ief_amd.p2p.model.sd_utils import P2P, P2P_NTI, P2P_XL, P2P_XL_NTI
from ief_amd.p2p.utils.save_image import PngWriter

CATEGORIES = [0, 1, 2, 3, 4, 6, 7, 8, 9]


def shard(n_items: int, rank: int, world: int):
    """indices of the items rank `rank` of `world` processes: i = rank (mod world)"""
    return list(range(rank, n_items, world))


def edit_type_of(source_prompt: str, target_prompt: str) -> str:
    return "replace" if len(source_prompt.split(" ")) == len(target_prompt.split(" ")) else "refine"


def run_items(pipe, editor, invertor, items, size, device, inversion_type="ddim", invert_batch=1, in_flight=1, save=None):
    """the per-image loop of `/root/reference/p2p/test.py:116-181` over `items` = [(image path, source prompt, target
    prompt)]; `save(image_path, original PIL image, uint8 images [2,H,W,3])` is called per image.  invert_batch /
    in_flight = 1 is the reference's order (invert one image, edit it, next); see the module docstring for the others.
    Also what `bench.py` times for its images/sec figures."""
    save = save or (lambda *a: None)
    nti = inversion_type == "null-text"
    bs = max(1, invert_batch)
    E = max(1, in_flight)
    group = max(bs, E)
    for c0 in range(0, len(items), group):
        chunk = items[c0:c0 + group]
        originals = [Image.open(path).convert("RGB").resize((size, size)) for path, _, _ in chunk]
        if group == 1:
            image_path, source_prompt, target_prompt = chunk[0]
            images = edit_one(pipe, editor, invertor, originals[0], [source_prompt], [target_prompt], inversion_type,
                              edit_type_of(source_prompt, target_prompt), device)
            save(image_path, originals[0], images)
            continue
        # inversion: batched over `bs` images at a time (the DDIM loop is the same for both inversion types)
        traj, ctxs = [], []
        for b0 in range(0, len(chunk), bs):
            latent = torch.cat([invertor.image2latent(model=pipe, image=im, device=device, dtype=torch.float32)
                                for im in originals[b0:b0 + bs]])
            latents, context = invertor.ddim_inversion_loop(pipe, latent, [src for _, src, _ in chunk[b0:b0 + bs]])
            k = latent.shape[0]
            for j in range(k):
                traj.append([t[j:j + 1].clone() for t in latents])
                if isinstance(context, tuple):       # SDXL family: (prompt, negative, pooled, negative pooled) embeddings
                    ctxs.append(tuple(t[j:j + 1] for t in context))
                else:
                    unc, cnd = context.chunk(2)
                    ctxs.append(torch.cat([unc[j:j + 1], cnd[j:j + 1]]))
        x_T = [t[-1] for t in traj]
        # null-text optimisation: E images in flight
        uncond = [None] * len(chunk)
        if nti:
            for e0 in range(0, len(chunk), E):
                part = list(range(e0, min(e0 + E, len(chunk))))
                outs = invertor.null_optimization_many(pipe, [traj[j] for j in part], [ctxs[j] for j in part], 10, 1e-5, 7.5)
                for j, o in zip(part, outs):
                    uncond[j] = o
        # edits: E in flight
        for e0 in range(0, len(chunk), E):
            part = list(range(e0, min(e0 + E, len(chunk))))
            if len(part) == 1:
                j = part[0]
                _, src, tgt = chunk[j]
                extra = {"uncond_embeddings_list": uncond[j]} if nti else None
                results = [edit_latent(pipe, editor, x_T[j], [src], [tgt], edit_type_of(src, tgt), device, extra)]
            else:
                jobs, ctrls = [], []
                for j in part:
                    _, src, tgt = chunk[j]
                    cls = AttentionReplace if edit_type_of(src, tgt) == "replace" else AttentionRefine
                    ctrl = cls(prompts=[src, tgt], tokenizer=pipe.tokenizer, num_steps=50, cross_replace_steps=0.8,
                               self_replace_steps=0.6, device=device)
                    ctrls.append(ctrl)
                    jobs.append(([src, tgt], ctrl, x_T[j]) + ((uncond[j],) if nti else ()))
                results = [im for im, _ in editor.edit_many(pipe, jobs, num_inference_steps=50, guidance_scale=7.5)]
                for ctrl in ctrls:
                    ctrl.reset()
            for j, images in zip(part, results):
                save(chunk[j][0], originals[j], images)


def main(argv=None):
    ap = argparse.ArgumentParser("PIE-Bench P2P")
    ap.add_argument("--sd_version", type=str, default="1.5")
    ap.add_argument("--dataset_path", type=str, default="./PIE")
    ap.add_argument("--exp_path", type=str, default="./test_exp")
    ap.add_argument("--inversion_type", type=str, default="ddim")
    ap.add_argument("--synthetic", type=int, default=0, help="use N generated images instead of ./PIE")
    ap.add_argument("--no_save", action="store_true")
    ap.add_argument("--invert_batch", type=int, default=1, help="images inverted per batched DDIM loop")
    ap.add_argument("--precision", type=str, default=os.environ.get("IEF_PRECISION", "f16x3"), choices=["f16", "f32", "f16x3"],
                    help="f16x3 (default): fp32 storage, contractions on split fp16 operands -- the reference's fp32 images to 4e-6; "
                         "f32: the same on the fp32-input MFMA; f16: fp16 storage, 2.5x faster, images within 2 grey levels")
    ap.add_argument("--in_flight", type=int, default=1,
                    help="independent images stepped concurrently on one GPU (null-text optimisations and edits)")
    args = ap.parse_args(argv)

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device(f"cuda:{local % max(1, torch.cuda.device_count())}")   # ranks beyond the device count share GPUs (gloo rehearsals)
    torch.cuda.set_device(device)
    if world > 1:
        from _bootstrap import init_distributed
        dist = init_distributed(device)       # rank 0 loads the weights; `load_pipe` broadcasts them (RCCL over xGMI)
    seed_everything(42)
    pipe = load_pipe(args.sd_version, device, precision=args.precision)
    xl = pipe.__class__.__name__ == "StableDiffusionXLPipeline"          # dispatch of test.py:86-104
    if args.inversion_type == "ddim":
        editor = (P2P_XL if xl else P2P)(model=pipe, num_inference_steps=50)
        invertor = ddim_inversion_xl() if xl else ddim_inversion()
    elif args.inversion_type == "null-text":
        editor = (P2P_XL_NTI if xl else P2P_NTI)(model=pipe, num_inference_steps=50)
        invertor = NTI_XL() if xl else NTI()
    else:
        raise ValueError("Please choose right inversion type")
    size = pipe.unet.config.sample_size * pipe.vae_scale_factor

    if args.synthetic > 0:
        items = list(SyntheticPIE(os.path.join(args.exp_path, "_synthetic_inputs"), args.synthetic, size=size).items)
        root = os.path.join(args.exp_path, "_synthetic_inputs")
    else:
        items, root = [], os.path.join(args.dataset_path, "annotation_images")
        for category in CATEGORIES:
            items += PIE(args.dataset_path, None, category=category).items
    mine = shard(len(items), rank, world)
    torch.cuda.synchronize()
    t0 = time.perf_counter()

    # PNG encoding on host threads: the GPU loop never waits for a file; the context manager drains the pool when the loop
    # raises too, WITHOUT letting a failed write replace the loop's own exception (PngWriter.__exit__)
    with PngWriter() as writer:

        def save(image_path, original, images):
            if args.no_save:
                return
            out_path = os.path.join(args.exp_path, os.path.relpath(image_path.split(".")[0], root))
            os.makedirs(out_path, exist_ok=True)
            writer.save_pil(original, os.path.join(out_path, "source.png"))
            writer.save_img(images[0], os.path.join(out_path, "inversion.png"))
            writer.save_img(images[1], os.path.join(out_path, "edit.png"))

        run_items(pipe, editor, invertor, [items[i] for i in mine], size, device, args.inversion_type, args.invert_batch,
                  args.in_flight, save)
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
