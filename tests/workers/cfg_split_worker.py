"""One rank of a 2-rank CFG split of ONE Prompt-to-Prompt edit (launched by tests/test_gpu_cli.py under
`python -m torch.distributed.run --nproc-per-node 2`).  Both ranks may share one GPU: the eps exchange then goes over a
gloo group (host staging); on two GPUs the same code runs over RCCL (`--backend nccl`)."""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ief_amd  # noqa: E402,F401
from ief_amd.pipeline import StableDiffusionPipeline  # noqa: E402
from ief_amd.p2p.model.attention_control import AttentionRefine  # noqa: E402
from ief_amd.p2p.model.register import unregister_attention_control  # noqa: E402
from ief_amd.p2p.model.sd_utils import P2P  # noqa: E402

PROMPTS = ["a photo of a house on a mountain", "a photo of a house on a mountain at fall"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--config", default="tiny")
    ap.add_argument("--steps", type=int, default=6)
    args = ap.parse_args()
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    dev = torch.device(f"cuda:{local % ndev}")
    torch.cuda.set_device(dev)
    dist.init_process_group(args.backend)
    pipe = StableDiffusionPipeline.from_pretrained(f"synthetic:{args.config}", device=str(dev))
    cfg = pipe.cfg
    n = args.steps
    editor = P2P(pipe, n)
    x_T = torch.randn(1, 4, cfg.sample_size, cfg.sample_size, generator=torch.Generator().manual_seed(8888))
    c = AttentionRefine(PROMPTS, pipe.tokenizer, n, 0.8, 0.4, device=dev)
    split, _ = editor.text2image_ldm_stable(pipe, PROMPTS, c, num_inference_steps=n, guidance_scale=7.5, latent=x_T.to(dev),
                                            return_latents=True, cfg_split_group=dist.group.WORLD)
    steps_seen = c.cur_step
    unregister_attention_control(pipe, c)
    # the same edit on ONE rank's full CFG batch
    c2 = AttentionRefine(PROMPTS, pipe.tokenizer, n, 0.8, 0.4, device=dev)
    full, _ = editor.text2image_ldm_stable(pipe, PROMPTS, c2, num_inference_steps=n, guidance_scale=7.5, latent=x_T.to(dev),
                                           return_latents=True)
    unregister_attention_control(pipe, c2)
    both = [None, None]
    dist.all_gather_object(both, split.cpu())
    rel = ((split - full).abs().max() / full.abs().max()).item()
    if rank == 0:
        print(json.dumps({"rel_split_vs_full": rel, "ranks_identical": bool(torch.equal(both[0], both[1])),
                          "cur_step": steps_seen, "finite": bool(torch.isfinite(split).all())}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
