"""Makes `python edit_syn.py` / `edit_real.py` / `test.py` runnable from inside this folder, the way the
reference's scripts are run (`/root/reference/README.md:36-58`): puts the repo root on sys.path and loads
the hyphen-named package as `ief_amd`."""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import ief_amd  # noqa: E402,F401

SCHEDULER_CONFIG = dict(ief_amd.SCHEDULER_CONFIG)


def seed_everything(seed: int):
    """stands in for `lightning.pytorch.seed_everything` (`/root/reference/p2p/edit_syn.py:32`)"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def load_pipe(sd_version: str, device: torch.device, dtype=torch.float32, precision=None):
    """the loader switch of `/root/reference/p2p/edit_syn.py:58-86` for the versions this tier supports.
    precision (`--precision` of the P2P CLIs, the IEF_PRECISION environment variable for the folders whose CLIs do not carry the
    flag): the reference computes in fp32 in EVERY folder (`/root/reference/p2p/edit_syn.py:38`, `masactrl/edit_syn.py:38`,
    `pnp/edit_syn.py:39-40`, `pix2pix-zero/model/sd_utils.py:28`), so the default of all four folders is "f16x3" -- fp32
    storage, every contraction on split fp16 operands: the fastest mode inside north_star's 1e-3 image bound; "f32" = the
    fp32-input MFMA; "f16" = fp16 storage (3e-3 on a 50-step image: outside the bound, 2x the speed).  Exception: the SDXL
    family defaults to "f16", the precision BASELINE.json states for its configuration 5 (measured image error: DESIGN.md §4)."""
    precision = precision or os.environ.get("IEF_PRECISION") or ("f16" if sd_version in ("xl-base", "smallxl") else "f16x3")
    # multi-GPU runs (the PIE drivers under torchrun): rank 0 loads / draws the weights, the others build the same module
    # tree from zeros and receive the packed tensors by ONE bucketed broadcast (RCCL over xGMI) -- `dist.broadcast_pipeline`
    import torch.distributed as tdist
    world = tdist.get_world_size() if tdist.is_available() and tdist.is_initialized() else 1
    empty = world > 1 and tdist.get_rank() != 0
    pipe = _build_pipe(sd_version, device, dtype, precision, empty)
    if world > 1:
        from ief_amd.dist import broadcast_pipeline
        pipe._broadcasts = broadcast_pipeline(pipe, src=0)
        if device.type == "cuda":
            torch.cuda.synchronize()
    return pipe


def init_distributed(device):
    """process group of a PIE driver started by torchrun: RCCL ("nccl") unless IEF_DIST_BACKEND says otherwise (gloo lets
    several ranks share ONE GPU: rehearsals of the multi-rank path on a one-GPU box)"""
    import torch.distributed as tdist
    backend = os.environ.get("IEF_DIST_BACKEND", "nccl")
    if backend == "nccl":
        tdist.init_process_group("nccl", device_id=device)
    else:
        tdist.init_process_group(backend)
    return tdist


def _build_pipe(sd_version, device, dtype, precision, empty):
    from ief_amd.pipeline import StableDiffusionPipeline
    from ief_amd.scheduler import DDIMScheduler
    from ief_amd.p2p.sd_mapping import sd_maps
    model_key = sd_maps[sd_version]          # KeyError for unknown versions, as in the reference (:26)
    scheduler = DDIMScheduler.from_config(SCHEDULER_CONFIG)
    if sd_version in ("1.5", "1.4", "2.1", "tiny", "small", "small21"):
        return StableDiffusionPipeline.from_pretrained(model_key, torch_dtype=dtype, scheduler=scheduler, device=device,
                                                       precision=precision, empty_weights=empty)
    if sd_version in ("xl-base", "smallxl"):       # `StableDiffusionXLPipeline` branch of edit_syn.py:63-65
        from ief_amd.pipeline import StableDiffusionXLPipeline
        return StableDiffusionXLPipeline.from_pretrained(model_key, torch_dtype=dtype, scheduler=scheduler, device=device,
                                                         precision=precision, empty_weights=empty)
    raise ValueError("please use the right sd_version")
