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
    precision: "f16" (default; fp16 storage, fp32 accumulation) or "f32" — the reference's own precision (its CLIs load the
    pipeline with torch_dtype=float32, :38): fp32 weights / activations on the fp32-MFMA kernels.  `--precision` of the
    CLIs, or the IEF_PRECISION environment variable for the folders whose CLIs do not carry the flag."""
    precision = precision or os.environ.get("IEF_PRECISION", "f16")
    from ief_amd.pipeline import StableDiffusionPipeline
    from ief_amd.scheduler import DDIMScheduler
    from ief_amd.p2p.sd_mapping import sd_maps
    model_key = sd_maps[sd_version]          # KeyError for unknown versions, as in the reference (:26)
    scheduler = DDIMScheduler.from_config(SCHEDULER_CONFIG)
    if sd_version in ("1.5", "1.4", "2.1", "tiny", "small", "small21"):
        return StableDiffusionPipeline.from_pretrained(model_key, torch_dtype=dtype, scheduler=scheduler, device=device,
                                                       precision=precision)
    if sd_version in ("xl-base", "smallxl"):       # `StableDiffusionXLPipeline` branch of edit_syn.py:63-65
        from ief_amd.pipeline import StableDiffusionXLPipeline
        return StableDiffusionXLPipeline.from_pretrained(model_key, torch_dtype=dtype, scheduler=scheduler, device=device,
                                                         precision=precision)
    raise ValueError("please use the right sd_version")
