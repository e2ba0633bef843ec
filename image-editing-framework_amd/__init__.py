"""MI355X-native attention-controlled diffusion editing engine (hot path only).

Layout
  csrc/         hand-written HIP kernels for gfx950 + the C-ABI (`include/ief_hip.h`)
  hip.py        ctypes binding of that C-ABI (fails loudly when the library is absent)
  unet.py       diffusers-shaped module tree whose forwards launch the HIP kernels
  scheduler.py  DDIM scheduler with the reference's config
  pipeline.py   duck-typed `StableDiffusionPipeline` the reference's editors expect
  p2p/          host-side mirror of `/root/reference/p2p` (controllers, hooks, loops, CLIs)

Import as ``import ief_amd`` (see ``ief_amd.py`` at the repo root).
"""
from .config import UNetConfig, SD15, TINY, SMALL, CONFIGS, SCHEDULER_CONFIG  # noqa: F401

__all__ = ["UNetConfig", "SD15", "TINY", "SMALL", "CONFIGS", "SCHEDULER_CONFIG"]
