"""`sd_version -> model location`, the surface of `/root/reference/p2p/sd_mapping.py:1-6`.

The reference maps versions to hub names and tells users to edit this file to local paths
(`/root/reference/README.md:30-32`).  Here every value is either a LOCAL diffusers-layout directory
(set the environment variable, or edit this file) or a `synthetic:<cfg>` key (seeded random weights
of that architecture; there are no checkpoints offline).
"""
import os

sd_maps = {
    "1.4": os.environ.get("IEF_SD14_DIR", "synthetic:sd15"),   # same architecture as 1.5
    "1.5": os.environ.get("IEF_SD15_DIR", "synthetic:sd15"),
    "xl-base": os.environ.get("IEF_SDXL_DIR", "synthetic:sdxl"),
    "2.1": os.environ.get("IEF_SD21_DIR", "synthetic:sd21"),
    "tiny": "synthetic:tiny",
    "small": "synthetic:small",
    "small21": "synthetic:small21",
    "smallxl": "synthetic:smallxl",
}
