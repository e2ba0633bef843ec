"""`sd_version -> model location` (`/root/reference/pix2pix-zero/sd_mapping.py:1-5`); same table as the P2P folder."""
from ..p2p.sd_mapping import sd_maps  # noqa: F401
