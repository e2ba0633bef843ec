"""UNet / scheduler configurations for the attention-controlled denoising path.

The reference never spells the architecture out: it loads it by hub name through diffusers
(`/root/reference/p2p/sd_mapping.py:1-6`, `/root/reference/p2p/edit_syn.py:60`).  The numbers
below restate the public SD1.5 `unet/config.json` and the scheduler
dict every reference script hard-codes (`/root/reference/p2p/edit_syn.py:46-57`).
"""
from dataclasses import dataclass, field
from typing import Tuple


@dataclass(frozen=True)
class UNetConfig:
    sample_size: int = 64
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    # True where the down block (and the mirrored up block) carries transformers
    down_has_attn: Tuple[bool, ...] = (True, True, True, False)
    layers_per_block: int = 2
    cross_attention_dim: int = 768
    # diffusers' misnamed `attention_head_dim`: for SD1.5 it is the NUMBER of heads
    num_heads: Tuple[int, ...] = (8, 8, 8, 8)
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    time_embed_dim_mult: int = 4
    text_max_length: int = 77
    # SD2.x: Transformer2DModel.proj_in / proj_out are nn.Linear on tokens ([C, C] weights) instead of 1x1 convs
    # ([C, C, 1, 1]); the arithmetic is the same GEMM, only the checkpoint's tensor shapes differ
    use_linear_projection: bool = False
    # SDXL: BasicTransformerBlocks per Transformer2DModel, per level (empty = one everywhere); the mid block takes the last
    transformer_layers: Tuple[int, ...] = ()
    # SDXL `addition_embed_type = "text_time"`: emb += add_embedding(cat([pooled text embeds, sinusoids of the 6 time ids]))
    addition_embed: bool = False
    addition_time_embed_dim: int = 256
    pooled_text_dim: int = 1280

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * self.time_embed_dim_mult

    def depth(self, level: int) -> int:
        return self.transformer_layers[level] if self.transformer_layers else 1

    @property
    def addition_input_dim(self) -> int:
        """`projection_class_embeddings_input_dim`: pooled text + 6 time ids x addition_time_embed_dim (2816 for SDXL)"""
        return self.pooled_text_dim + 6 * self.addition_time_embed_dim


SD15 = UNetConfig()

# A shape family small enough for the CPU oracle to run a whole edit in seconds.  Channel
# counts stay multiples of 64 (the implicit-GEMM K-tile) and head dims hit 64 and 32.
TINY = UNetConfig(
    sample_size=16,
    block_out_channels=(64, 128, 128, 128),
    down_has_attn=(True, True, True, False),
    cross_attention_dim=64,
    num_heads=(1, 2, 4, 4),
    text_max_length=77,
)

# SD1.5 head geometry (d = 40 / 80, C = 320 / 640) on a two-level net: the real channel
# widths, skip concats and the N<=256 self-replace rule without the full 860 M parameters.
SMALL = UNetConfig(
    sample_size=32,
    block_out_channels=(320, 640),
    down_has_attn=(True, True),
    cross_attention_dim=768,
    num_heads=(8, 8),
)

# SD2.1 (768-v) shape family — the public `unet/config.json` of stabilityai/stable-diffusion-2-1
# (`/root/reference/pnp/sd_mapping.py:4`): 96x96 latents (768x768 px), OpenCLIP context 1024, head dim 64
# (attention_head_dim = [5, 10, 20, 20] = number of heads), linear projections.  865,910,724 parameters.
SD21 = UNetConfig(sample_size=96, cross_attention_dim=1024, num_heads=(5, 10, 20, 20), use_linear_projection=True)

# SD2.1 head geometry (d = 64) on the small two-level net
SMALL21 = UNetConfig(sample_size=32, block_out_channels=(320, 640), down_has_attn=(True, True), cross_attention_dim=1024,
                     num_heads=(5, 10), use_linear_projection=True)

# SDXL base shape family — the public `unet/config.json` of stabilityai/stable-diffusion-xl-base-1.0
# (`/root/reference/pix2pix-zero/sd_mapping.py:2`, BASELINE.json config 5): 128x128 latents (1024x1024 px), three levels,
# no attention at the first, transformer depth 1 / 2 / 10, head dim 64 (attention_head_dim = [5, 10, 20]), context 2048
# (two text encoders concatenated), linear projections, text-time additional embedding.  2,567,463,684 parameters.
SDXL = UNetConfig(sample_size=128, block_out_channels=(320, 640, 1280), down_has_attn=(False, True, True),
                  cross_attention_dim=2048, num_heads=(5, 10, 20), use_linear_projection=True,
                  transformer_layers=(1, 2, 10), addition_embed=True)

# SDXL geometry (no attention at level 0, depth > 1, d = 64, additional embedding) small enough for the CPU oracle
SMALLXL = UNetConfig(sample_size=16, block_out_channels=(64, 128, 256), down_has_attn=(False, True, True),
                     cross_attention_dim=128, num_heads=(1, 2, 4), use_linear_projection=True,
                     transformer_layers=(1, 2, 3), addition_embed=True, addition_time_embed_dim=32, pooled_text_dim=64)

SCHEDULER_CONFIG = {
    "beta_end": 0.012,
    "beta_schedule": "scaled_linear",
    "beta_start": 0.00085,
    "clip_sample": False,
    "num_train_timesteps": 1000,
    "set_alpha_to_one": False,
    "skip_prk_steps": True,
    "steps_offset": 1,
    "trained_betas": None,
    "use_karras_sigmas": False,
}

CONFIGS = {"sd15": SD15, "tiny": TINY, "small": SMALL, "sd21": SD21, "small21": SMALL21, "sdxl": SDXL, "smallxl": SMALLXL}
