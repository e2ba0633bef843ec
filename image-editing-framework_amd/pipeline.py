"""Duck-typed `StableDiffusionPipeline` — the object the reference's editors call `model` / `pipe`.

The reference dispatches on `pipe.__class__.__name__ == "StableDiffusionPipeline"`
(`/root/reference/p2p/edit_syn.py:90`) and touches only the attributes listed in SURVEY.md §8b:
`tokenizer`, `text_encoder`, `unet`, `vae`, `scheduler`, `device`, `_execution_device`,
`vae_scale_factor`.  `from_pretrained(key, torch_dtype=..., scheduler=...)` accepts

  "synthetic:<cfg>[:seed]"   seeded random weights of a named UNet config (no checkpoints exist
                             offline) — what tests, smoke and bench use;
  a local directory          diffusers layout (`unet/diffusion_pytorch_model.safetensors`, optional
                             `tokenizer/`, `text_encoder/`) — real weights when a user has them.
Hub names cannot be fetched here and raise a clear error.

The VAE is `vae.AutoencoderKL` (diffusers' architecture and key names on the HIP kernels).  The text
encoder runs once per image, off the per-step path: a local checkpoint loads transformers'
`CLIPTextModel` (PyTorch-ROCm); without one a small seeded stand-in with the same interface is used.
"""
import os
from types import SimpleNamespace

import torch
import torch.nn.functional as F
from torch import nn

from .config import CONFIGS, UNetConfig
from .scheduler import DDIMScheduler
from .tokenizer import WordPieceTokenizer
from .vae import AutoencoderKL, SD_VAE, TINY_VAE, VAEConfig
from . import weights as _weights


class SyntheticTextEncoder(nn.Module):
    """token ids [B,77] -> ([B,77,C] fp32,): seeded embedding + position table, one mixing layer."""

    def __init__(self, dim: int, vocab: int = 49408, max_len: int = 77, seed: int = 1):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.register_buffer("tok", torch.randn(vocab, dim, generator=g) * 0.1)
        self.register_buffer("pos", torch.randn(max_len, dim, generator=g) * 0.02)
        self.register_buffer("mix", torch.eye(max_len) * 0.9 + torch.full((max_len, max_len), 0.1 / max_len))
        self.dtype = torch.float32

    @property
    def device(self):
        return self.tok.device

    def forward(self, input_ids, **kw):
        x = self.tok[input_ids.to(self.tok.device)] + self.pos[None, : input_ids.shape[1]]
        x = torch.einsum("ts,bsc->btc", self.mix[: x.shape[1], : x.shape[1]], x)
        return (x,)


class StableDiffusionPipeline:
    def __init__(self, unet, tokenizer, text_encoder, vae, scheduler, cfg: UNetConfig, state_dict=None):
        self.unet, self.tokenizer, self.text_encoder, self.vae, self.scheduler = unet, tokenizer, text_encoder, vae, scheduler
        self.cfg = cfg
        self.vae_scale_factor = 8
        self._state_dict = state_dict  # kept on the host only when the caller asked (tests / oracle side)

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_pretrained(cls, model_key: str, torch_dtype=None, scheduler=None, device="cuda:0",
                        keep_state_dict: bool = False, precision: str = "f16", empty_weights: bool = False, **unused):
        """empty_weights: build the UNet from ZERO tensors of the right shapes without drawing / reading any weight -- what
        every rank but 0 of a multi-GPU run does before `dist.broadcast_pipeline` overwrites the packed tensors.
        precision: "f16" (default: fp16 storage / fp32 accumulation, the fast path whatever `torch_dtype` says) or
        "f32" — the reference's own precision (`/root/reference/p2p/edit_syn.py:38` loads the pipeline in fp32): UNet and
        VAE run on the fp32-MFMA kernels, edited images then agree with the fp32 reference to ~1e-4"""
        from .unet import UNet2DConditionModel
        if model_key.startswith("synthetic:"):
            parts = model_key.split(":")
            cfg = CONFIGS[parts[1]]
            seed = int(parts[2]) if len(parts) > 2 else 0
            sd = _zeros_state_dict(cfg) if empty_weights else _weights.synthetic_state_dict(cfg, seed)
            tokenizer = WordPieceTokenizer(cfg.text_max_length)
            text_encoder = SyntheticTextEncoder(cfg.cross_attention_dim)
            vae = AutoencoderKL(SD_VAE if parts[1] in ("sd15", "sd21") else TINY_VAE, device=device, precision=precision)
        elif os.path.isdir(model_key):
            cfg, sd = _load_local_unet(model_key, empty=empty_weights)
            tokenizer, text_encoder = _load_local_text(model_key, cfg)
            vae = _load_local_vae(model_key, device, precision)
        else:
            raise FileNotFoundError(
                f"'{model_key}' is neither 'synthetic:<cfg>' nor a local directory.  Hub names cannot be fetched "
                "offline: point sd_mapping.sd_maps at a local diffusers-layout directory (README of the reference, "
                "lines 30-32) or use 'synthetic:sd15'.")
        unet = UNet2DConditionModel(cfg, sd, device=device, precision=precision)
        text_encoder = text_encoder.to(device)
        sched = scheduler if scheduler is not None else DDIMScheduler()
        return cls(unet, tokenizer, text_encoder, vae, sched, cfg, sd if keep_state_dict else None)

    def to(self, device):
        if torch.device(device) != self.unet.device:
            raise RuntimeError("the UNet's packed weights were placed at construction; pass device= to from_pretrained")
        return self

    @property
    def device(self):
        return self.unet.device

    @property
    def _execution_device(self):
        return self.unet.device


class StableDiffusionXLPipeline(StableDiffusionPipeline):
    """The surface the reference's `*_XL` samplers use (`/root/reference/p2p/model/sd_utils.py:186-224`): two text
    encoders whose hidden states are concatenated into the 2048-wide context, the second one's pooled output, and
    `_get_add_time_ids`.  With synthetic weights both encoders are seeded stand-ins (off the per-step path)."""

    def __init__(self, unet, tokenizer, text_encoder, vae, scheduler, cfg, state_dict=None, text_encoder_2=None,
                 tokenizer_2=None):
        super().__init__(unet, tokenizer, text_encoder, vae, scheduler, cfg, state_dict)
        self.tokenizer_2, self.text_encoder_2 = tokenizer_2 or tokenizer, text_encoder_2

    @classmethod
    def from_pretrained(cls, model_key: str, torch_dtype=None, scheduler=None, device="cuda:0",
                        keep_state_dict: bool = False, precision: str = "f16", empty_weights: bool = False, **unused):
        from .unet import UNet2DConditionModel
        import dataclasses
        sched = scheduler if scheduler is not None else DDIMScheduler()
        if os.path.isdir(model_key):
            # a local diffusers-layout SDXL directory: unet/ and vae/ are required; text_encoder/ and text_encoder_2/ (CLIP
            # weights for `transformers`) and tokenizer/ are used when present, seeded stand-ins otherwise
            cfg, sd = _load_local_unet(model_key, empty=empty_weights)
            if not cfg.addition_embed:
                raise ValueError(f"{model_key}: unet/config.json is not an SDXL-family configuration")
            vae = _load_local_vae(model_key, device, precision)
            tokenizer, tokenizer_2, enc1, enc2 = _load_local_text_xl(model_key, cfg)
            unet = UNet2DConditionModel(cfg, sd, device=device, precision=precision)
            return cls(unet, tokenizer, enc1.to(device), vae, sched, cfg, sd if keep_state_dict else None,
                       text_encoder_2=enc2.to(device), tokenizer_2=tokenizer_2)
        if not model_key.startswith("synthetic:"):
            raise FileNotFoundError(
                f"'{model_key}' is neither 'synthetic:<cfg>' nor a local directory.  Hub names cannot be fetched offline: "
                "point sd_mapping.sd_maps['xl-base'] (or IEF_SDXL_DIR) at a local diffusers-layout directory, or use "
                "'synthetic:sdxl' (seeded weights of the public architecture)")
        parts = model_key.split(":")
        cfg = CONFIGS[parts[1]]
        if not cfg.addition_embed:
            raise ValueError(f"{parts[1]} is not an SDXL-family configuration")
        seed = int(parts[2]) if len(parts) > 2 else 0
        sd = _zeros_state_dict(cfg) if empty_weights else _weights.synthetic_state_dict(cfg, seed)
        d2 = cfg.pooled_text_dim
        tokenizer = WordPieceTokenizer(cfg.text_max_length)
        enc1 = SyntheticTextEncoder(cfg.cross_attention_dim - d2, seed=1).to(device)
        enc2 = SyntheticTextEncoder(d2, seed=2).to(device)
        vcfg = SD_VAE if parts[1] == "sdxl" else TINY_VAE
        vae = AutoencoderKL(dataclasses.replace(vcfg, scaling_factor=0.13025), device=device, precision=precision)   # the SDXL VAE's factor
        unet = UNet2DConditionModel(cfg, sd, device=device, precision=precision)
        return cls(unet, tokenizer, enc1, vae, sched, cfg, sd if keep_state_dict else None, text_encoder_2=enc2)

    def _hidden_and_pooled(self, texts):
        """-> (penultimate hidden states of both encoders concatenated [B,77,C1+C2], pooled output of the second [B,C2]),
        as diffusers' SDXL `encode_prompt` does with CLIP (`hidden_states[-2]`, `text_embeds`); the seeded stand-ins have
        one layer and no projection head: their output and its token mean"""
        outs = []
        pooled = None
        for k, (tok, enc) in enumerate(((self.tokenizer, self.text_encoder), (self.tokenizer_2, self.text_encoder_2))):
            ids = tok(texts, padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt").input_ids
            if isinstance(enc, SyntheticTextEncoder):
                h = enc(ids.to(self.device))[0]
                p = h.mean(dim=1)
            else:
                o = enc(ids.to(self.device), output_hidden_states=True)
                h, p = o.hidden_states[-2].float(), o[0].float()
            outs.append(h)
            if k == 1:
                pooled = p
        return torch.cat(outs, dim=-1), pooled

    @torch.no_grad()
    def encode_prompt(self, prompt, prompt_2=None, device=None, num_images_per_prompt=1, do_classifier_free_guidance=True,
                      negative_prompt=None, negative_prompt_2=None, prompt_embeds=None, negative_prompt_embeds=None,
                      pooled_prompt_embeds=None, negative_pooled_prompt_embeds=None, lora_scale=None, **unused):
        """-> (prompt_embeds [B,77,C1+C2], negative_prompt_embeds, pooled [B,C2], negative_pooled).  As SDXL base does
        (`force_zeros_for_empty_prompt`), a missing negative prompt gives ZERO embeddings, not the encoding of ""."""
        prompt = [prompt] if isinstance(prompt, str) else list(prompt)
        embeds, pooled = self._hidden_and_pooled(prompt)
        if negative_prompt is None:
            neg, neg_pooled = torch.zeros_like(embeds), torch.zeros_like(pooled)
        else:
            negs = [negative_prompt] * len(prompt) if isinstance(negative_prompt, str) else list(negative_prompt)
            neg, neg_pooled = self._hidden_and_pooled(negs)
        return embeds, neg, pooled, neg_pooled

    def _get_add_time_ids(self, original_size, crops_coords_top_left, target_size, dtype=torch.float32,
                          text_encoder_projection_dim=None):
        return torch.tensor([list(original_size + crops_coords_top_left + target_size)], dtype=dtype)


def _zeros_state_dict(cfg):
    """zero tensors with the UNet's parameter names and shapes (a rank that will RECEIVE the packed weights)"""
    return {k: torch.zeros(shape) for k, shape in _weights.unet_param_shapes(cfg).items()}


def _load_local_unet(path, empty=False):
    import json
    from safetensors.torch import load_file
    with open(os.path.join(path, "unet", "config.json")) as f:
        c = json.load(f)
    heads = c["attention_head_dim"]
    nlev = len(c["block_out_channels"])
    heads = tuple(heads) if isinstance(heads, (list, tuple)) else (heads,) * nlev
    cfg = UNetConfig(
        sample_size=c["sample_size"], in_channels=c["in_channels"], out_channels=c["out_channels"],
        block_out_channels=tuple(c["block_out_channels"]),
        down_has_attn=tuple("CrossAttn" in t for t in c["down_block_types"]),
        layers_per_block=c["layers_per_block"], cross_attention_dim=c["cross_attention_dim"], num_heads=heads,
        norm_num_groups=c["norm_num_groups"], norm_eps=c.get("norm_eps", 1e-5),
        use_linear_projection=bool(c.get("use_linear_projection", False)),
        transformer_layers=(tuple(c["transformer_layers_per_block"])
                            if isinstance(c.get("transformer_layers_per_block"), (list, tuple)) else ()),
        addition_embed=c.get("addition_embed_type") == "text_time",
        addition_time_embed_dim=c.get("addition_time_embed_dim") or 256,
        pooled_text_dim=(c["projection_class_embeddings_input_dim"] - 6 * (c.get("addition_time_embed_dim") or 256))
        if c.get("addition_embed_type") == "text_time" else 1280)
    if empty:
        return cfg, _zeros_state_dict(cfg)
    sd = load_file(os.path.join(path, "unet", "diffusion_pytorch_model.safetensors"))
    missing = [k for k in _weights.unet_param_shapes(cfg) if k not in sd]
    if missing:
        raise KeyError(f"checkpoint lacks {len(missing)} UNet tensors, e.g. {missing[:3]}")
    return cfg, {k: v.float() for k, v in sd.items()}


def _load_local_vae(path, device, precision="f16"):
    import json
    from safetensors.torch import load_file
    vdir = os.path.join(path, "vae")
    with open(os.path.join(vdir, "config.json")) as f:
        c = json.load(f)
    cfg = VAEConfig(block_out_channels=tuple(c["block_out_channels"]), layers_per_block=c["layers_per_block"],
                    latent_channels=c["latent_channels"], in_channels=c["in_channels"],
                    norm_num_groups=c["norm_num_groups"], scaling_factor=c.get("scaling_factor", 0.18215))
    sd = {k: v.float() for k, v in load_file(os.path.join(vdir, "diffusion_pytorch_model.safetensors")).items()}
    return AutoencoderKL(cfg, sd, device=device, precision=precision)


def _load_local_text_xl(path, cfg):
    """-> (tokenizer, tokenizer_2, text_encoder, text_encoder_2) of a local SDXL directory"""
    d2 = cfg.pooled_text_dim
    toks = []
    for name in ("tokenizer", "tokenizer_2"):
        tok_dir = os.path.join(path, name)
        if os.path.isdir(tok_dir):
            from transformers import CLIPTokenizer
            toks.append(CLIPTokenizer.from_pretrained(tok_dir))
        else:
            toks.append(toks[0] if toks else WordPieceTokenizer(cfg.text_max_length))
    tokenizer, tokenizer_2 = toks
    e1, e2 = os.path.join(path, "text_encoder"), os.path.join(path, "text_encoder_2")
    if os.path.isdir(e1) and os.path.isdir(e2):
        from transformers import CLIPTextModel, CLIPTextModelWithProjection
        enc1, enc2 = CLIPTextModel.from_pretrained(e1).eval(), CLIPTextModelWithProjection.from_pretrained(e2).eval()
        if enc1.config.hidden_size + enc2.config.hidden_size != cfg.cross_attention_dim or enc2.config.projection_dim != d2:
            raise ValueError("text encoders do not match the UNet: hidden sizes must add up to cross_attention_dim and the "
                             "second encoder's projection must be the pooled width")
        return tokenizer, tokenizer_2, enc1, enc2
    return tokenizer, tokenizer_2, SyntheticTextEncoder(cfg.cross_attention_dim - d2, seed=1), SyntheticTextEncoder(d2, seed=2)


def _load_local_text(path, cfg):
    tok_dir, te_dir = os.path.join(path, "tokenizer"), os.path.join(path, "text_encoder")
    if os.path.isdir(tok_dir) and os.path.isdir(te_dir):
        from transformers import CLIPTextModel, CLIPTokenizer
        return CLIPTokenizer.from_pretrained(tok_dir), CLIPTextModel.from_pretrained(te_dir)
    return WordPieceTokenizer(cfg.text_max_length), SyntheticTextEncoder(cfg.cross_attention_dim)
