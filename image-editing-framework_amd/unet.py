"""SD-family conditional UNet whose every forward is a sequence of hand-written HIP kernels.

The module TREE is shaped like diffusers' `UNet2DConditionModel` because the reference's hooks
find their targets by class-name strings and attribute paths:
  - `register_attention_control` walks `unet.named_children()` for names containing
    "down" / "up" / "mid" and patches modules whose `__class__.__name__ == 'Attention'`
    (`/root/reference/p2p/model/register.py:78-96`);
  - the patched forward uses `to_q/to_k/to_v/to_out, head_to_batch_dim, get_attention_scores,
    batch_to_head_dim, prepare_attention_mask, spatial_norm, group_norm, norm_cross,
    residual_connection, rescale_output_factor` (register.py:5-62);
  - PnP addresses `up_blocks[r].attentions[b].transformer_blocks[0].attn1` and
    `up_blocks[1].resnets[1]` (`/root/reference/pnp/model/register.py:6-19,82-88`).
`torch.nn.Module` is used only for that tree bookkeeping (children, ModuleList); parameters are
plain device tensors already packed for the kernels:

  activations  fp16, channels-last  [B, H, W, C] == tokens-major [B, H*W, C]   (no transposes
               between conv and attention layers)
  conv 3x3     fp16 [Cout, 3, 3, Cin]      (K-major rows for the implicit GEMM)
  linear/1x1   fp16 [N, K];  q|k|v of self-attention and k|v of cross-attention concatenated
  bias/affine  fp32

Two execution paths share these modules:
  native  (default)  fused kernels; P2P / MasaCtrl control is a device-side plan consumed by the
          attention kernels (`control.ControlPlan`), no probability map is materialised;
  generic            a hook replaced `Attention.forward` (any Python controller): maps are
          materialised by `get_attention_scores` and handed to Python, as the reference does.
There is no CPU path: tensors must live on the GPU and `libief_hip.so` must be built.
"""
import math
import os
from typing import Dict, Optional

import torch
from torch import nn

from . import hip, planes
from .config import UNetConfig


# LayerNorm of the transformer blocks folded into the following linear (see include/ief_hip.h, IefGemmParams.rstat_*):
# removes the 48 LayerNorm launches of a UNet forward and one read + write of the residual stream each.  IEF_FOLD_LN=0
# keeps the separate kernels (A/B timing; the generic hook path always uses them).
FOLD_LN = os.environ.get("IEF_FOLD_LN", "1") == "1"
KV_ONE_GEMM = os.environ.get("IEF_KV_ONE_GEMM", "1") == "1"   # A/B switch of the one-GEMM per-step K/V projection


def _fold_ln(w16, bias32, gamma, beta):
    """(W gamma) fp16, bias + W beta, row sums of the fp16 folded weight (what the MFMA actually multiplies)"""
    wf = w16.float()
    wp = (wf * gamma[None, :]).half().contiguous()
    b = wf @ beta
    if bias32 is not None:
        b = b + bias32
    return wp, b.contiguous(), wp.float().sum(1).contiguous()


class UNetOutput(dict):
    """supports `out["sample"]` (sd_utils.py:73) and `out.sample` (ddim.py:29)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


# storage dtype of the weights being packed: fp16 (default path) or fp32 (reference-precision mode, `precision="f32"`:
# every hip.* wrapper then dispatches to the fp32-MFMA kernels of csrc/exact_f32.hip by the dtype of what it is handed).
# Set by UNet2DConditionModel.__init__ for the duration of the construction.
_PACK = torch.float16


def _f16(t, dev):
    """a weight / activation-side tensor in the model's storage dtype (fp16 unless the model is built for fp32)"""
    return t.detach().to(device=dev, dtype=_PACK).contiguous()


def _f32(t, dev):
    return t.detach().to(device=dev, dtype=torch.float32).contiguous()


# ------------------------------------------------------------------------------------- leaves
class Linear(nn.Module):
    def __init__(self, weight, bias=None):
        super().__init__()
        self.weight = weight  # fp16 [N, K] (may be a row-slice view of a fused weight)
        self.bias = bias      # fp32 [N] or None
        self.out_features, self.in_features = weight.shape

    def forward(self, x, residual=None):
        return hip.gemm(x, self.weight, bias=self.bias, residual=residual)


class Conv2d(nn.Module):
    """3x3 (pad 1) or 1x1 convolution over NHWC."""

    def __init__(self, weight, bias, kernel_size, stride=1):
        super().__init__()
        self.weight, self.bias = weight, bias
        self.kernel_size, self.stride = kernel_size, stride

    def forward(self, x, **kw):
        if self.kernel_size == 1:
            return hip.gemm(x, self.weight, bias=self.bias, residual=kw.get("residual"), col_stats=kw.get("col_stats", False))
        return hip.conv3x3(x, self.weight, self.bias, stride=self.stride, **kw)


class GroupNorm(nn.Module):
    def __init__(self, weight, bias, num_groups, eps):
        super().__init__()
        self.weight, self.bias, self.num_groups, self.eps = weight, bias, num_groups, eps

    def forward(self, x, silu=False, x2=None, return_stats=False, cstat=None, cstat2=None):
        return hip.groupnorm(x, self.weight, self.bias, self.num_groups, self.eps, silu=silu, x2=x2, return_stats=return_stats,
                             cstat=cstat, cstat2=cstat2)


class LayerNorm(nn.Module):
    def __init__(self, weight, bias, eps=1e-5):
        super().__init__()
        self.weight, self.bias, self.eps = weight, bias, eps

    def forward(self, x):
        return hip.layernorm(x, self.weight, self.bias, self.eps)


class SiLU(nn.Module):
    def forward(self, x):
        return hip.silu(x)


class Dropout(nn.Module):
    def forward(self, x):
        return x


# ------------------------------------------------------------------------------------- attention
class Attention(nn.Module):
    """Hook target.  Native forward = fused kernels; see module docstring for the generic path."""

    def __init__(self, sd, prefix, dim, heads, cross_dim, dev, name):
        super().__init__()
        self.heads = heads
        self.inner_dim = dim
        self.dim_head = dim // heads
        self.scale = self.dim_head ** -0.5
        self.is_cross = cross_dim is not None
        self.layer_name = name
        wq, wk, wv = (_f16(sd[f"{prefix}.to_{n}.weight"], dev) for n in "qkv")
        if self.is_cross:
            self.w_kv = torch.cat([wk, wv], 0).contiguous()
            self.to_q = Linear(wq)
            self.to_k, self.to_v = Linear(self.w_kv[:dim]), Linear(self.w_kv[dim:])
            self.w_qkv = None
        else:
            self.w_qkv = torch.cat([wq, wk, wv], 0).contiguous()
            self.to_q, self.to_k, self.to_v = (Linear(self.w_qkv[i * dim:(i + 1) * dim]) for i in range(3))
            self.w_kv = None
        self.to_out = nn.ModuleList([
            Linear(_f16(sd[f"{prefix}.to_out.0.weight"], dev), _f32(sd[f"{prefix}.to_out.0.bias"], dev)), Dropout()])
        # attributes the reference's patched forward reads (register.py:15-62)
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        self.processor = None
        self._plan = None          # control.ControlPlan when a lowered controller is registered
        self.map_out = None        # cross modules: fp16 [B*heads, N, 77] receiving this call's softmax maps (Pix2Pix-zero)
        self._kv_key, self._kv = None, None
        self.cache_kv = True       # False when the context changes every step (null-text embeddings)
        self._kv_view = None
        self.ln_w = self.ln_b = self.ln_c1 = None   # input projection with the preceding LayerNorm folded in
        self.ln_eps = 1e-5

    def fold_layernorm(self, norm):
        """pack W gamma / beta W^T of the q (cross) or q|k|v (self) projection for the LayerNorm that feeds this module"""
        w = self.w_qkv if self.w_qkv is not None else self.to_q.weight
        self.ln_w, self.ln_b, self.ln_c1 = _fold_ln(w, None, norm.weight, norm.bias)
        self.ln_eps = norm.eps

    # ---- protocol used by hook closures (generic path)
    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim=3):
        if attention_mask is not None:
            raise NotImplementedError("attention masks are not on the reference path (always None)")
        return None

    def head_to_batch_dim(self, t, out_dim=3):
        B, L, _ = t.shape
        t = t.reshape(B, L, self.heads, self.dim_head).permute(0, 2, 1, 3)
        return t.reshape(B * self.heads, L, self.dim_head) if out_dim == 3 else t

    def batch_to_head_dim(self, t):
        BH, L, d = t.shape
        B = BH // self.heads
        return t.reshape(B, self.heads, L, d).permute(0, 2, 1, 3).reshape(B, L, self.heads * d)

    def get_attention_scores(self, query, key, attention_mask=None):
        """[B*h, N, d], [B*h, L, d] -> materialised softmax maps [B*h, N, L] (fp16, contiguous)."""
        if attention_mask is not None:
            raise NotImplementedError("attention masks are not on the reference path")
        q = self.batch_to_head_dim(query).contiguous()
        k = self.batch_to_head_dim(key).contiguous()
        return hip.attn_probs(q, k, self.heads, self.scale)

    def apply_probs(self, probs, value):
        """probs [B*h,N,L] x value [B*h,L,d] -> [B,N,h*d] (the bmm + batch_to_head_dim of register.py:50-51)."""
        v = self.batch_to_head_dim(value).contiguous()
        return hip.attn_apply(probs.contiguous(), v, self.heads)

    def get_processor(self):
        return self.processor

    def set_processor(self, processor):
        self.processor = processor

    def is_native(self) -> bool:
        f = self.__dict__.get("forward")  # a hook assigns an instance attribute; the reference's unregister
        # restores the ORIGINAL bound method the same way (register.py:104), which is still native
        return (f is None or getattr(f, "__func__", None) is Attention.forward) and self.processor is None

    # ---- cross-attention K/V of a fixed context are step-invariant: project once per context
    def context_kv(self, ctx):
        if self._kv_view is not None:       # this forward's slice of the one-GEMM projection of all layers (UNet.forward)
            return self._kv_view
        if not self.cache_kv:
            return hip.gemm(ctx, self.w_kv)
        # identity (not address) of the context tensor: the cache holds a reference, so the allocator
        # cannot hand the same address to a different context while the entry is alive
        key = (ctx, ctx._version)
        if self._kv_key is None or self._kv_key[0] is not ctx or self._kv_key[1] != ctx._version:
            self._kv = hip.gemm(ctx, self.w_kv)
            self._kv_key = key
        return self._kv

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, residual=None,
                ln_stats=None, want_stats=False, **cross_attention_kwargs):
        """ln_stats: row moments of `hidden_states` (then the RAW residual stream: the LayerNorm is folded into the
        input projection); want_stats: also return the row moments of the output (for the next folded LayerNorm)."""
        if self.processor is not None:  # Pix2Pix-zero style processors own the dataflow
            return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                                  attention_mask=attention_mask, **cross_attention_kwargs)
        x = hidden_states
        B, N, C = x.shape
        self.last_tokens = N       # query count at this module's resolution level (sizes the Pix2Pix-zero map buffers)
        plan = self._plan
        ln = None if ln_stats is None else (ln_stats, self.ln_c1, self.ln_eps)
        if encoder_hidden_states is None:
            qkv = hip.gemm(x, self.w_qkv) if ln is None else hip.gemm(x, self.ln_w, bias=self.ln_b, ln=ln)
            q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
            qs = ks = vs = None
            if plan is not None:
                qs, ks, vs = plan.self_sources(B, N, self)
            o = hip.attn_flash(q, k, v, self.heads, self.scale, q_src=qs, k_src=ks, v_src=vs)
        else:
            q = hip.gemm(x, self.to_q.weight) if ln is None else hip.gemm(x, self.ln_w, bias=self.ln_b, ln=ln)
            kv = self.context_kv(encoder_hidden_states)
            args = plan.cross_edit(B, self) if plan is not None else {}
            if self.map_out is not None:     # `attn.attn_probs = attention_probs` (pix2pix-zero/model/attention_control.py:47)
                hip.attn_probs(q, kv[..., :C], self.heads, self.scale, out=self.map_out)
            o = hip.attn_cross_p2p(q, kv[..., :C], kv[..., C:], self.heads, self.scale, **args)
        if plan is not None:
            plan.layer_done(self)
        lin = self.to_out[0]
        if want_stats:
            return hip.gemm(o, lin.weight, bias=lin.bias, residual=residual, row_stats=True)
        return lin(o, residual=residual)


    def fold_layernorm_x3p(self, norm):
        """split-operand mode: W gamma / b + W beta / colsum of the q (cross) or q|k|v (self) projection (planes.FoldedLN)"""
        w = self.w_qkv if self.w_qkv is not None else self.to_q.weight
        self.ln3 = planes.FoldedLN(w, None, norm.weight, norm.bias, norm.eps)

    def forward_x3p(self, xp, ctx, residual, ln_stats=None, want_stats=False):
        """split-operand mode on operand planes (csrc/gemm_x3p.hip): xp = planes of the LayerNorm'ed input (written by the
        LayerNorm launch) -- or, with ln_stats (planes.RowStats of its rows), planes of the RAW residual stream: the LayerNorm is
        then folded into the input projection.  The attention kernel writes the planes to_out's GEMM stages by LDS-DMA; returns
        fp32 (the residual stream); want_stats: (fp32, Planes, RowStats) of the output for the next folded LayerNorm.  Same
        dataflow as `forward` (`/root/reference/p2p/model/register.py:11-64`)."""
        B, N, C = xp.shape
        self.last_tokens = N
        plan = self._plan
        w_in, b_in, ln = (self.w_qkv if ctx is None else self.to_q.weight), None, None
        if ln_stats is not None:
            w_in, b_in, ln = self.ln3.w, self.ln3.bias, (ln_stats, self.ln3.colsum, self.ln3.eps)
        if ctx is None:
            qs = ks = vs = None
            if plan is not None:
                qs, ks, vs = plan.self_sources(B, N, self)
            if planes.FLASH_PLANES and self.dim_head in planes.FLASH_PLANES_DIMS:
                # the q|k|v GEMM writes planes only; the attention kernel stages K / V tiles by LDS-DMA and splits nothing
                qkv = planes.gemm(xp, w_in, bias=b_in, ln=ln, out=False, out_planes=True)
                o = planes.attn_flash(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], self.heads, self.scale, q_src=qs,
                                      k_src=ks, v_src=vs)
            else:
                qkv = planes.gemm(xp, w_in, bias=b_in, ln=ln)
                o = hip.attn_flash(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], self.heads, self.scale, q_src=qs, k_src=ks,
                                   v_src=vs, out_planes=True)
        else:
            q = planes.gemm(xp, w_in, bias=b_in, ln=ln)
            kv = self.context_kv(ctx)
            args = plan.cross_edit(B, self) if plan is not None else {}
            if self.map_out is not None:
                hip.attn_probs(q, kv[..., :C], self.heads, self.scale, out=self.map_out)
            o = hip.attn_cross_p2p(q, kv[..., :C], kv[..., C:], self.heads, self.scale, out_planes=True, **args)
        if plan is not None:
            plan.layer_done(self)
        lin = self.to_out[0]
        if want_stats:
            return planes.gemm(o, lin.weight, bias=lin.bias, residual=residual, out=True, out_planes=True, row_stats=True)
        return planes.gemm(o, lin.weight, bias=lin.bias, residual=residual)


class GEGLU(nn.Module):
    """Linear(C -> 8C) + hidden * gelu(gate) in ONE GEMM: the [hidden | gate] halves of the weight are interleaved
    in groups of 8 rows at pack time so the GEMM epilogue sees a hidden chunk next to its gate chunk and writes
    the 4C-wide product directly (the 8C-wide intermediate never reaches HBM)."""

    def __init__(self, sd, prefix, dev):
        super().__init__()
        w, b = sd[prefix + ".proj.weight"], sd[prefix + ".proj.bias"]
        half = w.shape[0] // 2
        il = lambda t: torch.stack([t[:half].reshape(half // 8, 8, *t.shape[1:]),
                                    t[half:].reshape(half // 8, 8, *t.shape[1:])], 1).reshape(t.shape)
        self.proj = Linear(_f16(il(w), dev), _f32(il(b), dev))   # NOTE: rows interleaved [8 hidden | 8 gate] ...
        self.proj.interleaved = True

    def fold_layernorm(self, norm):
        self.ln_w, self.ln_b, self.ln_c1 = _fold_ln(self.proj.weight, self.proj.bias, norm.weight, norm.bias)
        self.ln_eps = norm.eps

    def forward(self, x, ln_stats=None):
        if ln_stats is not None:
            return hip.gemm(x, self.ln_w, bias=self.ln_b, geglu=True, ln=(ln_stats, self.ln_c1, self.ln_eps))
        return hip.gemm(x, self.proj.weight, bias=self.proj.bias, geglu=True)


class FeedForward(nn.Module):
    def __init__(self, sd, prefix, dev):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(sd, prefix + ".net.0", dev), Dropout(),
                                  Linear(_f16(sd[prefix + ".net.2.weight"], dev), _f32(sd[prefix + ".net.2.bias"], dev))])

    def forward(self, x, residual=None):
        return self.net[2](self.net[0](x), residual=residual)


class BasicTransformerBlock(nn.Module):
    def __init__(self, sd, prefix, dim, heads, cross_dim, dev, name):
        super().__init__()
        ln = lambda n: LayerNorm(_f32(sd[f"{prefix}.{n}.weight"], dev), _f32(sd[f"{prefix}.{n}.bias"], dev))
        self.norm1, self.norm2, self.norm3 = ln("norm1"), ln("norm2"), ln("norm3")
        self.attn1 = Attention(sd, prefix + ".attn1", dim, heads, None, dev, name + ".attn1")
        self.attn2 = Attention(sd, prefix + ".attn2", dim, heads, cross_dim, dev, name + ".attn2")
        self.ff = FeedForward(sd, prefix + ".ff", dev)
        if dev.type == "cuda" and _PACK == torch.float16:      # LayerNorm folding lives in the fp16 GEMM's epilogue
            self.attn1.fold_layernorm(self.norm1)
            self.attn2.fold_layernorm(self.norm2)
            self.ff.net[0].fold_layernorm(self.norm3)

    def foldable(self):
        return FOLD_LN and self.attn1.ln_w is not None and self.attn1.is_native() and self.attn2.is_native()

    @staticmethod
    def _attend(attn, x, res, ctx):
        if attn.is_native():
            return attn(x, encoder_hidden_states=ctx, residual=res)
        # a hook owns forward: reference signature, residual added here (one extra fp16 rounding)
        out = attn(x, encoder_hidden_states=ctx)
        return hip.add(out.contiguous(), res)

    def forward(self, h, ctx, ln_stats=None, want_stats=False):
        """ln_stats: row moments of h from the GEMM that produced it => the three LayerNorms run folded (no launch);
        want_stats: also return the row moments of the output (the next block of a deeper Transformer2DModel folds on them)"""
        if ln_stats is not None:
            h, st = self.attn1(h, residual=h, ln_stats=ln_stats, want_stats=True)
            h, st = self.attn2(h, encoder_hidden_states=ctx, residual=h, ln_stats=st, want_stats=True)
            g = self.ff.net[0](h, ln_stats=st)
            lin = self.ff.net[2]
            if want_stats:
                return hip.gemm(g, lin.weight, bias=lin.bias, residual=h, row_stats=True)
            return lin(g, residual=h)
        h = self._attend(self.attn1, self.norm1(h), h, None)
        h = self._attend(self.attn2, self.norm2(h), h, ctx)
        return self.ff(self.norm3(h), residual=h)


    def _attend_x3p(self, attn, norm, h, ctx):
        if attn.is_native():
            return attn.forward_x3p(planes.layernorm(h, norm.weight, norm.bias, norm.eps), ctx, residual=h)
        return self._attend(attn, norm(h), h, ctx)          # a hook owns forward: fp32 in, fp32 out (generic path)

    def fold_layernorm_x3p(self):
        self.attn1.fold_layernorm_x3p(self.norm1)
        self.attn2.fold_layernorm_x3p(self.norm2)
        f0 = self.ff.net[0].proj
        self.ff.net[0].ln3 = planes.FoldedLN(f0.weight, f0.bias, self.norm3.weight, self.norm3.bias, self.norm3.eps)

    def foldable_x3p(self):
        """folded weights are DERIVED tensors: made at the first eager forward (after a multi-GPU weight broadcast, like the weight
        planes), never while a graph is being captured"""
        if not (planes.LN_FOLD and self.attn1.is_native() and self.attn2.is_native()):
            return False
        if getattr(self.attn1, "ln3", None) is None:
            if hip._capturing():
                return False
            self.fold_layernorm_x3p()
        return True

    def forward_x3p_folded(self, h, hp, st, ctx, last, want_stats):
        """the three LayerNorms folded into the GEMMs they feed (no LayerNorm launch): h fp32 + hp its planes + st the row
        statistics its producer left; every producer of the residual stream writes fp32, planes and statistics"""
        h, hp, st = self.attn1.forward_x3p(hp, None, residual=h, ln_stats=st, want_stats=True)
        h, hp, st = self.attn2.forward_x3p(hp, ctx, residual=h, ln_stats=st, want_stats=True)
        f0, lin = self.ff.net[0], self.ff.net[2]
        g = planes.gemm(hp, f0.ln3.w, bias=f0.ln3.bias, geglu=True, out=False, out_planes=True, ln=(st, f0.ln3.colsum, f0.ln3.eps))
        if last:
            return planes.gemm(g, lin.weight, bias=lin.bias, residual=h, out=False, out_planes=True)
        if want_stats:
            return planes.gemm(g, lin.weight, bias=lin.bias, residual=h, out=True, out_planes=True, row_stats=True)
        return planes.gemm(g, lin.weight, bias=lin.bias, residual=h)

    def forward_x3p(self, h, ctx, last):
        """split-operand mode on operand planes: every LayerNorm writes the planes its GEMM consumes, FeedForward.net[0]
        writes the GEGLU product as planes; last: the block's output is read by proj_out only -> planes only"""
        h = self._attend_x3p(self.attn1, self.norm1, h, None)
        h = self._attend_x3p(self.attn2, self.norm2, h, ctx)
        n3 = planes.layernorm(h, self.norm3.weight, self.norm3.bias, self.norm3.eps)
        f0, lin = self.ff.net[0].proj, self.ff.net[2]
        g = planes.gemm(n3, f0.weight, bias=f0.bias, geglu=True, out=False, out_planes=True)
        if last:
            return planes.gemm(g, lin.weight, bias=lin.bias, residual=h, out=False, out_planes=True)
        return planes.gemm(g, lin.weight, bias=lin.bias, residual=h)


class Transformer2DModel(nn.Module):
    def __init__(self, sd, prefix, dim, heads, cross_dim, groups, dev, name, depth=1):
        super().__init__()
        self.norm = GroupNorm(_f32(sd[prefix + ".norm.weight"], dev), _f32(sd[prefix + ".norm.bias"], dev), groups, 1e-6)
        c1 = lambda n: Conv2d(_f16(sd[f"{prefix}.{n}.weight"].reshape(dim, dim), dev), _f32(sd[f"{prefix}.{n}.bias"], dev), 1)
        self.proj_in, self.proj_out = c1("proj_in"), c1("proj_out")
        # SDXL stacks several blocks per Transformer2DModel (config.transformer_layers); block 0 keeps the SD1.x layer name
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(sd, f"{prefix}.transformer_blocks.{k}", dim, heads, cross_dim, dev,
                                   name if k == 0 else f"{name}.b{k}") for k in range(depth)])

    def forward(self, x, encoder_hidden_states=None, cstat=None, col_stats=False):
        """cstat: the producer's column statistics of x (hip.ColStats) for the GroupNorm; col_stats: also return those of
        the output (proj_out's launch) for the next block's GroupNorm"""
        B, H, W, C = x.shape
        if all(blk.foldable() for blk in self.transformer_blocks):
            h, st = hip.gemm(self.norm(x, cstat=cstat), self.proj_in.weight, bias=self.proj_in.bias, row_stats=True)
            h = h.reshape(B, H * W, C)
            last = len(self.transformer_blocks) - 1
            for k, blk in enumerate(self.transformer_blocks):
                if k < last:
                    h, st = blk(h, encoder_hidden_states, ln_stats=st, want_stats=True)
                else:
                    h = blk(h, encoder_hidden_states, ln_stats=st)
            return self.proj_out(h.reshape(B, H, W, C), residual=x, col_stats=col_stats)
        h = self.proj_in(self.norm(x, cstat=cstat)).reshape(B, H * W, C)
        for blk in self.transformer_blocks:
            h = blk(h, encoder_hidden_states)
        return self.proj_out(h.reshape(B, H, W, C), residual=x, col_stats=col_stats)


    def forward_x3p(self, x, encoder_hidden_states=None, want_planes=False):
        """x fp32 NHWC -> (out fp32, Planes | None): GroupNorm writes proj_in's operand planes, the last block's FeedForward
        writes proj_out's; proj_out's epilogue adds the fp32 residual and (want_planes) also emits the planes of the result for
        a consumer that reads the raw stream (a shortcut source, Downsample2D / Upsample2D, a skip connection)"""
        B, H, W, C = x.shape
        n = self.norm
        hn = planes.groupnorm(x, n.weight, n.bias, n.num_groups, n.eps, silu=False)
        last = len(self.transformer_blocks) - 1
        if all(blk.foldable_x3p() for blk in self.transformer_blocks):
            h, hp, st = planes.gemm(hn.reshape(B, H * W, C), self.proj_in.weight, bias=self.proj_in.bias, out=True, out_planes=True,
                                    row_stats=True)
            for k, blk in enumerate(self.transformer_blocks):
                r = blk.forward_x3p_folded(h, hp, st, encoder_hidden_states, last=(k == last), want_stats=(k < last))
                if k < last:
                    h, hp, st = r
                else:
                    h = r
        else:
            h = planes.gemm(hn.reshape(B, H * W, C), self.proj_in.weight, bias=self.proj_in.bias)
            for k, blk in enumerate(self.transformer_blocks):
                h = blk.forward_x3p(h, encoder_hidden_states, last=(k == last))
        r = planes.gemm(h, self.proj_out.weight, bias=self.proj_out.bias, residual=x.reshape(B, H * W, C), out=True,
                        out_planes=want_planes)
        if want_planes:
            return r[0].reshape(B, H, W, C), r[1].reshape(B, H, W, C)
        return r.reshape(B, H, W, C), None


# ------------------------------------------------------------------------------------- resnet
class ResnetBlock2D(nn.Module):
    """norm1+SiLU -> conv1 (+bias +time-emb) -> norm2+SiLU -> conv2 (+bias +skip); the 1x1 shortcut of a
    channel-changing block is folded into conv2's implicit GEMM as an extra K range over the raw input(s).
    Dataflow: `/root/reference/pnp/model/register.py:102-175`."""

    def __init__(self, sd, prefix, cin, cout, groups, eps, dev, temb_slot):
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        self.norm1 = GroupNorm(_f32(sd[prefix + ".norm1.weight"], dev), _f32(sd[prefix + ".norm1.bias"], dev), groups, eps)
        self.norm2 = GroupNorm(_f32(sd[prefix + ".norm2.weight"], dev), _f32(sd[prefix + ".norm2.bias"], dev), groups, eps)
        pack = lambda w: _f16(w.permute(0, 2, 3, 1), dev)  # OIHW -> [O][kh][kw][I]
        self.conv1 = Conv2d(pack(sd[prefix + ".conv1.weight"]), _f32(sd[prefix + ".conv1.bias"], dev), 3)
        self.time_emb_proj = Linear(_f16(sd[prefix + ".time_emb_proj.weight"], dev), _f32(sd[prefix + ".time_emb_proj.bias"], dev))
        self.nonlinearity = SiLU()
        self.dropout = Dropout()
        self.upsample = self.downsample = None
        self.skip_time_act = False
        self.time_embedding_norm = "default"
        self.output_scale_factor = 1.0
        w2 = sd[prefix + ".conv2.weight"].permute(0, 2, 3, 1).reshape(cout, 9 * cout)
        b2 = sd[prefix + ".conv2.bias"]
        if cin != cout:
            ws = sd[prefix + ".conv_shortcut.weight"].reshape(cout, cin)
            self.conv_shortcut = Conv2d(_f16(ws, dev), _f32(sd[prefix + ".conv_shortcut.bias"], dev), 1)
            self.w2_fused = _f16(torch.cat([w2, ws], 1), dev)                       # [Cout][9*Cout + Cin]
            self.b2_fused = _f32(b2 + sd[prefix + ".conv_shortcut.bias"], dev)
        else:
            self.conv_shortcut = None
            self.w2_fused, self.b2_fused = None, None
        self.conv2 = Conv2d(_f16(w2.reshape(cout, 3, 3, cout), dev), _f32(b2, dev), 3)
        self.temb_slot = temb_slot  # (offset, width) into the UNet's per-step time-embedding row
        self._inject = None         # control.ControlPlan of a registered Plug-and-Play feature injection

    def forward(self, x, temb_row, skip=None, cstat=None, cstat_skip=None, col_stats=False):
        """x [B,H,W,C1] (+ skip [B,H,W,C2] = un-materialised channel concat); temb_row fp32 [B or 1, Cout].
        cstat / cstat_skip: column statistics the producers of x / skip left (hip.ColStats or None);
        col_stats: return (out, ColStats | None) so the next GroupNorm can skip its statistics pass."""
        h = self.norm1(x, silu=True, x2=skip, cstat=cstat, cstat2=cstat_skip)
        h, hs = hip.conv3x3(h, self.conv1.weight, self.conv1.bias, rowvec=temb_row, col_stats=True)
        h = self.norm2(h, silu=True, cstat=hs)
        if self._inject is not None:
            # Plug-and-Play replaces conv2's OUTPUT rows by the source image's (pnp/model/register.py:161-166); conv2 acts
            # per batch row, so gathering its input rows is the same thing and keeps the fused shortcut / residual add
            src = self._inject.feature_source(h.shape[0])
            if src is not None:
                h = hip.gather_rows(h, src)
        if self.conv_shortcut is None:
            return hip.conv3x3(h, self.conv2.weight, self.conv2.bias, residual=x, col_stats=col_stats)
        return hip.conv3x3_shortcut(h, self.w2_fused, self.b2_fused, x, skip, col_stats=col_stats)


    def forward_x3p(self, x, xp, temb_row, skip=None, skip_p=None, want_planes=False):
        """split-operand mode on operand planes: x (+ skip) fp32 NHWC; xp / skip_p their planes (needed by the fused 1x1
        shortcut only; split here when the producer did not emit them).  The two GroupNorms write the planes conv1 / conv2
        stage by LDS-DMA; returns (out fp32, Planes | None).  Dataflow: `/root/reference/pnp/model/register.py:102-175`."""
        n1, n2 = self.norm1, self.norm2
        g1 = planes.groupnorm(x, n1.weight, n1.bias, n1.num_groups, n1.eps, silu=True, x2=skip)
        h = planes.conv3x3(g1, self.conv1.weight, self.conv1.bias, rowvec=temb_row)
        if self._inject is not None and self._inject.feature_source(h.shape[0]) is not None:
            # Plug-and-Play: gather conv2's input rows (see `forward`); the gather runs on fp32, then one split
            g2 = planes.split(hip.gather_rows(hip.groupnorm(h, n2.weight, n2.bias, n2.num_groups, n2.eps, silu=True),
                                              self._inject.feature_source(h.shape[0])))
        else:
            g2 = planes.groupnorm(h, n2.weight, n2.bias, n2.num_groups, n2.eps, silu=True)
        if self.conv_shortcut is None:
            r = planes.conv3x3(g2, self.conv2.weight, self.conv2.bias, residual=x, out=True, out_planes=want_planes)
        else:
            xp = planes.split(x) if xp is None else xp
            if skip is not None and skip_p is None:
                skip_p = planes.split(skip)
            r = planes.conv3x3(g2, self.w2_fused, self.b2_fused, extra=(xp, skip_p), out=True, out_planes=want_planes)
        return r if want_planes else (r, None)


class Downsample2D(nn.Module):
    def __init__(self, sd, prefix, dev):
        super().__init__()
        self.conv = Conv2d(_f16(sd[prefix + ".conv.weight"].permute(0, 2, 3, 1), dev), _f32(sd[prefix + ".conv.bias"], dev), 3, stride=2)

    def forward(self, x, col_stats=False):
        return self.conv(x, col_stats=col_stats)

    def forward_x3p(self, xp, want_planes=True):
        r = planes.conv3x3(xp, self.conv.weight, self.conv.bias, stride=2, out=True, out_planes=want_planes)
        return r if want_planes else (r, None)


class Upsample2D(nn.Module):
    """nearest 2x + conv3x3, the interpolation fused into the conv's gather (never materialised)."""

    def __init__(self, sd, prefix, dev):
        super().__init__()
        self.conv = Conv2d(_f16(sd[prefix + ".conv.weight"].permute(0, 2, 3, 1), dev), _f32(sd[prefix + ".conv.bias"], dev), 3)

    def forward(self, x, col_stats=False):
        return hip.conv3x3(x, self.conv.weight, self.conv.bias, upsample=True, col_stats=col_stats)

    def forward_x3p(self, xp, want_planes=True):
        r = planes.conv3x3(xp, self.conv.weight, self.conv.bias, upsample=True, out=True, out_planes=want_planes)
        return r if want_planes else (r, None)


# ------------------------------------------------------------------------------------- blocks
class _Block(nn.Module):
    def __init__(self):
        super().__init__()
        self.resnets = nn.ModuleList()
        self.attentions = nn.ModuleList()
        self.has_cross_attention = False


class CrossAttnDownBlock2D(_Block):
    pass


class DownBlock2D(_Block):
    pass


class CrossAttnUpBlock2D(_Block):
    pass


class UpBlock2D(_Block):
    pass


class UNetMidBlock2DCrossAttn(_Block):
    pass


class Timesteps(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.num_channels = dim
        self.dtype = _PACK

    def forward(self, t):
        return hip.timestep_embedding(t, self.num_channels, dtype=self.dtype)


class TimestepEmbedding(nn.Module):
    def __init__(self, sd, dev, prefix="time_embedding"):
        super().__init__()
        self.linear_1 = Linear(_f16(sd[prefix + ".linear_1.weight"], dev), _f32(sd[prefix + ".linear_1.bias"], dev))
        self.act = SiLU()
        self.linear_2 = Linear(_f16(sd[prefix + ".linear_2.weight"], dev), _f32(sd[prefix + ".linear_2.bias"], dev))

    def forward(self, x):
        return self.linear_2(self.act(self.linear_1(x)))


class UNet2DConditionModel(nn.Module):
    def __init__(self, cfg: UNetConfig, state_dict: Dict[str, torch.Tensor], device="cuda:0", precision: str = "f16"):
        """precision: "f16" — fp16 storage, fp32 accumulation (the fast path); "f32" — the reference's precision
        (`/root/reference/p2p/edit_syn.py:38`): fp32 weights and activations on the fp32-input MFMA kernels; "f16x3" — fp32
        weights and activations, every contraction on split fp16 operands (hi + lo halves, three fp16 MFMAs per product,
        `csrc/split_x3.hip`): ~21 operand bits at a third of the fp16 matrix rate"""
        super().__init__()
        global _PACK
        if precision not in ("f16", "f32", "f16x3"):
            raise ValueError('precision must be "f16", "f32" or "f16x3"')
        self.precision = precision
        self.contract = "x3" if precision == "f16x3" else "f32"       # hip.f32_contraction mode of this model's calls
        saved, _PACK = _PACK, (torch.float16 if precision == "f16" else torch.float32)
        try:
            self._build(cfg, state_dict, device)
        finally:
            _PACK = saved

    def _build(self, cfg, state_dict, device):
        hip.load()  # fail loudly before touching anything else
        dev = torch.device(device)
        # a CPU device is accepted for CONSTRUCTION only (module-tree / packing tests); every forward
        # launches HIP kernels and rejects host tensors — there is no CPU compute path
        sd = state_dict
        self.cfg = cfg
        self.config = _Config(cfg)
        self._device = dev
        self.dtype = _PACK
        ch = cfg.block_out_channels
        nlev = len(ch)
        G, eps = cfg.norm_num_groups, cfg.norm_eps
        self.conv_in = Conv2d(_f16(sd["conv_in.weight"].permute(2, 3, 1, 0), dev), _f32(sd["conv_in.bias"], dev), 3)  # [3,3,Cin,Cout]
        self.time_proj = Timesteps(ch[0])
        self.time_embedding = TimestepEmbedding(sd, dev)
        if cfg.addition_embed:      # SDXL `addition_embed_type = "text_time"`
            self.add_time_proj = Timesteps(cfg.addition_time_embed_dim)
            self.add_embedding = TimestepEmbedding(sd, dev, "add_embedding")
        self._temb_width = 0
        self._resnets = []

        def resnet(prefix, cin, cout):
            r = ResnetBlock2D(sd, prefix, cin, cout, G, eps, dev, (self._temb_width, cout))
            self._temb_width += cout
            self._resnets.append(r)
            return r

        self.down_blocks = nn.ModuleList()
        cout = ch[0]
        for i in range(nlev):
            cin, cout = cout, ch[i]
            blk = CrossAttnDownBlock2D() if cfg.down_has_attn[i] else DownBlock2D()
            blk.has_cross_attention = cfg.down_has_attn[i]
            for j in range(cfg.layers_per_block):
                blk.resnets.append(resnet(f"down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout))
                if cfg.down_has_attn[i]:
                    blk.attentions.append(Transformer2DModel(sd, f"down_blocks.{i}.attentions.{j}", cout, cfg.num_heads[i],
                                                             cfg.cross_attention_dim, G, dev, f"down{i}.{j}", cfg.depth(i)))
            blk.downsamplers = nn.ModuleList([Downsample2D(sd, f"down_blocks.{i}.downsamplers.0", dev)]) if i < nlev - 1 else None
            self.down_blocks.append(blk)
        self.up_blocks = nn.ModuleList()
        rev, rev_attn, rev_heads = tuple(reversed(ch)), tuple(reversed(cfg.down_has_attn)), tuple(reversed(cfg.num_heads))
        out_c = rev[0]
        up_specs = []
        for i in range(nlev):
            prev, out_c = out_c, rev[i]
            in_c = rev[min(i + 1, nlev - 1)]
            up_specs.append((i, prev, out_c, in_c))
        cm = ch[-1]
        self.mid_block = UNetMidBlock2DCrossAttn()
        self.mid_block.has_cross_attention = True
        self.mid_block.resnets.append(resnet("mid_block.resnets.0", cm, cm))
        self.mid_block.attentions.append(Transformer2DModel(sd, "mid_block.attentions.0", cm, cfg.num_heads[-1],
                                                            cfg.cross_attention_dim, G, dev, "mid", cfg.depth(nlev - 1)))
        self.mid_block.resnets.append(resnet("mid_block.resnets.1", cm, cm))
        for i, prev, out_c, in_c in up_specs:
            blk = CrossAttnUpBlock2D() if rev_attn[i] else UpBlock2D()
            blk.has_cross_attention = rev_attn[i]
            blk.skip_channels = []
            for j in range(cfg.layers_per_block + 1):
                skip = in_c if j == cfg.layers_per_block else out_c
                rin = prev if j == 0 else out_c
                blk.skip_channels.append(skip)
                blk.resnets.append(resnet(f"up_blocks.{i}.resnets.{j}", rin + skip, out_c))
                if rev_attn[i]:
                    blk.attentions.append(Transformer2DModel(sd, f"up_blocks.{i}.attentions.{j}", out_c, rev_heads[i],
                                                             cfg.cross_attention_dim, G, dev, f"up{i}.{j}",
                                                             cfg.depth(nlev - 1 - i)))
            blk.upsamplers = nn.ModuleList([Upsample2D(sd, f"up_blocks.{i}.upsamplers.0", dev)]) if i < nlev - 1 else None
            self.up_blocks.append(blk)
        self.conv_norm_out = GroupNorm(_f32(sd["conv_norm_out.weight"], dev), _f32(sd["conv_norm_out.bias"], dev), G, eps)
        self.conv_act = SiLU()
        self.conv_out = Conv2d(_f16(sd["conv_out.weight"].permute(0, 2, 3, 1), dev), _f32(sd["conv_out.bias"], dev), 3)
        # all 22 time_emb_proj linears as ONE [sum Cout, temb] GEMM
        self._temb_w = torch.cat([r.time_emb_proj.weight for r in self._resnets], 0).contiguous()
        self._temb_b = torch.cat([r.time_emb_proj.bias for r in self._resnets], 0).contiguous()
        # execution-order index of every Attention module (down, mid, up; attn1 then attn2 per transformer):
        # controllers that count calls (`cur_att_layer`) see the layers in THIS order
        order = [m for blk in self.down_blocks for m in blk.modules() if m.__class__.__name__ == "Attention"]
        order += [m for m in self.mid_block.modules() if m.__class__.__name__ == "Attention"]
        order += [m for blk in self.up_blocks for m in blk.modules() if m.__class__.__name__ == "Attention"]
        for i, m in enumerate(order):
            m._exec_index = i
        # split-operand mode: activations travel as pre-split operand planes between the kernels (planes.py, csrc/gemm_x3p.hip)
        # when every channel count suits the planes kernels (K tiles of 32 channels, column tiles of 160 / 80 or 64)
        self.x3p = (self.precision == "f16x3" and planes.ENABLED and dev.type == "cuda"
                    and all(c % 160 == 0 or c % 64 == 0 for c in ch) and cfg.cross_attention_dim % 32 == 0)
        self._temb_table = None   # per-step rows, see precompute_time_table
        self._temb_static = None  # [1, width] fp32 buffer a captured graph reads
        self._plan = None

    # ------------------------------------------------------------------ small API
    @property
    def device(self):
        return self._device

    def attention_modules(self):
        """Attention modules in the order the reference's registration walk counts them."""
        out = []
        for name, child in self.named_children():
            if "down" in name or "up" in name or "mid" in name:
                out += [m for m in child.modules() if m.__class__.__name__ == "Attention"]
        return out

    def packed_tensors(self):
        """every owning device tensor of the packed model (views excluded) — what a weight broadcast sends"""
        seen, out = set(), []
        holders = list(self.modules()) + [self]
        for m in holders:
            for v in vars(m).values():
                if isinstance(v, torch.Tensor) and v.is_cuda and v._base is None and v.data_ptr() not in seen:
                    seen.add(v.data_ptr())
                    out.append(v)
        return out

    def aug_embedding(self, added_cond_kwargs):
        with hip.f32_contraction(self.contract):
            return self._aug_embedding(added_cond_kwargs)

    def time_rows(self, timesteps_f32, aug=None):
        with hip.f32_contraction(self.contract):
            return self._time_rows(timesteps_f32, aug)

    def forward(self, sample, timestep=None, encoder_hidden_states=None, cross_attention_kwargs=None,
                added_cond_kwargs=None, return_dict=True, temb_row=None, taps=None, **kw):
        with hip.f32_contraction(self.contract):
            return self._forward(sample, timestep, encoder_hidden_states, cross_attention_kwargs, added_cond_kwargs,
                                 return_dict, temb_row, taps, **kw)

    def _aug_embedding(self, added_cond_kwargs):
        """SDXL: fp16 [B, time_embed_dim] = add_embedding(cat([text_embeds, sinusoids of the 6 time ids])) — constant over
        the steps of one edit (`/root/reference/pix2pix-zero/model/sd_utils.py:408-421` builds the kwargs)"""
        if not self.cfg.addition_embed:
            return None
        if added_cond_kwargs is None:
            raise ValueError("this UNet has an additional text-time embedding: added_cond_kwargs is required")
        te = added_cond_kwargs["text_embeds"].to(self._device).float().contiguous()
        ids = added_cond_kwargs["time_ids"].to(self._device).float().contiguous()
        B = te.shape[0]
        tim = self.add_time_proj(ids.reshape(-1).contiguous()).reshape(B, -1)
        return self.add_embedding(torch.cat([self._act(te), tim], dim=-1).contiguous())

    def _time_rows(self, timesteps_f32, aug=None):
        """fp32 [T, sum Cout]: time_emb_proj(silu(time_embedding(time_proj(t)))) for every resnet at once.
        aug (fp16 [B, time_embed_dim], `aug_embedding`): rows become [T * B, sum Cout], step-major."""
        emb = self.time_embedding(self.time_proj(timesteps_f32))
        if aug is not None:
            T, B = emb.shape[0], aug.shape[0]
            emb = hip.add(emb[:, None, :].expand(T, B, -1).contiguous(), aug[None].expand(T, B, -1).contiguous())
            emb = emb.reshape(T * B, -1)
        rows = hip.gemm(hip.silu(emb), self._temb_w, bias=self._temb_b)
        return rows if rows.dtype == torch.float32 else hip.to_f32(rows)

    def precompute_time_table(self, timesteps):
        """Rows for a whole schedule (one per step); the fused loop then selects a row per step on device."""
        t = torch.as_tensor(timesteps).to(device=self._device, dtype=torch.float32).reshape(-1).contiguous()
        self._temb_table = self.time_rows(t).contiguous()
        self._temb_static = torch.empty(1, self._temb_width, dtype=torch.float32, device=self._device)
        return self._temb_table

    # ------------------------------------------------------------------ forward
    def _forward(self, sample, timestep=None, encoder_hidden_states=None, cross_attention_kwargs=None,
                 added_cond_kwargs=None, return_dict=True, temb_row=None, taps=None, **kw):
        """sample fp32/fp16 NCHW [B,4,H,W]; timestep scalar / 0-d tensor; ctx [B,77,Cc] -> eps fp32 NCHW.

        `temb_row` (fp32 [1, width]) short-circuits the time embedding for the captured-graph loop."""
        if not sample.is_cuda:
            raise RuntimeError("UNet input must be a device tensor (no CPU path)")
        x = sample if sample.dtype == torch.float32 else sample.float()
        x = x.contiguous()
        B = x.shape[0]
        ctx = encoder_hidden_states
        if ctx.dtype != self.dtype:
            ctx = self._ctx_f16(ctx)
        if temb_row is None:
            t = torch.as_tensor(timestep).to(device=self._device, dtype=torch.float32).reshape(-1)[:1].contiguous()
            temb_row = self.time_rows(t, self.aug_embedding(added_cond_kwargs))     # [1 or B, width]
        trow = lambda r: temb_row[:, r.temb_slot[0]:r.temb_slot[0] + r.temb_slot[1]].contiguous()
        if self._plan is not None:
            self._plan.begin_forward(B)
        # a context that changes every step (null-text embeddings: Attention.cache_kv is off) is projected for ALL 16
        # cross-attention layers by ONE GEMM; each layer reads its column slice
        cross = self._cross_modules()
        per_step_kv = KV_ONE_GEMM and bool(cross) and not cross[0].cache_kv and all(m.is_native() for m in cross)
        if per_step_kv:
            kv_all = hip.gemm(ctx, self._kv_all_weight())
            off = 0
            for m in cross:
                m._kv_view = kv_all[..., off:off + 2 * m.inner_dim]
                off += 2 * m.inner_dim

        def tap(name, v):  # debugging / parity aid: same tap points as oracle/unet_ref.py
            if taps is not None:
                taps[name] = v.float().permute(0, 3, 1, 2).cpu()

        if self.x3p:
            eps = self._trunk_x3p(x, trow, ctx, tap)
            if per_step_kv:
                for m in cross:
                    m._kv_view = None
            if self._plan is not None:
                self._plan.end_forward(B)
            return (eps,) if not return_dict else UNetOutput(sample=eps)
        # `hs` travels beside `h`: the column statistics its producing launch left for the next GroupNorm (hip.ColStats),
        # or None when that launch does not emit them (small levels, split-K plans, conv_in)
        h, hs = hip.conv_in(x, self.conv_in.weight, self.conv_in.bias), None
        tap("conv_in", h)
        skips = [(h, hs)]
        for bi, blk in enumerate(self.down_blocks):
            for j, res in enumerate(blk.resnets):
                h, hs = res(h, trow(res), cstat=hs, col_stats=True)
                if blk.has_cross_attention:
                    h, hs = blk.attentions[j](h, ctx, cstat=hs, col_stats=True)
                skips.append((h, hs))
            if blk.downsamplers is not None:
                h, hs = blk.downsamplers[0](h, col_stats=True)
                skips.append((h, hs))
            tap(f"down{bi}", h)
        h, hs = self.mid_block.resnets[0](h, trow(self.mid_block.resnets[0]), cstat=hs, col_stats=True)
        h, hs = self.mid_block.attentions[0](h, ctx, cstat=hs, col_stats=True)
        h, hs = self.mid_block.resnets[1](h, trow(self.mid_block.resnets[1]), cstat=hs, col_stats=True)
        tap("mid", h)
        for bi, blk in enumerate(self.up_blocks):
            for j, res in enumerate(blk.resnets):
                sk, sks = skips.pop()
                h, hs = res(h, trow(res), skip=sk, cstat=hs, cstat_skip=sks, col_stats=True)
                if blk.has_cross_attention:
                    h, hs = blk.attentions[j](h, ctx, cstat=hs, col_stats=True)
            if blk.upsamplers is not None:
                h, hs = blk.upsamplers[0](h, col_stats=True)
            tap(f"up{bi}", h)
        h = self.conv_norm_out(h, silu=True, cstat=hs)
        eps = hip.conv_out(h, self.conv_out.weight, self.conv_out.bias)
        if per_step_kv:
            for m in cross:
                m._kv_view = None
        if self._plan is not None:
            self._plan.end_forward(B)
        if not return_dict:
            return (eps,)
        return UNetOutput(sample=eps)

    def _trunk_x3p(self, x, trow, ctx, tap):
        """conv_in .. conv_out of the split-operand mode on operand planes.  `hp` travels beside `h`: the planes of h, emitted
        by the launch that produced h WHERE A CONSUMER READS THE RAW STREAM through a GEMM / convolution — a skip connection or
        block input feeding a fused 1x1 shortcut, Downsample2D / Upsample2D (no normalisation in front of them) — else None.
        Everything a normalisation feeds gets its planes from that normalisation's launch."""
        h = hip.conv_in(x, self.conv_in.weight, self.conv_in.bias)
        hp = planes.split(h)
        tap("conv_in", h)
        skips = [(h, hp)]
        for bi, blk in enumerate(self.down_blocks):
            for j, res in enumerate(blk.resnets):
                attn = blk.has_cross_attention
                h, hp = res.forward_x3p(h, hp, trow(res), want_planes=not attn)       # every down-path output is a skip connection
                if attn:
                    h, hp = blk.attentions[j].forward_x3p(h, ctx, want_planes=True)
                skips.append((h, hp))
            if blk.downsamplers is not None:
                h, hp = blk.downsamplers[0].forward_x3p(hp, want_planes=True)
                skips.append((h, hp))
            tap(f"down{bi}", h)
        mid = self.mid_block
        h, hp = mid.resnets[0].forward_x3p(h, hp, trow(mid.resnets[0]))
        h, hp = mid.attentions[0].forward_x3p(h, ctx)
        h, hp = mid.resnets[1].forward_x3p(h, None, trow(mid.resnets[1]), want_planes=True)     # x of up_blocks[0].resnets[0]'s shortcut
        tap("mid", h)
        nb = len(self.up_blocks)
        for bi, blk in enumerate(self.up_blocks):
            nr = len(blk.resnets)
            for j, res in enumerate(blk.resnets):
                sk, skp = skips.pop()
                attn = blk.has_cross_attention
                final = bi == nb - 1 and j == nr - 1            # feeds conv_norm_out only
                h, hp = res.forward_x3p(h, hp, trow(res), skip=sk, skip_p=skp, want_planes=not attn and not final)
                if attn:
                    h, hp = blk.attentions[j].forward_x3p(h, ctx, want_planes=not final)
            if blk.upsamplers is not None:
                h, hp = blk.upsamplers[0].forward_x3p(hp, want_planes=True)
            tap(f"up{bi}", h)
        h = self.conv_norm_out(h, silu=True)
        return hip.conv_out(h, self.conv_out.weight, self.conv_out.bias)

    def _cross_modules(self):
        c = getattr(self, "_cross_cache", None)
        if c is None:
            c = self._cross_cache = [m for m in self.attention_modules() if m.is_cross]
        return c

    def _kv_all_weight(self):
        w = getattr(self, "_w_kv_all", None)
        if w is None:
            w = self._w_kv_all = torch.cat([m.w_kv for m in self._cross_modules()], 0).contiguous()
        return w

    def _act(self, t32):
        """an fp32 device tensor in the model's activation dtype"""
        t32 = t32.float().contiguous()
        return t32 if self.dtype == torch.float32 else hip.to_f16(t32)

    def _ctx_f16(self, ctx):
        """context -> the model's activation dtype, once per distinct tensor (keeps the cross-attention K/V cache valid)."""
        cache = getattr(self, "_ctx_cache", None)
        if cache is None or cache[0] is not ctx or cache[1] != ctx._version:
            self._ctx_cache = (ctx, ctx._version, self._act(ctx))
        return self._ctx_cache[2]


class _Config:
    """`unet.config.in_channels`, `.sample_size` (sd_utils.py:16, pnp/model/sd_utils.py)."""

    def __init__(self, cfg: UNetConfig):
        self.in_channels = cfg.in_channels
        self.out_channels = cfg.out_channels
        self.sample_size = cfg.sample_size
        self.cross_attention_dim = cfg.cross_attention_dim
        self.block_out_channels = cfg.block_out_channels
        self.layers_per_block = cfg.layers_per_block
        self.attention_head_dim = cfg.num_heads
