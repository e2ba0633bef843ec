"""Reverse pass of the UNet with respect to `encoder_hidden_states` — what null-text inversion needs.

The reference differentiates `model.unet(latent_cur, t, uncond_embeddings)` with torch autograd and
updates ONLY the unconditional embedding (`/root/reference/p2p/inversion/nti.py:15-33`); every weight
is frozen.  So no weight gradient exists on this path and the backward is a chain of ACTIVATION
gradients, each of which is one of the forward kernels run on re-packed weights or a small adjoint
kernel (csrc/backward.hip, csrc/attention_bwd.hip):

  linear            dX = dY . W          = `hip.gemm(dY, W^T)`                 (W^T packed once)
  conv3x3 (s=1)     dX = conv3x3(dY, W') with W'[ci][ky][kx][co] = W[co][2-ky][2-kx][ci]
  conv3x3 (s=2)     the same on the zero-inserted gradient (`hip.zero_insert2x`)
  upsample + conv   conv data gradient, then the 2x2 block sum (`hip.pool2x2_sum`)
  GroupNorm(+SiLU)  `hip.groupnorm_bwd`  (channel-concat inputs split back into their two sources)
  LayerNorm         `hip.layernorm_bwd`
  GEGLU             `hip.geglu_il_bwd` on the interleaved FF1 projection
  attention         `hip.attn_bwd` from the forward's row log-sum-exp (no map is ever materialised)

Only what a backward needs is kept from the forward: the INPUTS of norms / GEGLU and q, k, v, o, lse of
each attention (no GEMM input is kept — there is no weight gradient).  The gradient of the context is
the sum over the 16 cross-attention layers of dK . W_k + dV . W_v; all layers' [dK | dV] are written
side by side into one [B, 77, sum 2C] buffer and contracted by ONE fp32-accumulating GEMM, so nothing
is summed in fp16.  Likewise all K/V projections of the forward are one GEMM.

Gradients are fp16 with a global scale: the loss kernel normalises d_eps to max |.| = `grad_scale`
and returns the factor that `hip.nti_adam` multiplies back in fp32.
"""
from typing import Dict, Optional

import torch

from . import hip


class UNetAdjoint:
    """mode "context" (default): d objective / d encoder_hidden_states given d objective / d eps (null-text inversion).
    mode "input": d objective / d sample where the objective is Pix2Pix-zero's cross-attention-map term
    (`/root/reference/pix2pix-zero/model/sd_utils.py:166-173`): the gradient ENTERS at the queries of every cross-attention
    module (`hip.attn_map_loss_bwd` against the reference maps), nothing enters at eps, and the chain runs all the way
    down through the first resnet and conv_in."""

    def __init__(self, unet, grad_scale: float = 1.0, mode: str = "context"):
        if mode not in ("context", "input"):
            raise ValueError("UNetAdjoint: mode must be 'context' or 'input'")
        # fp32-storage modes ("f32" / "f16x3"): same chain on the fp32 kernels (csrc/backward_f32.hip; attention gradients
        # on materialised fp32 maps, hip._attn_bwd_f32) -- the reference's own precision for this pass
        self.f32 = unet.dtype == torch.float32
        self.contract = getattr(unet, "contract", "f32")
        self.unet = unet
        self.mode = mode
        self.grad_scale = float(grad_scale)
        # packed adjoint weights are shared by every adjoint of this UNet (several images in flight: nti.run_many)
        shared = unet.__dict__.setdefault("_adjoint_shared", {})
        self._wt: Dict[int, torch.Tensor] = shared.setdefault("wt", {})
        cross = [m for m in unet.attention_modules() if m.is_cross]
        off = 0
        for m in cross:
            m._kv_off = off
            off += 2 * m.inner_dim
        self.kv_width = off
        if "w_kv_all" not in shared:
            shared["w_kv_all"] = torch.cat([m.w_kv for m in cross], 0).contiguous()      # [sum 2C, ctx_dim]
            shared["w_kv_all_t"] = shared["w_kv_all"].t().contiguous()                  # [ctx_dim, sum 2C]
        self.w_kv_all, self.w_kv_all_t = shared["w_kv_all"], shared["w_kv_all_t"]
        # backward stops at the FIRST transformer of the forward order: nothing upstream of it depends on the context
        self._first_tr = None
        for blk in unet.down_blocks:
            if blk.has_cross_attention:
                self._first_tr = blk.attentions[0]
                break
        if mode == "input":
            self._first_tr = None
        self.cross = cross
        self.ref_maps = None               # mode "input": per cross module (forward order) fp16 [B*heads, N, 77]
        self.loss_parts = None             # mode "input": fp32 partial sums of the objective (sum() = the loss)
        self.rec: Dict[int, tuple] = {}
        self.taps: Optional[dict] = None   # name -> max |grad| (debug / scale selection)

    # ------------------------------------------------------------------ packed adjoint weights (built once, on demand)
    def wt_lin(self, w):
        t = self._wt.get(id(w))
        if t is None:
            t = self._wt[id(w)] = w.t().contiguous()
        return t

    def wt_conv(self, w):
        t = self._wt.get(id(w))
        if t is None:
            t = self._wt[id(w)] = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()
        return t

    def wt_conv_in(self, w):
        """conv_in [3,3,Cin,Cout] -> the [Cin,3,3,Cout] weight of its data gradient (a Cout -> Cin 3x3 conv = `hip.conv_out`)"""
        t = self._wt.get(id(w))
        if t is None:
            t = self._wt[id(w)] = w.flip(0, 1).permute(2, 0, 1, 3).contiguous()
        return t

    def prepack(self):
        """Build every transposed weight now (otherwise the first backward does it)."""
        u = self.unet
        for r in u._resnets:
            self.wt_conv(r.conv1.weight), self.wt_conv(r.conv2.weight)
            if r.conv_shortcut is not None:
                self.wt_lin(r.conv_shortcut.weight)
        for blk in list(u.down_blocks) + list(u.up_blocks):
            for s in (blk.downsamplers if getattr(blk, "downsamplers", None) else []):
                self.wt_conv(s.conv.weight)
            for s in (blk.upsamplers if getattr(blk, "upsamplers", None) else []):
                self.wt_conv(s.conv.weight)
        for blk in list(u.down_blocks) + [u.mid_block] + list(u.up_blocks):
            for t in blk.attentions:
                self.wt_lin(t.proj_in.weight), self.wt_lin(t.proj_out.weight)
                for b in t.transformer_blocks:
                    for w in (b.attn1.w_qkv, b.attn1.to_out[0].weight, b.attn2.to_q.weight, b.attn2.to_out[0].weight,
                              b.ff.net[0].proj.weight, b.ff.net[2].weight):
                        self.wt_lin(w)
        if self.mode == "input":
            self.wt_conv_in(u.conv_in.weight)

    def _tap(self, name, t):
        if self.taps is not None and t is not None:
            self.taps[name] = float(t.float().abs().max())

    # ------------------------------------------------------------------ forward, keeping what the backward reads
    def _res_fwd(self, r, x, trow, skip=None):
        g1, st1 = r.norm1(x, silu=True, x2=skip, return_stats=True)
        h1 = hip.conv3x3(g1, r.conv1.weight, r.conv1.bias, rowvec=trow)
        g2, st2 = r.norm2(h1, silu=True, return_stats=True)
        if r.conv_shortcut is None:
            out = hip.conv3x3(g2, r.conv2.weight, r.conv2.bias, residual=x)
        else:
            out = hip.conv3x3_shortcut(g2, r.w2_fused, r.b2_fused, x, skip)
        self.rec[id(r)] = (x, skip, h1, st1, st2)
        return out

    def _tr_fwd(self, t, x, kv_all):
        B, H, W, C = x.shape
        N = H * W
        hn, st0 = t.norm(x, return_stats=True)
        h = t.proj_in(hn).reshape(B, N, C)
        recs = []
        for blk in t.transformer_blocks:            # SDXL stacks several blocks between proj_in and proj_out
            a1, a2, ff = blk.attn1, blk.attn2, blk.ff
            h0 = h
            qkv = hip.gemm(blk.norm1(h0), a1.w_qkv)
            # fp16: the row log-sum-exp feeds the fused backward kernels; fp32 storage: the backward re-materialises the maps, except
            # where the split-operand mode has its own fused recomputing kernel (hip.x3_fused_bwd_ok: the large self-attention levels)
            want_lse = not self.f32 or hip.x3_fused_bwd_ok(C // a1.heads, N)
            lse1 = torch.empty(B, a1.heads, N, dtype=torch.float32, device=x.device) if want_lse else None
            o1 = hip.attn_flash(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], a1.heads, a1.scale, lse=lse1)
            h1 = a1.to_out[0](o1, residual=h0)
            q2 = hip.gemm(blk.norm2(h1), a2.to_q.weight)
            off = a2._kv_off
            lse2 = None if self.f32 else torch.empty(B, a2.heads, N, dtype=torch.float32, device=x.device)
            o2 = hip.attn_flash(q2, kv_all[..., off:off + C], kv_all[..., off + C:off + 2 * C], a2.heads, a2.scale, lse=lse2)
            h2 = a2.to_out[0](o2, residual=h1)
            pre = hip.gemm(blk.norm3(h2), ff.net[0].proj.weight, bias=ff.net[0].proj.bias)
            h = ff.net[2](hip.geglu_il(pre), residual=h2)
            recs.append((h0, qkv, o1, lse1, h1, q2, o2, lse2, h2, pre))
        out = t.proj_out(h.reshape(B, H, W, C), residual=x)
        self.rec[id(t)] = (x, st0, recs)
        return out

    def forward(self, sample, temb_row, ctx16):
        with hip.f32_contraction(self.contract):
            return self._forward(sample, temb_row, ctx16)

    def backward(self, d_eps):
        with hip.f32_contraction(self.contract):
            return self._backward(d_eps)

    def _forward(self, sample, temb_row, ctx16):
        """sample fp32 NCHW [B,4,h,w]; temb_row fp32 [1, width] (`unet.time_rows`); ctx16 [B,77,Cc] in the model's activation
        dtype (fp16; fp32 in the fp32-storage modes) -> eps fp32 NCHW."""
        u = self.unet
        if u._plan is not None or not all(m.is_native() for m in u.attention_modules()):
            raise RuntimeError("UNetAdjoint: an attention controller is registered; the reference runs null-text "
                               "optimisation before any controller is installed (edit_real.py)")
        trow = lambda r: temb_row[:, r.temb_slot[0]:r.temb_slot[0] + r.temb_slot[1]].contiguous()
        self.rec.clear()
        self.kv_all = hip.gemm(ctx16, self.w_kv_all)
        h = hip.conv_in(sample, u.conv_in.weight, u.conv_in.bias)
        skips = [h]
        for blk in u.down_blocks:
            for j, res in enumerate(blk.resnets):
                h = self._res_fwd(res, h, trow(res))
                if blk.has_cross_attention:
                    h = self._tr_fwd(blk.attentions[j], h, self.kv_all)
                skips.append(h)
            if blk.downsamplers is not None:
                h = blk.downsamplers[0](h)
                skips.append(h)
        mb = u.mid_block
        h = self._res_fwd(mb.resnets[0], h, trow(mb.resnets[0]))
        h = self._tr_fwd(mb.attentions[0], h, self.kv_all)
        h = self._res_fwd(mb.resnets[1], h, trow(mb.resnets[1]))
        for blk in u.up_blocks:
            for j, res in enumerate(blk.resnets):
                h = self._res_fwd(res, h, trow(res), skip=skips.pop())
                if blk.has_cross_attention:
                    h = self._tr_fwd(blk.attentions[j], h, self.kv_all)
            if blk.upsamplers is not None:
                h = blk.upsamplers[0](h)
        hn, stf = u.conv_norm_out(h, silu=True, return_stats=True)
        self.rec["final"] = (h, stf)
        return hip.conv_out(hn, u.conv_out.weight, u.conv_out.bias)

    # ------------------------------------------------------------------ backward
    def _res_bwd(self, r, d_out):
        x, skip, h1, st1, st2 = self.rec[id(r)]
        n1, n2 = r.norm1, r.norm2
        d_g2 = hip.conv3x3(d_out, self.wt_conv(r.conv2.weight))
        d_h1 = hip.groupnorm_bwd(h1, d_g2, n2.weight, n2.bias, n2.num_groups, n2.eps, silu=True, stats=st2)
        d_g1 = hip.conv3x3(d_h1, self.wt_conv(r.conv1.weight))
        add = d_out if r.conv_shortcut is None else hip.gemm(d_out, self.wt_lin(r.conv_shortcut.weight))
        res = hip.groupnorm_bwd(x, d_g1, n1.weight, n1.bias, n1.num_groups, n1.eps, silu=True, x2=skip, add=add, stats=st1)
        return res if skip is not None else (res, None)

    def _blk_bwd(self, blk, rec, d_h3, B, N, C, stop=False):
        """one BasicTransformerBlock: gradient w.r.t. its output -> gradient w.r.t. its input (None when `stop`)"""
        h0, qkv, o1, lse1, h1, q2, o2, lse2, h2, pre = rec
        a1, a2, ff = blk.attn1, blk.attn2, blk.ff
        d_gg = hip.gemm(d_h3, self.wt_lin(ff.net[2].weight))
        d_n3 = hip.gemm(hip.geglu_il_bwd(pre, d_gg), self.wt_lin(ff.net[0].proj.weight))
        d_h2 = hip.layernorm_bwd(h2, d_n3, blk.norm3.weight, blk.norm3.eps, add=d_h3)
        d_o2 = hip.gemm(d_h2, self.wt_lin(a2.to_out[0].weight))
        off = a2._kv_off
        k2 = self.kv_all[..., off:off + C]
        if self.mode == "input":
            dq2, _, _ = hip.attn_bwd(q2, k2, self.kv_all[..., off + C:off + 2 * C], o2, d_o2, lse2, a2.heads, a2.scale,
                                     want_dkv=False)
            i = self._cross_index[id(a2)]
            parts = self.loss_parts[self._loss_off[i]:self._loss_off[i + 1]]
            hip.attn_map_loss_bwd(q2, k2, self.ref_maps[i], dq2, a2.heads, a2.scale,
                                  gcoef=2.0 * self.grad_scale / (B * a2.heads), accumulate=True, loss=parts,
                                  loss_coef=1.0 / (B * a2.heads))
        else:
            dq2, _, _ = hip.attn_bwd(q2, k2, self.kv_all[..., off + C:off + 2 * C], o2, d_o2, lse2,
                                     a2.heads, a2.scale, dk=self.dkv_all[..., off:off + C],
                                     dv=self.dkv_all[..., off + C:off + 2 * C], want_dq=not stop)
        if stop:
            return None
        d_h1 = hip.layernorm_bwd(h1, hip.gemm(dq2, self.wt_lin(a2.to_q.weight)), blk.norm2.weight, blk.norm2.eps, add=d_h2)
        d_o1 = hip.gemm(d_h1, self.wt_lin(a1.to_out[0].weight))
        dqkv = torch.empty_like(qkv)
        hip.attn_bwd(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], o1, d_o1, lse1, a1.heads, a1.scale,
                     dq=dqkv[..., :C], dk=dqkv[..., C:2 * C], dv=dqkv[..., 2 * C:])
        return hip.layernorm_bwd(h0, hip.gemm(dqkv, self.wt_lin(a1.w_qkv)), blk.norm1.weight, blk.norm1.eps, add=d_h1)

    def _tr_bwd(self, t, d_out, stop=False):
        """stop (mode "context"): t is the first transformer of the forward; nothing upstream of its FIRST block's
        cross-attention depends on the context, so the chain ends there"""
        x, st0, recs = self.rec[id(t)]
        B, H, W, C = x.shape
        N = H * W
        d_h = hip.gemm(d_out.reshape(B, N, C), self.wt_lin(t.proj_out.weight))
        for k in reversed(range(len(t.transformer_blocks))):
            d_h = self._blk_bwd(t.transformer_blocks[k], recs[k], d_h, B, N, C, stop=stop and k == 0)
        if d_h is None:
            return None
        d_hn = hip.gemm(d_h, self.wt_lin(t.proj_in.weight)).reshape(B, H, W, C)
        n = t.norm
        return hip.groupnorm_bwd(x, d_hn, n.weight, n.bias, n.num_groups, n.eps, silu=False, add=d_out, stats=st0)

    def set_reference_maps(self, ref_maps):
        """mode "input": the maps the objective compares against, one per cross-attention module in forward order"""
        if len(ref_maps) != len(self.cross):
            raise ValueError(f"UNetAdjoint: {len(self.cross)} cross-attention modules, {len(ref_maps)} reference maps")
        self.ref_maps = list(ref_maps)
        self._cross_index = {id(m): i for i, m in enumerate(self.cross)}
        off = [0]
        for m, r in zip(self.cross, ref_maps):
            off.append(off[-1] + (hip.map_loss_blocks_f32(r.shape[0] * r.shape[1]) if self.f32 else
                                  r.shape[0] * hip.map_loss_blocks(r.shape[1], m.dim_head)))
        self._loss_off = off
        if self.loss_parts is None or self.loss_parts.numel() != off[-1]:
            self.loss_parts = torch.zeros(off[-1], dtype=torch.float32, device=ref_maps[0].device)

    def _backward(self, d_eps):
        """mode "context": d_eps fp32 NCHW (gradient of the objective w.r.t. the UNet output, already scaled) -> fp16
        [B,77,Cc]: the gradient w.r.t. the fp16 context the last `forward` ran on, in the same scale.
        mode "input": d_eps is the (usually zero) gradient entering at eps; -> fp32 NCHW gradient w.r.t. `sample`,
        times `grad_scale`; `loss_parts.sum()` is the objective's value."""
        u = self.unet
        hf, stf = self.rec["final"]
        B = hf.shape[0]
        if self.mode == "input":
            if self.ref_maps is None:
                raise RuntimeError("UNetAdjoint(mode='input'): set_reference_maps() first")
        else:
            self.dkv_all = torch.empty(B, self.kv_all.shape[1], self.kv_width, dtype=self.unet.dtype, device=hf.device)
        n = u.conv_norm_out
        d = hip.conv_out_bwd(d_eps, u.conv_out.weight)
        d = hip.groupnorm_bwd(hf, d, n.weight, n.bias, n.num_groups, n.eps, silu=True, stats=stf)
        self._tap("conv_norm_out", d)
        dskips = []
        for bi in reversed(range(len(u.up_blocks))):
            blk = u.up_blocks[bi]
            if blk.upsamplers is not None:
                d = hip.pool2x2_sum(hip.conv3x3(d, self.wt_conv(blk.upsamplers[0].conv.weight)))
            for j in reversed(range(len(blk.resnets))):
                if blk.has_cross_attention:
                    d = self._tr_bwd(blk.attentions[j], d)
                d, ds = self._res_bwd(blk.resnets[j], d)
                dskips.append(ds)
            self._tap(f"up{bi}", d)
        mb = u.mid_block
        d, _ = self._res_bwd(mb.resnets[1], d)
        d = self._tr_bwd(mb.attentions[0], d)
        d, _ = self._res_bwd(mb.resnets[0], d)
        self._tap("mid", d)
        done = False
        for bi in reversed(range(len(u.down_blocks))):
            blk = u.down_blocks[bi]
            if blk.downsamplers is not None:
                d = hip.add(d, dskips.pop())
                d = hip.conv3x3(hip.zero_insert2x(d), self.wt_conv(blk.downsamplers[0].conv.weight))
            for j in reversed(range(len(blk.resnets))):
                d = hip.add(d, dskips.pop())
                if blk.has_cross_attention:
                    t = blk.attentions[j]
                    d = self._tr_bwd(t, d, stop=t is self._first_tr)
                    if t is self._first_tr:
                        done = True
                        break
                d, _ = self._res_bwd(blk.resnets[j], d)
            self._tap(f"down{bi}", d)
            if done:
                break
        if self.mode == "input":
            d = hip.add(d, dskips.pop())               # conv_in's output is also the first skip connection
            return hip.conv_out(d, self.wt_conv_in(u.conv_in.weight), None)
        return hip.gemm(self.dkv_all, self.w_kv_all_t)
