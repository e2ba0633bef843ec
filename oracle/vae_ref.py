"""ORACLE (test infrastructure only — never imported by the product path).

Eager fp32 CPU restatement of diffusers' `AutoencoderKL` encode / decode [ext, un-vendored; parity unpinned against
diffusers itself] as used at `/root/reference/p2p/inversion/ddim.py:39-40` and `/root/reference/p2p/model/sd_utils.py:83-84`:
Encoder (conv_in, DownEncoderBlock2D x4 with pad-(0,1,0,1) stride-2 downsamples, mid block with single-head attention,
GroupNorm+SiLU, conv_out, quant_conv) and Decoder (post_quant_conv, conv_in, mid block, UpDecoderBlock2D x4 with
nearest-2x + conv, GroupNorm+SiLU, conv_out).  Weights: diffusers-keyed state dict (OIHW)."""
import torch
import torch.nn.functional as F


def _resnet(sd, p, x, G, eps):
    h = F.silu(F.group_norm(x, G, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps))
    h = F.conv2d(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = F.silu(F.group_norm(h, G, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps))
    h = F.conv2d(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if (p + ".conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return x + h


def _attn(sd, p, x, G, eps):
    B, C, H, W = x.shape
    h = F.group_norm(x, G, sd[p + ".group_norm.weight"], sd[p + ".group_norm.bias"], eps)
    h = h.reshape(B, C, H * W).transpose(1, 2)
    q = F.linear(h, sd[p + ".to_q.weight"], sd[p + ".to_q.bias"])
    k = F.linear(h, sd[p + ".to_k.weight"], sd[p + ".to_k.bias"])
    v = F.linear(h, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])
    a = torch.softmax(q @ k.transpose(1, 2) * C ** -0.5, -1) @ v
    a = F.linear(a, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return x + a.transpose(1, 2).reshape(B, C, H, W)


def _mid(sd, p, x, G, eps):
    x = _resnet(sd, p + ".resnets.0", x, G, eps)
    x = _attn(sd, p + ".attentions.0", x, G, eps)
    return _resnet(sd, p + ".resnets.1", x, G, eps)


def encode_mean(sd, cfg, image):
    G, eps, ch = cfg.norm_num_groups, cfg.eps, cfg.block_out_channels
    h = F.conv2d(image, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    for i in range(len(ch)):
        for j in range(cfg.layers_per_block):
            h = _resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", h, G, eps)
        if i < len(ch) - 1:
            q = f"encoder.down_blocks.{i}.downsamplers.0.conv"
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[q + ".weight"], sd[q + ".bias"], stride=2)
    h = _mid(sd, "encoder.mid_block", h, G, eps)
    h = F.silu(F.group_norm(h, G, sd["encoder.conv_norm_out.weight"], sd["encoder.conv_norm_out.bias"], eps))
    h = F.conv2d(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    h = F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])
    return h.chunk(2, dim=1)[0]


def decode(sd, cfg, z):
    G, eps, ch = cfg.norm_num_groups, cfg.eps, cfg.block_out_channels
    h = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    h = F.conv2d(h, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    h = _mid(sd, "decoder.mid_block", h, G, eps)
    for i in range(len(ch)):
        for j in range(cfg.layers_per_block + 1):
            h = _resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", h, G, eps)
        if i < len(ch) - 1:
            q = f"decoder.up_blocks.{i}.upsamplers.0.conv"
            h = F.conv2d(F.interpolate(h, scale_factor=2.0, mode="nearest"), sd[q + ".weight"], sd[q + ".bias"], padding=1)
    h = F.silu(F.group_norm(h, G, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], eps))
    return F.conv2d(h, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)
