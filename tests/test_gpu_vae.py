"""AutoencoderKL on the HIP kernels vs the fp32 CPU oracle (image2latent / latent2image)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from ief_amd import hip  # noqa: E402
from ief_amd.vae import AutoencoderKL, SD_VAE, TINY_VAE, synthetic_vae_state_dict, vae_param_shapes  # noqa: E402
from oracle import vae_ref  # noqa: E402
from oracle.p2p_ref import latent_to_uint8  # noqa: E402


def rel(got, ref):
    got, ref = got.float().cpu(), ref.float()
    assert got.shape == ref.shape and torch.isfinite(got).all()
    return ((got - ref).abs().max() / ref.abs().max()).item()


def test_small_kernels():
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(300, 1024, generator=g) * 3).half()
    y = hip.softmax_rows_(x.clone().cuda())
    assert rel(y, torch.softmax(x.float(), -1)) < 2e-3
    t = torch.randn(70, 200, generator=g).half()
    assert torch.equal(hip.transpose(t.cuda()).cpu(), t.t().contiguous())
    z = torch.randn(2, 8, 5, 7, generator=g)
    w, b = torch.randn(4, 8, generator=g), torch.randn(4, generator=g)
    got = hip.pointwise_f32(z.cuda(), w.cuda(), b.cuda())
    assert rel(got, torch.einsum("oc,bchw->bohw", w, z) + b[None, :, None, None]) < 1e-6
    # pad (0,1,0,1) stride-2 conv
    xa = (torch.randn(2, 10, 14, 64, generator=g)).half()
    wa = (torch.randn(128, 3, 3, 64, generator=g) / 24).half()
    got = hip.conv3x3(xa.cuda(), wa.cuda(), None, stride=2, pad_hi_only=True)
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(xa.float().permute(0, 3, 1, 2), (0, 1, 0, 1)),
                                     wa.float().permute(0, 3, 1, 2), stride=2).permute(0, 2, 3, 1)
    assert rel(got, ref) < 2e-3


@pytest.mark.parametrize("cfg,size,B", [(TINY_VAE, 64, 2), (SD_VAE, 256, 1)])
def test_vae_encode_decode_vs_oracle(cfg, size, B):
    sd = synthetic_vae_state_dict(cfg, 2)
    vae = AutoencoderKL(cfg, sd)
    g = torch.Generator().manual_seed(1)
    img = torch.rand(B, 3, size, size, generator=g) * 2 - 1
    lat = vae.encode(img.cuda())["latent_dist"].mean
    lat_ref = vae_ref.encode_mean(sd, cfg, img)
    e1 = rel(lat, lat_ref)
    z = torch.randn(B, 4, size // 8, size // 8, generator=g)
    dec = vae.decode(z.cuda())["sample"]
    dec_ref = vae_ref.decode(sd, cfg, z)
    e2 = rel(dec, dec_ref)
    print(f"VAE {cfg.block_out_channels} {size}px: encode rel err {e1:.3e}, decode rel err {e2:.3e}")
    assert lat.dtype == torch.float32 and dec.dtype == torch.float32
    assert e1 < 1e-2 and e2 < 1e-2
    # the uint8 image epilogue (sd_utils.py:85-88): at most one grey level apart wherever the fp16 path rounds differently
    u_got, u_ref = latent_to_uint8(dec.cpu()), latent_to_uint8(dec_ref)
    assert (abs(u_got.astype(int) - u_ref.astype(int)) <= 2).mean() > 0.999


def test_vae_param_names_are_diffusers_keys():
    names = list(vae_param_shapes(SD_VAE))
    for k in ("encoder.conv_in.weight", "encoder.down_blocks.0.downsamplers.0.conv.weight", "quant_conv.weight",
              "encoder.mid_block.attentions.0.to_q.weight", "decoder.up_blocks.3.resnets.2.conv2.bias",
              "decoder.up_blocks.2.resnets.0.conv_shortcut.weight", "post_quant_conv.bias", "decoder.conv_out.weight"):
        assert k in names
    n = sum(int(torch.tensor(s).prod()) for s in vae_param_shapes(SD_VAE).values())
    assert n == 83_653_863      # parameter count of the SD1.x AutoencoderKL
