"""Launch ONE kernel shape a few times (for rocprofv3 --pmc runs).  usage: one_kernel.py attn40|conv64|conv32|gemmff|gemmsq[x3|f32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ief_amd
from ief_amd import hip
DEV = torch.device("cuda:0")
h = lambda *s, scale=1.0: (torch.randn(*s, device=DEV) * scale).half()
what = sys.argv[1]
if what == "attn40":
    B, heads, N, d = 4, 8, 4096, 40
    qkv = h(B, N, 3 * heads * d); C = heads * d
    fn = lambda: hip.attn_flash(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, d ** -0.5)
elif what == "conv64":
    x, w, b = h(4, 64, 64, 320), h(320, 3, 3, 320, scale=0.02), torch.randn(320, device=DEV)
    fn = lambda: hip.conv3x3(x, w, b)
elif what in ("conv64t7", "conv64halo", "conv64igemm"):
    x, w, b = h(4, 64, 64, 320), h(320, 3, 3, 320, scale=0.02), torch.randn(320, device=DEV)
    fn = (lambda: hip.conv3x3(x, w, b, tile_hint=7, splits=1, stages=3)) if what in ("conv64t7", "conv64igemm") else \
         (lambda: hip.conv3x3(x, w, b, tile_hint=15, splits=1, stages=4))
elif what == "conv32":
    x, w, b = h(4, 32, 32, 640), h(640, 3, 3, 640, scale=0.02), torch.randn(640, device=DEV)
    fn = lambda: hip.conv3x3(x, w, b)
elif what == "gemmff":
    a, w = h(4096, 640), h(5120, 640, scale=0.04)
    fn = lambda: hip.gemm(a, w)
elif what == "gemmsq":      # the most frequent linear of the step: a square projection with bias + residual (to_out / proj_out)
    a, w, b, r = h(4, 4096, 320), h(320, 320, scale=0.05), torch.randn(320, device=DEV), h(4, 4096, 320)
    fn = lambda: hip.gemm(a, w, bias=b, residual=r)
elif what.endswith("x3p"):      # the planes kernels (csrc/gemm_x3p.hip); optional tile id after a colon: conv64x3p:2
    from ief_amd import planes
    tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    f = lambda *s, scale=1.0: torch.randn(*s, device=DEV) * scale
    base = what[:-3]
    with hip.f32_contraction("x3"):
        if base == "conv64":
            x, w, b = planes.split(f(4, 64, 64, 320)), f(320, 3, 3, 320, scale=0.02), torch.randn(320, device=DEV)
            g = lambda: planes.conv3x3(x, w, b, tile=tile)
        elif base == "conv32":
            x, w, b = planes.split(f(4, 32, 32, 640)), f(640, 3, 3, 640, scale=0.02), torch.randn(640, device=DEV)
            g = lambda: planes.conv3x3(x, w, b, tile=tile)
        elif base == "gemmsq":
            a, w, b, r = planes.split(f(4, 4096, 320)), f(320, 320, scale=0.05), torch.randn(320, device=DEV), f(4, 4096, 320)
            g = lambda: planes.gemm(a, w, bias=b, residual=r, tile=tile)
        elif base == "gemmff":
            a, w = planes.split(f(4096, 640)), f(5120, 640, scale=0.04)
            g = lambda: planes.gemm(a, w, tile=tile)
        elif base in ("attn40", "attn40long"):
            B, heads, N, d = 4, 8, (16384 if base == "attn40long" else 4096), 40
            qkv = planes.split(f(B, N, 3 * heads * d)); C = heads * d
            g = lambda: planes.attn_flash(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, d ** -0.5)

    def fn():
        with hip.f32_contraction("x3"):
            g()
elif what.endswith("x3") or what.endswith("f32"):      # fp32-storage modes: split-operand ("x3") or fp32-MFMA contractions
    mode = "x3" if what.endswith("x3") else "f32"
    base = what[:-2] if mode == "x3" else what[:-3]
    f = lambda *s, scale=1.0: torch.randn(*s, device=DEV) * scale
    if base == "conv64":
        x, w, b = f(4, 64, 64, 320), f(320, 3, 3, 320, scale=0.02), torch.randn(320, device=DEV)
        g = lambda: hip.conv3x3(x, w, b)
    elif base == "conv32":
        x, w, b = f(4, 32, 32, 640), f(640, 3, 3, 640, scale=0.02), torch.randn(640, device=DEV)
        g = lambda: hip.conv3x3(x, w, b)
    elif base == "gemmsq":
        a, w, b, r = f(4, 4096, 320), f(320, 320, scale=0.05), torch.randn(320, device=DEV), f(4, 4096, 320)
        g = lambda: hip.gemm(a, w, bias=b, residual=r)
    elif base == "gemmff":
        a, w = f(4096, 640), f(5120, 640, scale=0.04)
        g = lambda: hip.gemm(a, w)
    elif base == "attn40":
        B, heads, N, d = 4, 8, 4096, 40
        qkv = f(B, N, 3 * heads * d); C = heads * d
        g = lambda: hip.attn_flash(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, d ** -0.5)

    def fn():
        with hip.f32_contraction(mode):
            g()
for _ in range(4):
    fn()
torch.cuda.synchronize()
