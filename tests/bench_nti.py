"""Timing of the null-text-optimisation inner iteration (UNet forward + adjoint pass + Adam) at SD1.5 scale.

    python tests/bench_nti.py [--config sd15] [--iters 20] [--latent 64]
Prints one JSON line; used for DESIGN.md's NTI numbers and the rocprof summaries under profiles/.
"""
import argparse
import json
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ief_amd  # noqa: E402,F401
from ief_amd import hip  # noqa: E402
from ief_amd.nti import NullTextOptimizer  # noqa: E402
from ief_amd.pipeline import StableDiffusionPipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="sd15")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--latent", type=int, default=0)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--skip-full", action="store_true")
    ap.add_argument("--precision", default="f16", choices=["f16", "f16x3", "f32"],
                    help="f16x3 / f32: the reverse pass at the reference's precision (fp32 storage, csrc/backward_f32.hip)")
    a = ap.parse_args()
    pipe = StableDiffusionPipeline.from_pretrained(f"synthetic:{a.config}", precision=a.precision)
    cfg = pipe.cfg
    hw = a.latent or cfg.sample_size
    pipe.scheduler.set_timesteps(50)
    g = torch.Generator().manual_seed(0)
    ctx = torch.randn(2, 77, cfg.cross_attention_dim, generator=g) * 0.1
    lats = [torch.randn(1, 4, hw, hw, generator=g) for _ in range(51)]
    opt = NullTextOptimizer(pipe, ctx[1:], 7.5, (hw, hw), use_graph=not a.no_graph)
    opt.run(lats, ctx[:1], 1, 0.0, num_outer=1)          # capture + warm-up
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    for _ in range(a.iters):
        opt._run(1)
    ev[1].record()
    for _ in range(a.iters):
        opt._run(0)
    ev[2].record()
    torch.cuda.synchronize()
    inner_ms = ev[0].elapsed_time(ev[1]) / a.iters
    fwd_ms = ev[1].elapsed_time(ev[2]) / a.iters
    # a whole image the way edit_real.py runs it: 50 timesteps x (cond forward + 10 inner + tail forward)
    full_s = float("nan")
    if not a.skip_full:
        t0 = time.time()
        opt.run(lats, ctx[:1], 10, 0.0)
        torch.cuda.synchronize()
        full_s = time.time() - t0
    print(json.dumps({"workload": f"NTI {a.config} latent {hw}x{hw}", "precision": a.precision, "inner_iteration_ms": round(inner_ms, 3),
                      "plain_forward_B1_ms": round(fwd_ms, 3), "full_50x10_s": round(full_s, 2),
                      "inner_iterations_per_s": round(1000.0 / inner_ms, 2)}))


if __name__ == "__main__":
    main()
