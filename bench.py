"""Headline benchmark: denoising steps/sec of the SD1.5 512x512 Prompt-to-Prompt edit loop.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

The headline (`value`, `roofline`) is measured in the mode that meets north_star's tolerance -- edited images within 1e-3
max-abs of the fp32 reference (`/root/reference/p2p/edit_syn.py:38` computes in fp32): `--precision f16x3`, fp32 storage with
every contraction on split fp16 operands (three fp16 MFMAs per product, fp32 accumulate; csrc/split_x3.hip; a 50-step
512x512 edit ends 4e-6 from the fp32 oracle's images, tests/test_gpu_zz_fullsize.py).  The fp16-storage path (3e-3 on the same
edit: outside the bound) is reported beside it as `fp16_mode`, the fp32-input-MFMA mode as `exact_mode`.

A "step" is one pass of the hot path over one batch: one denoising step of the P2P edit loop =
UNet forward at batch 4 (2 prompts x classifier-free guidance) with the AttentionRefine controller
active + CFG combine + DDIM update (`/root/reference/p2p/model/sd_utils.py:67-79`), SD1.5-shaped
UNet (859.5 M parameters), 64x64 latents (512x512 images), 50-step schedule, synthetic seeded
weights / latents / embeddings resident in HBM before the timed region.

N > 1: one process per GPU; every rank runs its OWN edit (PIE-Bench images are independent:
`/root/reference/p2p/test.py:114-181`), so there is no data-path collective; the only collective is
the one-time RCCL broadcast of the packed weights from rank 0.  value = total steps of all ranks
divided by the slowest rank's time ("weak" scaling).

The JSON line also carries
  roofline      the kernel TEMPLATE with the largest share of the step (all tile / ring-depth instantiations of one
                template count as one kernel): sum of algorithmic FLOP / sum of launch durations measured live with HIP
                events on the launch stream, against the matrix peak of the mode: 2.5 PFLOP/s dense fp16 MFMA for the
                fp16 path, a third of it for f16x3 (three MFMAs per algorithmic product), 157.3 TFLOP/s for the fp32-input
                MFMA; `frac_rocprof` is the same fraction from the committed rocprofv3 kernel-trace of this command
                (profiles/), `roofline_frac` prices every launch against min(matrix peak, arithmetic intensity x HBM peak)
  cpu_baseline  the fp32 CPU oracle (`oracle/`, reference execution semantics: materialised maps +
                Python controller) timed on this host's cores on a bounded sample of the same workload (>= 3 steps)
  images_per_sec          PIE-Bench loop of `/root/reference/p2p/test.py:114-181` on synthetic 512x512 images, in the
                          reference's per-image order (invert at UNet batch 1, edit at batch 4, VAE both ends)
  images_per_sec_batched  the same images with 4 inverted per batched DDIM loop and 4 edits in flight (same pixels)
  steps_per_sec_1024      the same edit step on 128x128 latents (1024x1024 px), with its own `roofline_1024`
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PROMPTS = ["a photo of a house on a mountain", "a photo of a house on a mountain at fall"]  # edit_syn.py:20-21
FLOP_PER_STEP = 3.213e12      # 0.803 TFLOP / sample-forward x 4
PEAK_MFMA_F16 = 2.5e15        # MI355X_MICROARCH.md: dense fp16/bf16 MFMA peak
PEAK_MFMA_F32 = 157.3e12      # MI355X_MICROARCH.md: fp32-input MFMA (= fp32 vector) peak
# per mode: (algorithmic-FLOP peak of its contractions, the line's "dtype", what the arithmetic is)
MODES = {
    "f16": (PEAK_MFMA_F16, "f16", "fp16 storage, fp16 MFMA operands, fp32 accumulate / statistics"),
    "f16x3": (PEAK_MFMA_F16 / 3.0, "f16x3", "fp32 storage; every contraction on split operands: hi + lo fp16 halves of each fp32 "
              "element, Ah Bh + Al Bh + Ah Bl = three v_mfma_f32_*_f16 per product, fp32 accumulate; softmax / norms in fp32"),
    "f32": (PEAK_MFMA_F32, "f32", "fp32 storage, fp32-input MFMA (v_mfma_f32_32x32x2_f32)"),
}
# NOT measured by this program: the 50-step image error of each mode as tests/test_gpu_zz_fullsize.py::test_sd15_edit50_vs_oracle_fixture
# last measured it (a 50-step edit + VAE decode against fixture G13).  What this program measures itself is `parity_probe` below.
IMAGE_ERR_50_GPUTEST = {"f16x3": 4.3e-6, "f32": 5.3e-6, "f16": 3.0e-3}
# bound on the latents after the probe's ten steps, relative to max |latent| (the full-size test's bounds, tests/test_gpu_zz_fullsize.py)
PROBE_BOUND = {"f16x3": 1e-5, "f32": 1e-5, "f16": 4e-3}
MAX_STEPS = 50                # controller tables cover one 50-step edit


def parity_probe(pipe, dev, precision):
    """Parity measured IN THIS RUN: the first ten steps of the reference's unit of work (fixture G13, tests/golden/sd15_edit50.npz:
    SD1.5 512x512, CLI default prompts, AttentionRefine 0.8 / 0.4, guidance 7.5, seed 8888, `/root/reference/p2p/model/sd_utils.py:24-79`)
    in the captured step graph of the timed configuration, against the latents the fp32 CPU oracle reached after ten steps.
    ~0.2 s of GPU time.  Raises when the error exceeds the mode's bound: a kernel change that breaks parity breaks the bench."""
    import numpy as np
    from ief_amd.denoise import acquire
    from ief_amd.p2p.model.attention_control import AttentionRefine
    from ief_amd.p2p.model.register import register_attention_control, unregister_attention_control
    from ief_amd.p2p.model.sd_utils import _encode_prompts
    path = os.path.join(ROOT, "tests", "golden", "sd15_edit50.npz")
    g = np.load(path)
    with torch.no_grad():
        u, c = _encode_prompts(pipe, PROMPTS)
    context = torch.cat([u, c]).float()
    if not torch.allclose(context[:, :8, :16].cpu(), torch.from_numpy(g["context_probe"]), rtol=0, atol=1e-6):
        raise SystemExit("parity probe: the regenerated context is not the fixture's (seeded text encoder changed?)")
    x_T = torch.from_numpy(g["x_T"]).to(dev)
    ctl = AttentionRefine(PROMPTS, pipe.tokenizer, MAX_STEPS, 0.8, 0.4, device=dev)
    register_attention_control(pipe, ctl)
    loop = acquire(pipe, context.to(dev), 2, (x_T.shape[-2], x_T.shape[-1]), 7.5)
    lat = loop.run(x_T.expand(2, -1, -1, -1), num_steps=10)
    loop.release()
    unregister_attention_control(pipe, ctl)
    ref = torch.from_numpy(g["lat_10"])
    err = ((lat.float().cpu() - ref).abs().max() / ref.abs().max()).item()
    bound = PROBE_BOUND[precision]
    if not (err <= bound):
        raise SystemExit(f"parity probe FAILED: latents after 10 steps are {err:.3e} (relative) from the fp32 oracle's, bound {bound:.0e} "
                         f"for precision {precision}")
    return {"what": "latents after the first 10 steps of the 50-step SD1.5 512x512 P2P edit (fixture G13) in the captured step graph vs "
                    "the fp32 CPU oracle's, max |diff| / max |ref|; measured in this run",
            "lat10_rel_err": float(f"{err:.3e}"), "bound": bound, "fixture": "tests/golden/sd15_edit50.npz"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=str, default="sd15")
    ap.add_argument("--latent", type=int, default=0, help="latent side (default: the config's sample_size; 128 = 1024x1024 px)")
    ap.add_argument("--uncond", choices=["per-step", "fixed"], default="per-step",
                    help="per-step: the unconditional embedding changes every step, as after null-text inversion "
                    "(P2P_NTI, BASELINE.json configs[1]): cross-attention K/V are re-projected inside every step; "
                    "fixed: plain P2P (edit_syn.py), K/V projected once per edit")
    ap.add_argument("--in-flight", type=str, default="2,4", help="also time E independent edits stepped concurrently per GPU "
                    "(comma list, '' to skip); reported beside the headline value, which is ONE edit at a time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--split", choices=["none", "cfg"], default="none",
                    help="cfg: --gpus 2 run ONE edit together (rank 0 the unconditional rows of the CFG batch, rank 1 the "
                    "conditional rows + controller plan, one 128 KiB eps all-gather per step over RCCL; SURVEY.md 8e); "
                    "value is then the steps/s of that one edit (strong scaling)")
    ap.add_argument("--precision", choices=["f16x3", "f16", "f32"], default="f16x3",
                    help="the mode the headline value is measured in (default: the one that meets the 1e-3 image bound)")
    ap.add_argument("--other-modes", type=str, default="f16,f32",
                    help="modes reported beside the headline (N = 1 only): f16 -> `fp16_mode` with its own roofline, in-flight "
                    "throughput, 1024x1024 and PIE numbers; f32 -> `exact_mode` (a few steps + roofline); '' to skip")
    ap.add_argument("--exact-steps", type=int, default=5, help="timed steps of the fp32-input-MFMA mode")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo lets two ranks share one "
                    "GPU, which RCCL refuses: rehearsals of the N > 1 paths on a one-GPU box)")
    ap.add_argument("--pie-images", type=int, default=8, help="synthetic PIE images timed per GPU for images_per_sec (0: skip)")
    ap.add_argument("--steps-1024", type=int, default=10, help="timed edit steps on 128x128 latents (0: skip)")
    ap.add_argument("--nti-images", type=int, default=1, help="BASELINE.json configs[1]: synthetic images timed through null-text "
                    "inversion + edit in the headline mode (0: skip)")
    return ap.parse_args()


def build_pipe(cfg_name, dev, rank, world, precision="f16"):
    """rank 0 draws the synthetic weights; other ranks receive the PACKED device tensors by RCCL broadcast."""
    import ief_amd  # noqa: F401
    from ief_amd import config, weights
    from ief_amd.pipeline import StableDiffusionPipeline, SyntheticTextEncoder
    from ief_amd.vae import AutoencoderKL, SD_VAE, TINY_VAE
    from ief_amd.scheduler import DDIMScheduler
    from ief_amd.tokenizer import WordPieceTokenizer
    from ief_amd.unet import UNet2DConditionModel
    cfg = config.CONFIGS[cfg_name]
    if rank == 0:
        sd = weights.synthetic_state_dict(cfg, 0)
    else:
        sd = {k: torch.zeros(s) for k, s in weights.unet_param_shapes(cfg).items()}
    unet = UNet2DConditionModel(cfg, sd, device=dev, precision=precision)
    pipe = StableDiffusionPipeline(unet, WordPieceTokenizer(cfg.text_max_length),
                                   SyntheticTextEncoder(cfg.cross_attention_dim).to(dev),
                                   AutoencoderKL(SD_VAE if cfg_name in ("sd15", "sd21") else TINY_VAE, device=dev, precision=precision),
                                   DDIMScheduler(), cfg, sd if rank == 0 else None)
    if world > 1:
        from ief_amd.dist import broadcast_pipeline
        broadcast_pipeline(pipe, src=0)   # RCCL over xGMI: UNet, VAE and text encoder as a few flat buckets (the drivers' collective)
        torch.cuda.synchronize()
    return pipe, cfg


def make_added(cfg, hw, rank, dev):
    """SDXL family: pooled text embedding + six time ids per UNet batch row (P2P_XL.encode_prompt_xl); constant over the
    steps, folded by the loop into its per-step time-embedding rows.  None for the other families."""
    if not cfg.addition_embed:
        return None
    g = torch.Generator().manual_seed(4321 + rank)
    size = float(hw * 8)
    return {"text_embeds": (torch.randn(4, cfg.pooled_text_dim, generator=g) * 0.5).to(dev),
            "time_ids": torch.tensor([[size, size, 0.0, 0.0, size, size]] * 4, device=dev)}


class EditJob:
    """one P2P AttentionRefine edit (controller + captured step loop) on `hw` x `hw` latents"""

    def __init__(self, pipe, cfg, ctx, hw, dev, rank, uncond_list, split_group=None):
        from ief_amd.denoise import CfgSplitDenoiser, FusedDenoiser
        from ief_amd.p2p.model.attention_control import AttentionRefine
        from ief_amd.p2p.model.register import register_attention_control
        self.pipe, self.hw = pipe, hw
        # every rank edits its own image: same prompts, rank-specific x_T (seed 8888 + rank, edit_syn.py:19); the two
        # ranks of a CFG split share ONE image
        seed = 8888 + (rank if split_group is None else 0)
        self.x_T = torch.randn(1, 4, hw, hw, generator=torch.Generator().manual_seed(seed)).to(dev)
        self.ctrl = AttentionRefine(PROMPTS, pipe.tokenizer, MAX_STEPS, 0.8, 0.4, device=dev)
        self.added = make_added(cfg, hw, rank, dev)
        if split_group is not None:
            register_attention_control(pipe, self.ctrl, rows="cond" if rank == 1 else "uncond")
            self.loop = CfgSplitDenoiser(pipe, ctx, 2, (hw, hw), 7.5, group=split_group, uncond_list=uncond_list)
        else:
            register_attention_control(pipe, self.ctrl)
            self.loop = FusedDenoiser(pipe, ctx, 2, (hw, hw), 7.5, uncond_list=uncond_list, added_cond_kwargs=self.added)
        self.loop.run(self.x_T, num_steps=0)      # allocates, warms up eagerly (untimed) and captures the graph

    def rewind(self):
        self.ctrl.reset()
        self.loop.rewind(self.x_T)

    def run_steps(self, n):
        """n steps, restarting the edit (controller + step counters) whenever 50 are used up"""
        done = 0
        while done < n:
            room = MAX_STEPS - self.ctrl.cur_step
            if room == 0:
                self.ctrl.reset()
                self.loop.rewind()
                room = MAX_STEPS
            k = min(room, n - done)
            for _ in range(k):
                self.loop.step_once()
            done += k

    def timed(self, steps, warmup, barrier, dist, dev):
        """(seconds for `steps` steps: max over ranks)"""
        self.run_steps(warmup)
        self.rewind()
        barrier()
        t0 = time.perf_counter()
        self.run_steps(steps)
        barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            te = torch.tensor([elapsed], device=dev)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            elapsed = te.item()
        assert torch.isfinite(self.loop.lat).all()
        return elapsed

    def close(self):
        from ief_amd.p2p.model.register import unregister_attention_control
        self.loop.release()
        unregister_attention_control(self.pipe, self.ctrl)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device(f"cuda:{local % max(1, torch.cuda.device_count())}")
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    from ief_amd import hip
    from ief_amd.p2p.model.sd_utils import _encode_prompts
    hip.load()

    pipe, cfg = build_pipe(args.config, dev, rank, world, precision=args.precision)
    from ief_amd import weights as _w
    nparams = _w.num_params(cfg)
    pipe.scheduler.set_timesteps(MAX_STEPS)
    hw = args.latent or cfg.sample_size
    with torch.no_grad():
        u, c = _encode_prompts(pipe, PROMPTS)
    ctx = torch.cat([u, c])
    # configs[1] (edit_real.py, null-text inversion): P2P_NTI swaps in one optimised unconditional embedding per step
    # (/root/reference/p2p/model/sd_utils.py:133-138); synthetic stand-ins of the right shape here
    uncond_list = None
    if args.uncond == "per-step":
        g = torch.Generator().manual_seed(1234 + rank)
        uncond_list = [(u[:1].cpu() + 0.01 * torch.randn(1, *u.shape[1:], generator=g)).to(dev) for _ in range(MAX_STEPS)]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.split == "cfg":
        if world != 2 or cfg.addition_embed:
            raise SystemExit("--split cfg runs one SD1.x / SD2.x edit on exactly two ranks (--gpus 2 under torchrun)")
        job = EditJob(pipe, cfg, ctx, hw, dev, rank, uncond_list, split_group=dist.group.WORLD)
        elapsed = job.timed(args.steps, args.warmup, barrier, dist, dev)
        if rank == 0:
            print(json.dumps({
                "metric": f"denoising steps/sec ({MODEL_NAMES.get(args.config, args.config)} {hw * 8}x{hw * 8} P2P edit step, ONE "
                          "edit CFG-split over 2 GPUs)",
                "value": round(args.steps / elapsed, 3), "unit": "steps/s", "n_gpus": 2, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": MODES[args.precision][1], "data": "synthetic",
                "config": {"workload": f"{args.config} UNet P2P AttentionRefine edit step, rank 0: unconditional rows (UNet batch 2), "
                                       f"rank 1: conditional rows + controller plan (UNet batch 2), one eps all-gather per step, "
                                       f"{hw}x{hw} latents"}}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return

    def measure(pipe_m, precision, steps, warmup, full):
        """the numbers of ONE mode: K edit steps between barriers (+ in-flight throughput, roofline, 1024x1024, PIE if `full`)"""
        peak, dtype, what = MODES[precision]
        # parity first, measured in this run (every rank: same work in front of the timed region; raises beyond the bound)
        probe = parity_probe(pipe_m, dev, precision) if (args.config, hw) == ("sd15", 64) and steps > 0 else None
        job = EditJob(pipe_m, cfg, ctx, hw, dev, rank, uncond_list)
        elapsed = job.timed(steps, warmup, barrier, dist, dev)
        value = world * steps / elapsed
        res = {"value": round(value, 3), "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps, "dtype": dtype,
               "arithmetic": what}
        if probe is not None:
            res["parity_probe"] = probe
            res["image_max_abs_err_50_step_edit_from_gputest"] = {
                "value": IMAGE_ERR_50_GPUTEST[precision], "meets_1e-3_image_bound": IMAGE_ERR_50_GPUTEST[precision] <= 1e-3,
                "source": "tests/test_gpu_zz_fullsize.py::test_sd15_edit50_vs_oracle_fixture (50 steps + VAE decode vs fixture G13), last "
                          "measured value; NOT measured by this run -- see parity_probe"}
        if full:
            # throughput schedule: E independent edits in flight per GPU (denoise.run_interleaved); `value` stays E = 1
            res["throughput_edits_in_flight"] = {
                str(E): round(edits_in_flight(pipe_m, cfg, ctx, job.x_T, hw, E, dev, dist, barrier, world, rank, uncond_list), 3)
                for E in [int(e) for e in args.in_flight.split(",") if e.strip()]}
        if rank == 0:
            register_job(pipe_m, job)
            res["roofline"] = roofline(job, value, world, args.config, precision)      # last use of this job: it ends eager
        job.close()
        if full and args.steps_1024 > 0 and hw != 128:       # the same edit step on 128x128 latents (1024x1024 px)
            job2 = EditJob(pipe_m, cfg, ctx, 128, dev, rank, uncond_list)
            el2 = job2.timed(args.steps_1024, 2, barrier, dist, dev)
            res["steps_per_sec_1024"] = round(world * args.steps_1024 / el2, 3)
            res["ms_per_step_1024"] = round(el2 / args.steps_1024 * 1e3, 3)
            if rank == 0:
                register_job(pipe_m, job2)
                res["roofline_1024"] = roofline(job2, res["steps_per_sec_1024"], world, args.config, precision)
            job2.close()
            del job2
            torch.cuda.empty_cache()
        # BASELINE.json's second metric: PIE-Bench images/sec (reference per-image order, then the batched schedule)
        if full and args.pie_images > 0 and not cfg.addition_embed:
            res.update(pie_images_per_sec(pipe_m, dev, rank, world, dist, barrier, args.pie_images))
        # BASELINE.json configs[1] (edit_real.py --inversion_type null-text): the optimisation's inner iteration and a whole image
        if full and args.nti_images > 0 and not cfg.addition_embed and hw == cfg.sample_size:
            res["null_text"] = null_text_numbers(pipe_m, cfg, dev, rank, world, dist, barrier, args.nti_images)
        return res, job

    # ---- headline mode
    head, job = measure(pipe, args.precision, args.steps, args.warmup, True)
    out = {
        "metric": f"denoising steps/sec ({MODEL_NAMES.get(args.config, args.config)} {hw * 8}x{hw * 8} P2P edit step, UNet batch 4)",
        "value": head["value"], "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": head["dtype"], "data": "synthetic",
        "config": {"workload": f"{args.config} UNet ({nparams / 1e6:.1f}M params) P2P AttentionRefine edit step, 2 prompts x CFG = batch 4, "
                               f"{hw}x{hw} latents ({hw * 8}x{hw * 8} px), 50-step DDIM, guidance 7.5, cross 0.8 / self 0.4, "
                               + ("per-step null-text unconditional embeddings (P2P_NTI, edit_real.py); "
                                  if args.uncond == "per-step" else "fixed unconditional embedding (edit_syn.py); ")
                               + "one independent edit per GPU; precision mode " + args.precision + ": " + head["arithmetic"]
                               + (" -- the mode that meets north_star's bound (edited images within 1e-3 max-abs of the fp32 "
                                  "reference: measured 4e-6 on a 50-step 512x512 edit)" if args.precision == "f16x3" else "")},
    }
    for k, v in head.items():
        if k not in ("value", "ms_per_step", "steps", "dtype"):
            out[k] = v

    # ---- the other modes beside it (N = 1: they are context for the headline, not scaled)
    others = [m.strip() for m in args.other_modes.split(",") if m.strip() and m.strip() != args.precision]
    if world == 1 and not cfg.addition_embed:
        sd_host = pipe._state_dict
        for m in others:
            pipe_m, _ = build_pipe(args.config, dev, rank, world, precision=m)
            pipe_m.scheduler.set_timesteps(MAX_STEPS)
            if m == "f32":
                r, j = measure(pipe_m, m, args.exact_steps, 1, False)
                r["peak_tflops_f32_mfma"] = PEAK_MFMA_F32 / 1e12
                r["achieved_tflops"] = round(r["value"] * FLOP_PER_STEP / 1e12, 2) if (args.config, hw) == ("sd15", 64) else None
                out["exact_mode"] = r
            else:
                r, j = measure(pipe_m, m, args.steps, args.warmup, True)
                out["fp16_mode" if m == "f16" else m + "_mode"] = r
            del pipe_m, j
            torch.cuda.empty_cache()
        pipe._state_dict = sd_host

    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pipe, cfg, ctx, job.x_T, job.ctrl, job.added, args.cpu_seconds)
            if "images_per_sec" in out:
                # one PIE image = 50 inversion forwards at batch 1 + 50 edit steps at batch 4 (+ VAE, ignored here):
                # 250 sample-forwards = 62.5 batch-4 steps of the CPU oracle
                out["cpu_baseline"]["images_per_sec_derived"] = round(out["cpu_baseline"]["value"] / 62.5, 6)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


MODEL_NAMES = {"sd15": "SD1.5", "sd21": "SD2.1", "sdxl": "SDXL", "tiny": "tiny", "small": "small", "small21": "small21",
               "smallxl": "smallxl"}


def register_job(pipe, job):
    from ief_amd.p2p.model.register import register_attention_control
    register_attention_control(pipe, job.ctrl)


def edits_in_flight(pipe, cfg, ctx, x_T, hw, E, dev, dist, barrier, world, rank, uncond_list=None, steps=40):
    """steps/s (all ranks) with E independent P2P edits stepped concurrently on each GPU"""
    from ief_amd.denoise import FusedDenoiser, run_interleaved
    from ief_amd.p2p.model.attention_control import AttentionRefine
    from ief_amd.p2p.model.register import register_attention_control, unregister_attention_control
    loops, ctrls = [], []
    added = make_added(cfg, hw, rank, dev)
    for _ in range(E):
        c = AttentionRefine(PROMPTS, pipe.tokenizer, MAX_STEPS, 0.8, 0.4, device=dev)
        register_attention_control(pipe, c)
        lp = FusedDenoiser(pipe, ctx, 2, (hw, hw), 7.5, uncond_list=uncond_list, added_cond_kwargs=added)
        lp.start(x_T)
        unregister_attention_control(pipe, None)
        loops.append(lp); ctrls.append(c)
    run_interleaved(loops, 4)
    for lp, c in zip(loops, ctrls):
        c.reset(); lp.rewind(x_T)
    barrier()
    t0 = time.perf_counter()
    run_interleaved(loops, steps)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        te = torch.tensor([dt], device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        dt = te.item()
    for lp in loops:
        lp.release()
    return world * E * steps / dt


def pie_images_per_sec(pipe, dev, rank, world, dist, barrier, n):
    """the PIE-Bench loop (`/root/reference/p2p/test.py:114-181`, our `p2p/test.py:run_items`) on n synthetic 512x512
    images per GPU: VAE encode -> 50-step DDIM inversion (UNet batch 1) -> 50-step P2P edit (batch 4) -> VAE decode ->
    uint8, nothing saved.  One untimed image first (captures / pools the two step graphs)."""
    import importlib.util
    import tempfile
    pdir = os.path.join(ROOT, "image-editing-framework_amd", "p2p")
    if pdir not in sys.path:
        sys.path.insert(0, pdir)
    spec = importlib.util.spec_from_file_location("ief_p2p_pie_driver", os.path.join(pdir, "test.py"))
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    from ief_amd.p2p.dataset.pie import SyntheticPIE
    from ief_amd.p2p.inversion.ddim import ddim_inversion
    from ief_amd.p2p.model.sd_utils import P2P
    size = pipe.unet.config.sample_size * pipe.vae_scale_factor
    root = tempfile.mkdtemp(prefix=f"ief_bench_pie_r{rank}_")
    items = list(SyntheticPIE(root, n + 4, size=size, seed=rank).items)
    editor, invertor = P2P(model=pipe, num_inference_steps=MAX_STEPS), ddim_inversion()
    res = {}
    for key, kw, warm in (("images_per_sec", dict(invert_batch=1, in_flight=1), 1),
                          ("images_per_sec_batched", dict(invert_batch=4, in_flight=4), 4)):
        drv.run_items(pipe, editor, invertor, items[:warm], size, dev, "ddim", **kw)
        barrier()
        t0 = time.perf_counter()
        drv.run_items(pipe, editor, invertor, items[-n:], size, dev, "ddim", **kw)
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            te = torch.tensor([dt], device=dev)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            dt = te.item()
        res[key] = round(world * n / dt, 4)
    res["images_config"] = (f"{n} synthetic {size}x{size} images per GPU, ddim inversion, 50 + 50 steps, guidance 7.5, cross 0.8 / "
                            "self 0.6, PNGs not written; images_per_sec = reference per-image order, images_per_sec_batched = "
                            "--invert_batch 4 --in_flight 4 (identical pixels)")
    return res


def null_text_numbers(pipe, cfg, dev, rank, world, dist, barrier, n):
    """config 2: (a) one inner iteration of `NTI.null_optimization` (`/root/reference/p2p/inversion/nti.py:15-33`: UNet forward at
    batch 1 keeping the adjoint's inputs, the objective, the activation-gradient pass, Adam) from its captured graph; (b) images/s of
    the PIE loop with `--inversion_type null-text` (VAE encode, 50-step DDIM inversion, up to 50 x 10 inner iterations with the
    reference's early stop, 50-step edit with the per-step embeddings, VAE decode) on n synthetic images per GPU, one at a time"""
    import importlib.util
    import tempfile
    from ief_amd.nti import NullTextOptimizer
    hw = cfg.sample_size
    g = torch.Generator().manual_seed(0)
    ctx = torch.randn(2, 77, cfg.cross_attention_dim, generator=g) * 0.1
    lats = [torch.randn(1, 4, hw, hw, generator=g) for _ in range(51)]
    opt = NullTextOptimizer(pipe, ctx[1:], 7.5, (hw, hw), use_graph=True)
    opt.run(lats, ctx[:1], 1, 0.0, num_outer=1)          # capture + warm-up
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    iters = 20
    ev[0].record()
    for _ in range(iters):
        opt._run(1)
    ev[1].record()
    torch.cuda.synchronize()
    inner_ms = ev[0].elapsed_time(ev[1]) / iters
    opt.release()
    pdir = os.path.join(ROOT, "image-editing-framework_amd", "p2p")
    if pdir not in sys.path:
        sys.path.insert(0, pdir)
    spec = importlib.util.spec_from_file_location("ief_p2p_pie_driver_nti", os.path.join(pdir, "test.py"))
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    from ief_amd.p2p.dataset.pie import SyntheticPIE
    from ief_amd.p2p.inversion.nti import NTI
    from ief_amd.p2p.model.sd_utils import P2P_NTI
    size = pipe.unet.config.sample_size * pipe.vae_scale_factor
    root = tempfile.mkdtemp(prefix=f"ief_bench_nti_r{rank}_")
    items = list(SyntheticPIE(root, n + 1, size=size, seed=100 + rank).items)
    editor, invertor = P2P_NTI(model=pipe, num_inference_steps=MAX_STEPS), NTI()
    drv.run_items(pipe, editor, invertor, items[:1], size, dev, "null-text")          # untimed: captures / pools the graphs
    barrier()
    t0 = time.perf_counter()
    drv.run_items(pipe, editor, invertor, items[1:], size, dev, "null-text")
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        te = torch.tensor([dt], device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        dt = te.item()
    return {"nti_inner_ms": round(inner_ms, 3), "inner_iterations_per_s": round(1000.0 / inner_ms, 2),
            "images_per_sec_null_text": round(world * n / dt, 4), "seconds_per_image": round(dt / n, 2),
            "config": f"{n} synthetic {size}x{size} image(s) per GPU after one untimed image, edit_real.py --inversion_type null-text: 50 "
                      "timesteps x up to 10 inner Adam iterations (early stop as `/root/reference/p2p/inversion/nti.py:31-33`), UNet batch 1"}


def _family(kernel_name: str) -> str:
    if kernel_name.startswith("igemm_f16_kernel"):
        return ("igemm_f16_kernel<.., CONV=true> (3x3 implicit-GEMM convolution)" if kernel_name.rstrip(">").endswith("true")
                else "igemm_f16_kernel<.., CONV=false> (linear / 1x1)")
    if kernel_name.startswith("conv3x3_halo_kernel"):
        return "conv3x3_halo_kernel<..> (3x3 convolution, input tile resident in LDS across the taps)"
    if kernel_name.startswith("igemm_x3_kernel") or kernel_name.startswith("igemm_f32_kernel"):
        base = kernel_name.split("<")[0]
        inner = kernel_name.split("<", 1)[1] if "<" in kernel_name else ""
        if inner.startswith("true"):
            return base + "<.., conv> (3x3 implicit-GEMM convolution)"
        if inner.startswith("scores") or inner.startswith("apply"):
            return base + "<.., batched> (materialised attention products)"
        return base + "<.., linear> (linear / 1x1)"
    if kernel_name.startswith("igemm_x3p_kernel<false>"):
        return "igemm_x3p_kernel<.., linear> (linear / 1x1 on operand planes, LDS-DMA staged)"
    if kernel_name.startswith("igemm_x3p_kernel<true>") or kernel_name.startswith("conv3x3_halo_x3p_kernel"):
        return "conv3x3 on operand planes (conv3x3_halo_x3p_kernel + igemm_x3p_kernel<.., conv>)"
    if kernel_name.startswith("attn_flash_x3p"):
        return "attn_flash_x3p_kernel<..> (self-attention on operand planes)"
    if kernel_name.startswith("attn_flash"):
        return kernel_name.split("<")[0] + "<..> (self-attention, all head dims)"
    return kernel_name


def physical_cores():
    """distinct (physical id, core id) pairs of /proc/cpuinfo; falls back to the logical count"""
    try:
        pairs, phys, core = set(), None, None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        pairs.add((phys, core))
                    phys = core = None
        return len(pairs) or os.cpu_count()
    except OSError:
        return os.cpu_count()


PEAK_HBM = 8.0e12             # MI355X_MICROARCH.md: HBM3E spec (6.3e12 measured with a float4 copy)
LDS_FILL_TBS = 6.4            # MI355X_MICROARCH.md, row `ldsdma-fill`: LDS-DMA fill ~25 GB/s per CU, 6.4 TB/s per chip (default cache policy)


def roofline(job, steps_per_sec, world, config, precision="f16"):
    """dominant kernel's algorithmic FLOP/s from per-launch HIP-event timings of one eager step; `peak` is the matrix peak
    in ALGORITHMIC FLOP/s of the mode (f16x3: a third of the fp16 MFMA peak, three MFMAs per product)"""
    PEAK = MODES[precision][0]
    from ief_amd import hip
    loop, ctrl, x_T = job.loop, job.ctrl, job.x_T
    loop.release()                         # drops the captured graph: the rest of this job runs eagerly
    ctrl.reset()
    loop.use_graph = False
    loop.lat.copy_(x_T.expand_as(loop.lat)); loop.step.zero_()
    loop._step_body()                      # eager warm-up of the un-captured path
    torch.cuda.synchronize()
    # keep the GPU busy while the host enqueues the step: with ~430 short launches the host (10 us per launch from
    # Python) would otherwise starve the stream and every event pair would include the wait for its launch
    torch.cuda._sleep(int(1.2e8 * max(1, (job.hw // 64) ** 2 // 2)))
    hip.profile_begin()
    loop._step_body()
    rec = hip.profile_end(with_bytes=True)
    staged = {}                            # LDS-DMA bytes per family (the planes kernels report them per launch)
    for kname, b in hip.profile_staged().items():
        staged[_family(kname)] = staged.get(_family(kname), 0.0) + b
    agg, fam = {}, {}
    for name, flops, ms, nbytes in rec:
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1; a[1] += flops; a[2] += ms
        # price of this launch on the roofline: the larger of its MFMA time and its HBM time
        floor_ms = max(flops / PEAK, nbytes / PEAK_HBM) * 1e3
        f = fam.setdefault(_family(name), [0, 0.0, 0.0, 0.0, 0.0])
        f[0] += 1; f[1] += flops; f[2] += ms; f[3] += nbytes; f[4] += floor_ms
    total_ms = sum(a[2] for a in agg.values())
    alg_flop = sum(a[1] for a in agg.values())   # matmul / conv / attention FLOPs of one step, counted per launch
    # dominant kernel = the MFMA kernel template with the largest share of the step; its tile / ring-depth
    # instantiations (chosen per layer shape by the tuned plan table) are one kernel for this purpose
    # the committed rocprofv3 --kernel-trace summary of this command (profiles/): per template, launches and total duration
    # without the HIP-event pair's few us per launch, and the PMC traffic of its most frequent member -- only for the
    # configuration those profiles were taken on.  WHICH template is "dominant" is decided by the committed trace's total
    # durations where it covers this configuration (the live event pairs inflate families of many short launches); the live
    # timing decides otherwise.
    prof_all = {}
    for tag in ("r04", "r03"):
        try:
            with open(os.path.join(ROOT, "profiles", f"{tag}_roofline_inputs.json")) as f:
                prof_all = json.load(f).get(f"{config}|{job.hw}|{precision}", {})
        except (OSError, ValueError):
            prof_all = {}
        if prof_all:
            break
    mfma_fams = {k: v for k, v in fam.items() if v[1] > 0}
    by_trace = {k: prof_all[k]["total_us_rocprof"] for k in mfma_fams if k in prof_all and "total_us_rocprof" in prof_all[k]}
    if by_trace:
        name = max(by_trace, key=by_trace.get)
        dominant_by = "total duration in the committed rocprofv3 kernel trace"
    else:
        name = max(mfma_fams, key=lambda k: mfma_fams[k][2])
        dominant_by = "live HIP-event timing of this run"
    n, flops, ms, nbytes, floor_ms = fam[name]
    achieved = flops / (ms * 1e-3) / 1e12
    prof = prof_all.get(name, {})
    out = {
        "bound": "mfma" if flops / PEAK >= nbytes / PEAK_HBM else "hbm",
        "kernel": name, "launches_per_step": n,
        "avg_launch_ms": round(ms / n, 4), "alg_gflop_per_launch": round(flops / n / 1e9, 3),
        "alg_mbytes_per_launch": round(nbytes / n / 1e6, 3),
        "achieved": round(achieved, 2), "peak": round(PEAK / 1e12, 1), "unit": "TFLOP/s",
        "frac": round(achieved * 1e12 / PEAK, 4),
        "frac_of_fp16_mfma_peak": round(achieved * 1e12 * (3.0 if precision == "f16x3" else 1.0) / PEAK_MFMA_F16, 4) if precision != "f32" else None,
        "frac_of_fp16_mfma_peak_note": "matrix-pipe FLOP/s actually executed (f16x3: 3 MFMAs per algorithmic product) / 2500 TFLOP/s",
        "dominant_chosen_by": dominant_by,
        "peak_note": {"f16": "dense fp16 MFMA peak", "f32": "fp32-input MFMA peak",
                      "f16x3": "algorithmic FLOP/s: dense fp16 MFMA peak (2500) / 3 MFMAs per product; the matrix pipe itself runs at "
                               "3 x `achieved`"}[precision],
        "timing": "HIP events around every launch on the launch stream (includes ~3 us of event pair per launch)",
        # each launch priced at max(FLOP / MFMA peak, algorithmic bytes / HBM peak): what share of the measured time the
        # roofline accounts for (the square K = C projections are HBM-side on this measure)
        "roofline_frac": round(floor_ms / ms, 4),
        "achieved_hbm_gbs": round(nbytes / (ms * 1e-3) / 1e9, 1),
        "traffic": prof.get("hbm_bytes_per_launch"), "traffic_unit": "HBM bytes per launch (PMC, committed profile; null when "
                                                                      "no profile of this configuration is committed)",
        "traffic_source": prof.get("traffic_source"),
        "kernel_share_of_step": round(ms / total_ms, 3),
        "whole_step": {"alg_tflop_per_step": round(alg_flop / 1e12, 3),
                       "achieved": round(steps_per_sec / world * alg_flop / 1e12, 2),
                       "frac": round(steps_per_sec / world * alg_flop / PEAK, 4)},
        "per_kernel_ms": {k: round(v[2], 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][2])},
    }
    if staged.get(name):
        # the bound that actually fits the planes GEMMs: bytes moved L2 -> LDS by LDS-DMA / the template's time,
        # against the chip-level LDS-DMA fill rate of the microarchitecture guide (`ldsdma-fill`: ~25 GB/s per CU, 6.4 TB/s)
        out["lds_fill"] = {"staged_mbytes_per_launch": round(staged[name] / n / 1e6, 2), "achieved_tb_s": round(staged[name] / (ms * 1e-3) / 1e12, 2),
                           "per_cu_gb_s": round(staged[name] / (ms * 1e-3) / 1e9 / 256, 1), "reference_tb_s": LDS_FILL_TBS,
                           "frac_of_reference": round(staged[name] / (ms * 1e-3) / 1e12 / LDS_FILL_TBS, 3),
                           "note": "bytes staged into LDS (two fp16 planes of both operands, every K tile of every output tile) / the template's "
                                   "live time; reference = the guide's chip-level LDS-DMA fill rate"}
    if "avg_launch_us_rocprof" in prof:
        us = prof["avg_launch_us_rocprof"]
        out["avg_launch_ms_rocprof"] = round(us * 1e-3, 4)
        out["frac_rocprof"] = round(flops / n / (us * 1e-6) / PEAK, 4)
        if "lds_fill" in out:
            out["lds_fill"]["frac_of_reference_rocprof"] = round(staged[name] / n / (us * 1e-6) / 1e12 / LDS_FILL_TBS, 3)
        out["rocprof_source"] = prof.get("source")
    if job.hw == 64 and config == "sd15":
        out["whole_step"]["survey_tflop_per_step_512px"] = FLOP_PER_STEP / 1e12
    return out


def cpu_baseline(pipe, cfg, ctx, x_T, ctrl, added, budget_s):
    """the oracle (reference semantics, fp32 eager) on this host: bounded sample of the same workload"""
    from oracle import p2p_ref, unet_ref
    sd = pipe._state_dict
    threads, cores = torch.get_num_threads(), physical_cores()
    ref_ctrl = p2p_ref.P2PControlRef(mode="refine", num_prompts=2, cross_alpha=ctrl.cross_replace_alpha.float().cpu(),
                                     num_self_replace=ctrl.num_self_replace, mapper=ctrl.mapper.cpu(),
                                     alphas=ctrl.alphas.float().cpu())
    ref_ctrl.num_att_layers = unet_ref.count_attention_layers(cfg)
    sched = p2p_ref.DDIMRef(MAX_STEPS)
    c = ctx.float().cpu()
    lat = x_T.cpu().expand(2, -1, -1, -1).clone()

    def step(i, lat):
        t = sched.timesteps[i]
        with torch.no_grad():
            eps = unet_ref.unet_forward(sd, cfg, torch.cat([lat] * 2), t, c, hook=ref_ctrl,
                                        added_cond_kwargs=None if added is None else {k: v.cpu() for k, v in added.items()})
        e_u, e_c = eps.chunk(2)
        return sched.step(e_u + 7.5 * (e_c - e_u), int(t), lat)

    lat = step(0, lat)  # warm-up (untimed)
    n, t0 = 0, time.perf_counter()
    while True:
        lat = step(1 + n, lat)
        n += 1
        dt = time.perf_counter() - t0
        if n >= 3 and (dt >= budget_s or n >= 8):      # at least 3 timed steps after the warm-up
            break
    return {"value": round(n / dt, 5), "unit": "steps/s", "cores": min(cores, threads), "physical_cores": cores, "torch_threads": threads,
            "kind": "port",
            "sample": f"{n} timed B=4 P2P edit steps of the fp32 eager oracle (materialised maps + Python controller) "
                      f"after 1 warm-up step, {dt:.1f} s, torch intra-op threads={threads} on {cores} physical cores"}


if __name__ == "__main__":
    main()
