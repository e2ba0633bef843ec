"""End-to-end CLI runs on the GPU: the mirrored `edit_syn.py`, `edit_real.py --inversion_type ddim` and the
sharded `test.py` driver (TINY shape family so they take seconds), plus bench.py under torch.distributed.run."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
from PIL import Image

from _procs import call_main, run_mixed

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P2P = os.path.join(ROOT, "image-editing-framework_amd", "p2p")


def run(args, cwd, timeout=600, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable] + args, cwd=cwd, capture_output=True, text=True, timeout=timeout, env=e)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_edit_syn_cli(tmp_path):
    run([os.path.join(P2P, "edit_syn.py"), "--sd_version", "tiny", "--seed", "8888"], cwd=str(tmp_path))
    src = np.array(Image.open(tmp_path / "exp" / "source.png"))
    edit = np.array(Image.open(tmp_path / "exp" / "edit.png"))
    assert src.shape == edit.shape == (128, 128, 3) and src.dtype == np.uint8
    assert src.std() > 1 and (src.astype(int) - edit.astype(int)).__abs__().max() > 0


def test_masactrl_edit_syn_cli(tmp_path):
    masa = os.path.join(ROOT, "image-editing-framework_amd", "masactrl")
    run([os.path.join(masa, "edit_syn.py"), "--sd_version", "tiny"], cwd=str(tmp_path))
    src = np.array(Image.open(tmp_path / "exp" / "source.png"))
    edit = np.array(Image.open(tmp_path / "exp" / "edit.png"))
    assert src.shape == edit.shape == (128, 128, 3) and (src.astype(int) - edit.astype(int)).__abs__().max() > 0


def test_edit_real_cli_ddim(tmp_path):
    rng = np.random.RandomState(0)
    img = np.kron(rng.randint(0, 255, (8, 8, 3)), np.ones((16, 16, 1))).astype(np.uint8)
    Image.fromarray(img).save(tmp_path / "test.jpg")
    run([os.path.join(P2P, "edit_real.py"), "--sd_version", "tiny", "--inversion_type", "ddim", "--source_image",
         str(tmp_path / "test.jpg")], cwd=str(tmp_path))
    for name in ("source.png", "inversion.png", "edit.png"):
        assert (tmp_path / "exp" / name).exists()


def test_edit_real_cli_null_text(tmp_path):
    """default inversion type of the reference (`edit_real.py:26`): null-text optimisation on the HIP adjoint kernels"""
    rng = np.random.RandomState(0)
    img = np.kron(rng.randint(0, 255, (8, 8, 3)), np.ones((16, 16, 1))).astype(np.uint8)
    Image.fromarray(img).save(tmp_path / "test.jpg")
    run([os.path.join(P2P, "edit_real.py"), "--sd_version", "tiny", "--source_image", str(tmp_path / "test.jpg")],
        cwd=str(tmp_path))
    for name in ("source.png", "inversion.png", "edit.png"):
        assert (tmp_path / "exp" / name).exists()
    src = np.array(Image.open(tmp_path / "exp" / "source.png")).astype(int)
    inv = np.array(Image.open(tmp_path / "exp" / "inversion.png")).astype(int)
    assert src.shape == inv.shape


def test_pie_driver_synthetic(tmp_path):
    out = run([os.path.join(P2P, "test.py"), "--sd_version", "tiny", "--synthetic", "3", "--exp_path",
               str(tmp_path / "test_exp")], cwd=str(tmp_path))
    rec = json.loads(out.strip().splitlines()[-1])
    assert rec["images"] == 3 and rec["images_per_sec"] > 0
    done = [d for d in os.listdir(tmp_path / "test_exp") if d.startswith("syn_")]
    assert len(done) == 3
    for d in done:
        assert sorted(os.listdir(tmp_path / "test_exp" / d)) == ["edit.png", "inversion.png", "source.png"]


def test_bench_under_torchrun_one_rank(tmp_path):
    """the exact launch line the driver uses for N > 1, with N = 1 (one GPU on this box)"""
    out = run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
               "--config", "small", "--no-cpu-baseline", "--pie-images", "2", "--steps-1024", "2", "--other-modes", "f16", "--in-flight", "2",
               "--nti-images", "0"], cwd=ROOT, timeout=900)
    rec = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["steps"] == 6 and rec["value"] > 0 and "roofline" in rec
    # BASELINE.json's second metric and north_star's second latent size ride on the same line
    assert rec["images_per_sec"] > 0 and rec["images_per_sec_batched"] > 0
    assert rec["steps_per_sec_1024"] > 0 and rec["roofline_1024"]["frac"] > 0
    assert rec["roofline"]["bound"] in ("mfma", "hbm") and 0 < rec["roofline"]["roofline_frac"] <= 1.5


def test_cfg_split_two_ranks_one_edit(tmp_path):
    """ONE edit on two ranks (unconditional rows | conditional rows + controller plan), one eps exchange per
    step.  Two processes share this box's GPU, so the exchange runs over gloo; the HIP path, the captured half-step graphs
    and the cond-only plan are the ones two GPUs would run over RCCL.  The split latents must equal the full-batch edit
    (different UNet batch => different tile plans, so to fp16 rounding, not bit for bit) and be identical on both ranks."""
    out = run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29633", os.path.join(ROOT, "tests", "workers", "cfg_split_worker.py")], cwd=ROOT, timeout=600)
    rec = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    print(rec)
    assert rec["finite"] and rec["ranks_identical"] and rec["cur_step"] == 6
    assert rec["rel_split_vs_full"] < 5e-3


def test_bench_two_ranks_weak_and_cfg_split(tmp_path):
    """`bench.py --gpus 2` under the driver's launch line, both ranks on this box's one GPU over gloo (RCCL refuses two ranks
    per device): the weight broadcast, the barriers / max-over-ranks timing of the weak-scaling line, and `--split cfg`
    (one edit on two ranks, strong scaling)"""
    base = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1"]
    tail = ["--gpus", "2", "--steps", "4", "--warmup", "1", "--config", "tiny", "--no-cpu-baseline", "--dist-backend", "gloo",
            "--pie-images", "1", "--steps-1024", "0", "--in-flight", "", "--exact-steps", "0"]
    out = run(base + ["--master-port", "29641", os.path.join(ROOT, "bench.py")] + tail, cwd=ROOT, timeout=600)
    rec = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0 and rec["images_per_sec"] > 0
    out = run(base + ["--master-port", "29643", os.path.join(ROOT, "bench.py")] + tail + ["--split", "cfg"], cwd=ROOT, timeout=600)
    rec = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["value"] > 0


def test_pie_driver_batched_inversion_matches_per_image(tmp_path):
    """--invert_batch K inverts K images in one batched DDIM loop and --in_flight E steps E edits concurrently; images
    are independent, so the PNGs must match the per-image run."""
    a, b, c = tmp_path / "a", tmp_path / "b", tmp_path / "c"
    base = [os.path.join(P2P, "test.py"), "--sd_version", "tiny", "--synthetic", "3"]
    script, base = base[0], base[1:]
    done = [call_main(script, base + ["--exp_path", str(a)], tmp_path / "cwd_a"),
            call_main(script, base + ["--invert_batch", "2", "--exp_path", str(b)], tmp_path / "cwd_b"),
            call_main(script, base + ["--invert_batch", "3", "--in_flight", "2", "--exp_path", str(c)], tmp_path / "cwd_c")]
    for d in done:
        assert d.last_json()["images"] == 3
    for other in (b, c):
        for d in sorted(x for x in os.listdir(a) if x.startswith("syn_")):
            for name in ("inversion.png", "edit.png"):
                pa = np.array(Image.open(a / d / name)).astype(int)
                pb = np.array(Image.open(other / d / name)).astype(int)
                assert pa.shape == pb.shape
                # every kernel sums in a fixed order and the schedules only regroup independent images: same pixels
                assert np.abs(pa - pb).max() <= 1, (other.name, d, name, np.abs(pa - pb).max())


def test_masactrl_edit_real_and_pie_driver(tmp_path):
    masa = os.path.join(ROOT, "image-editing-framework_amd", "masactrl")
    rng = np.random.RandomState(0)
    img = np.kron(rng.randint(0, 255, (8, 8, 3)), np.ones((16, 16, 1))).astype(np.uint8)
    Image.fromarray(img).save(tmp_path / "test.jpg")
    jobs = [([os.path.join(masa, "edit_real.py"), "--sd_version", "tiny", "--inversion_type", inv, "--source_image",
              str(tmp_path / "test.jpg")], tmp_path / inv) for inv in ("ddim", "null-text")]
    jobs.append(([os.path.join(masa, "test.py"), "--sd_version", "tiny", "--synthetic", "2", "--exp_path", str(tmp_path / "t")],
                 tmp_path / "cwd_t"))
    done = run_mixed(jobs, in_process=(1,))
    for inv in ("ddim", "null-text"):
        for name in ("source.png", "inversion.png", "edit.png"):
            assert (tmp_path / inv / "exp" / name).exists()
    rec = done[2].last_json()
    assert rec["images"] == 2 and rec["images_per_sec"] > 0


def test_pie_driver_two_ranks_weight_broadcast_and_shards(tmp_path):
    """the PIE driver itself on two ranks (torchrun, both on this box's one GPU over gloo: RCCL refuses two ranks per device):
    rank 0 draws the weights, rank 1 builds its pipeline from zeros and receives the packed UNet + VAE tensors by the
    driver's broadcast (`_bootstrap.load_pipe` -> `dist.broadcast_pipeline`); the shards are disjoint and complete, and the
    PNGs equal a one-rank run's -- which they can only do if rank 1 got rank 0's weights."""
    a, b = tmp_path / "one", tmp_path / "two"
    run([os.path.join(P2P, "test.py"), "--sd_version", "tiny", "--synthetic", "4", "--exp_path", str(a)], cwd=str(tmp_path))
    out = run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29655", os.path.join(P2P, "test.py"), "--sd_version", "tiny", "--synthetic", "4",
               "--exp_path", str(b)], cwd=str(tmp_path), env={"IEF_DIST_BACKEND": "gloo"}, timeout=900)
    rec = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert rec["images"] == 4 and rec["n_gpus"] == 2
    dirs = sorted(x for x in os.listdir(a) if x.startswith("syn_"))
    assert len(dirs) == 4 and sorted(x for x in os.listdir(b) if x.startswith("syn_")) == dirs
    for d in dirs:
        for name in ("inversion.png", "edit.png"):
            pa, pb = np.array(Image.open(a / d / name)).astype(int), np.array(Image.open(b / d / name)).astype(int)
            assert pa.shape == pb.shape and np.abs(pa - pb).max() == 0, (d, name)


def test_cfg_split_two_ranks_one_edit_rccl(tmp_path):
    """the same split with each rank on its OWN GPU and the per-step eps all-gather over RCCL (`--backend nccl`): the leg the
    one-GPU box cannot run.  Skipped unless two GPUs are visible (the round-end 8-GPU node; never on the builder's box)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: the RCCL leg of the CFG split (its gloo twin above runs on one)")
    out = run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29641", os.path.join(ROOT, "tests", "workers", "cfg_split_worker.py"), "--backend", "nccl"],
              cwd=ROOT, timeout=600)
    rec = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    print(rec)
    assert rec["finite"] and rec["ranks_identical"] and rec["cur_step"] == 6
    assert rec["rel_split_vs_full"] < 5e-3
