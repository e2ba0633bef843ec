"""per-kernel time of ONE eager edit step in an fp32-storage mode (HIP events per launch):
    python tests/prof_exact.py [sd15] [f32 | f16x3]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ief_amd import hip
from ief_amd.p2p.model.sd_utils import _encode_prompts
cfgname = sys.argv[1] if len(sys.argv) > 1 else "sd15"
precision = sys.argv[2] if len(sys.argv) > 2 else "f32"
dev = torch.device("cuda:0")
pipe, cfg = bench.build_pipe(cfgname, dev, 0, 1, precision=precision)
pipe.scheduler.set_timesteps(50)
with torch.no_grad():
    u, c = _encode_prompts(pipe, bench.PROMPTS)
job = bench.EditJob(pipe, cfg, torch.cat([u, c]), cfg.sample_size, dev, 0, None)
job.loop.release(); job.ctrl.reset(); job.loop.use_graph = False; job.loop.rewind(job.x_T)
job.loop._step_body(); torch.cuda.synchronize()
torch.cuda._sleep(int(2e8))
hip.profile_begin(); job.loop._step_body(); rec = hip.profile_end(with_bytes=True)
agg = collections.OrderedDict()
for name, fl, ms, nb in rec:
    a = agg.setdefault(name, [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += fl; a[2] += ms; a[3] += nb
tot = sum(a[2] for a in agg.values())
for k, (n, fl, ms, nb) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print(f"{k:44s} n={n:4d} {ms:8.3f} ms  {fl / ms / 1e9 if ms else 0:8.1f} TFLOP/s  {nb / ms / 1e6 if ms else 0:8.1f} GB/s")
print("timed kernels total", round(tot, 2), "ms (untimed: elementwise, layernorm, softmax-in-probs...)")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); job.loop._step_body(); e1.record(); torch.cuda.synchronize()
print("whole eager step", round(e0.elapsed_time(e1), 2), "ms")
