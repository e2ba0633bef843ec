"""EXPERIMENT: one B=4 UNet forward vs two concurrent B=2 forwards on two streams inside one hipGraph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ief_amd
from ief_amd.pipeline import StableDiffusionPipeline
dev = torch.device("cuda:0")
pipe = StableDiffusionPipeline.from_pretrained("synthetic:sd15")
unet = pipe.unet
g = torch.Generator().manual_seed(0)
x4 = torch.randn(4, 4, 64, 64, generator=g).to(dev)
ctx4 = (torch.randn(4, 77, 768, generator=g) * 0.1).half().to(dev)
temb = unet.time_rows(torch.tensor([501.0], device=dev))
xa, xb = x4[:2].contiguous(), x4[2:].contiguous()
ca, cb = ctx4[:2].contiguous(), ctx4[2:].contiguous()
for m in unet.attention_modules():
    m.cache_kv = False

def one():
    return unet(x4, encoder_hidden_states=ctx4, temb_row=temb)["sample"]

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        a = unet(xa, encoder_hidden_states=ca, temb_row=temb)["sample"]
    with torch.cuda.stream(s2):
        b = unet(xb, encoder_hidden_states=cb, temb_row=temb)["sample"]
    cur.wait_stream(s1); cur.wait_stream(s2)
    return a, b

def two_seq():
    a = unet(xa, encoder_hidden_states=ca, temb_row=temb)["sample"]
    b = unet(xb, encoder_hidden_states=cb, temb_row=temb)["sample"]
    return a, b

def bench(fn, name):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = fn()
    for _ in range(3): gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): gr.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 20:.3f} ms", flush=True)
    return out

o4 = bench(one, "one B=4 forward")
oa, ob = bench(two_seq, "two B=2 forwards, one stream")
pa, pb = bench(two, "two B=2 forwards, two streams")
print("match", (torch.cat([pa, pb]) - o4).abs().max().item(), (torch.cat([oa, ob]) - o4).abs().max().item())


# E INDEPENDENT chains (E images' steps) as E graphs replayed on E streams
def make(B, seed):
    gg = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4, 64, 64, generator=gg).to(dev)
    c = (torch.randn(B, 77, 768, generator=gg) * 0.1).half().to(dev)
    fn = lambda: unet(x, encoder_hidden_states=c, temb_row=temb)["sample"]
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    return gr
def run(graphs, streams, rounds=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(rounds):
        for gr, st in zip(graphs, streams):
            with torch.cuda.stream(st):
                gr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / rounds * 1e3
streams = [torch.cuda.Stream() for _ in range(4)]
g4 = [make(4, i) for i in range(4)]
for E in (1, 2, 3, 4):
    dt = run(g4[:E], streams[:E])
    print(f"{E} concurrent B=4 chains: {dt:.3f} ms per round = {dt / E:.3f} ms per edit step", flush=True)
g1 = [make(1, 10 + i) for i in range(4)]
for E in (1, 2, 4):
    dt = run(g1[:E], streams[:E])
    print(f"{E} concurrent B=1 chains: {dt:.3f} ms per round = {dt / E:.3f} ms per inversion step", flush=True)
dt = run([g4[0], g1[0]], streams[:2])
print(f"B=4 edit chain + B=1 inversion chain together: {dt:.3f} ms per round", flush=True)
dt = run([g4[0], g4[1], g1[0]], streams[:3])
print(f"2 x B=4 + B=1 together: {dt:.3f} ms per round", flush=True)
