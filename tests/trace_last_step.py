"""per-kernel totals of the LAST denoising step in a rocprofv3 kernel trace (the launches between the last two cfg_ddim kernels)
    python tests/trace_last_step.py <..._kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "cfg_ddim" in r["Kernel_Name"]]
seq = rows[idx[-2] + 1: idx[-1] + 1]
agg = collections.OrderedDict()
for r in seq:
    n = r["Kernel_Name"].replace("void ", "").split("(")[0]
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in agg.values())
wall = (int(seq[-1]["End_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / 1e3
print(f"{len(seq)} launches, {tot / 1e3:.3f} ms of kernel time, {wall / 1e3:.3f} ms first start -> last end")
for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{us / 1e3:8.3f} ms  x{c:4d}  {us / c:8.1f} us  {n[:100]}")
