"""profiles/r02_roofline_inputs.json from a rocprofv3 `--kernel-trace --stats` summary (kernel_stats.csv) of bench.py.

    python profiles/make_roofline_inputs.py <kernel_stats.csv> <config> <latent> [pmc.json]

For every kernel TEMPLATE bench.py's `roofline` can name (all tile / ring-depth instantiations of one template count as
one kernel) it records launches, total and average duration; `bench.py` then reports `frac_rocprof` = algorithmic FLOP
per launch / this average, next to its live HIP-event figure.  With a PMC file (profiles/r0x_pmc_traffic.json) the HBM
bytes per launch of the template's most frequent member ride along as `roofline.traffic`.
"""
import csv
import json
import os
import sys


def family(name: str) -> str:
    name = name.replace("void ", "").split("(")[0]
    if name.startswith("igemm_f16_kernel"):
        return ("igemm_f16_kernel<.., CONV=true> (3x3 implicit-GEMM convolution)" if name.rstrip(">").endswith("true")
                else "igemm_f16_kernel<.., CONV=false> (linear / 1x1)")
    if name.startswith("conv3x3_halo_kernel"):
        return "conv3x3_halo_kernel<..> (3x3 convolution, input tile resident in LDS across the taps)"
    if name.startswith("attn_flash"):
        return name.split("<")[0] + "<..> (self-attention, all head dims)"
    return name


def main():
    path, config, latent = sys.argv[1], sys.argv[2], int(sys.argv[3])
    pmc = json.load(open(sys.argv[4])) if len(sys.argv) > 4 else {}
    fam = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            k = family(row["Name"])
            a = fam.setdefault(k, [0, 0])
            a[0] += int(row["Calls"])
            a[1] += int(row["TotalDurationNs"])
    here = os.path.dirname(os.path.abspath(__file__))
    out_path = os.path.join(here, "r02_roofline_inputs.json")
    out = json.load(open(out_path)) if os.path.exists(out_path) else {}
    entry = {}
    for k, (calls, ns) in fam.items():
        e = {"launches_in_trace": calls, "avg_launch_us_rocprof": round(ns / calls / 1e3, 3), "source": os.path.basename(path)}
        hit = [v for kk, v in pmc.items() if not kk.startswith("_") and v.get("family") == k]
        if hit:
            e["hbm_bytes_per_launch"] = hit[0]["hbm_bytes"]
            e["traffic_source"] = f"{os.path.basename(sys.argv[4])}: {hit[0]['shape']}"
        entry[k] = e
    out[f"{config}|{latent}"] = entry
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
    print(out_path, len(entry), "kernel templates")


if __name__ == "__main__":
    main()
