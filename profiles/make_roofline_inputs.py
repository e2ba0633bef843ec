"""profiles/<tag>_roofline_inputs.json from a rocprofv3 `--kernel-trace --stats` summary (kernel_stats.csv) of bench.py.

    python profiles/make_roofline_inputs.py <kernel_stats.csv> <config> <latent> [pmc.json] [--precision f16x3|f16|f32] [--tag r03]

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
    if name.startswith("igemm_x3_kernel") or name.startswith("igemm_f32_kernel"):
        # template arguments: x3 <WM, WN, TM, TN, KIND, TRANSB, BPRE> (KIND >= 2: convolution), f32 <CONV, TRANSB, NT>; the
        # batched attention products (TRANSB or KIND 1 with fp32 B) are what bench.py names "<.., batched>"
        base = name.split("<")[0]
        args = [a.strip() for a in name.split("<", 1)[1].rstrip(">").split(",")] if "<" in name else []
        if base == "igemm_x3_kernel" and len(args) >= 7:
            if int(args[4]) >= 2:
                return base + "<.., conv> (3x3 implicit-GEMM convolution)"
            if args[5] == "true" or args[6] == "false":
                return base + "<.., batched> (materialised attention products)"
            return base + "<.., linear> (linear / 1x1)"
        if base == "igemm_f32_kernel" and args:
            if args[0] == "true":
                return base + "<.., conv> (3x3 implicit-GEMM convolution)"
            return base + "<.., linear> (linear / 1x1)"
    if name.startswith("igemm_x3p_kernel"):
        if name.rstrip(">").endswith("true"):
            return "conv3x3 on operand planes (conv3x3_halo_x3p_kernel + igemm_x3p_kernel<.., conv>)"
        return "igemm_x3p_kernel<.., linear> (linear / 1x1 on operand planes, LDS-DMA staged)"
    if name.startswith("conv3x3_halo_x3p_kernel"):
        return "conv3x3 on operand planes (conv3x3_halo_x3p_kernel + igemm_x3p_kernel<.., conv>)"
    if name.startswith("attn_flash_x3p"):
        return "attn_flash_x3p_kernel<..> (self-attention on operand planes)"
    if name.startswith("attn_flash"):
        return name.split("<")[0] + "<..> (self-attention, all head dims)"
    return name


def main():
    argv = list(sys.argv)
    precision, tag = None, "r02"
    for flag in ("--precision", "--tag"):
        if flag in argv:
            i = argv.index(flag)
            val = argv[i + 1]
            del argv[i:i + 2]
            if flag == "--precision":
                precision = val
            else:
                tag = val
    path, config, latent = argv[1], argv[2], int(argv[3])
    pmc = json.load(open(argv[4])) if len(argv) > 4 else {}
    fam = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            k = family(row["Name"])
            a = fam.setdefault(k, [0, 0])
            a[0] += int(row["Calls"])
            a[1] += int(row["TotalDurationNs"])
    here = os.path.dirname(os.path.abspath(__file__))
    out_path = os.path.join(here, f"{tag}_roofline_inputs.json")
    out = json.load(open(out_path)) if os.path.exists(out_path) else {}
    entry = {}
    for k, (calls, ns) in fam.items():
        e = {"launches_in_trace": calls, "avg_launch_us_rocprof": round(ns / calls / 1e3, 3), "total_us_rocprof": round(ns / 1e3, 1),
             "source": os.path.basename(path)}
        hit = [v for kk, v in pmc.items() if not kk.startswith("_") and v.get("family") == k]
        if hit:
            e["hbm_bytes_per_launch"] = hit[0]["hbm_bytes"]
            e["traffic_source"] = f"{os.path.basename(argv[4])}: {hit[0]['shape']}"
        entry[k] = e
    out[f"{config}|{latent}" + (f"|{precision}" if precision else "")] = entry
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
    print(out_path, len(entry), "kernel templates")


if __name__ == "__main__":
    main()
