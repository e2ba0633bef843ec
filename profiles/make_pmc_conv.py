"""profiles/r02_pmc_conv_halo_vs_igemm.json from the SQ counter passes of tests/pmc_conv.sh (gpurun_out/pmc_conv).

    python profiles/make_pmc_conv.py gpurun_out/pmc_conv

Counter units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves,
SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs, SQ_INSTS_* count wave-instructions.
"""
import csv, glob, json, os, sys

root = sys.argv[1]
out = {"_how": __doc__.strip(), "shape": "conv3x3 B=4 64x64 320->320 (M=16384 N=320 K=2880), 30.2 GFLOP, one launch"}
for case, needle in (("conv64igemm", "igemm_f16_kernel"), ("conv64halo", "conv3x3_halo_kernel")):
    c = {}
    for part in ("a", "b"):
        for f in glob.glob(os.path.join(root, f"{case}_{part}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if needle in r["Kernel_Name"]:
                    c.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                    name = r["Kernel_Name"].replace("void ", "").split("(")[0]
                    waves = int(r["Grid_Size"]) // 64 if "Grid_Size" in r else None
    c = {k: sum(v) / len(v) for k, v in c.items()}
    dur = None
    for f in glob.glob(os.path.join(root, f"{case}_t", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if needle in r["Name"]:
                dur = float(r["AverageNs"]) / 1e3
    e = {"kernel": name, "avg_launch_us": dur, "counters": {k: int(v) for k, v in sorted(c.items())}}
    if dur:
        e["tflops"] = round(30.2e9 / (dur * 1e-6) / 1e12, 1)
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        e["share_of_wave_cycles"] = {k: round(c[k] / wc, 3) for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") if k in c}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
        e["mfma_busy_cycles_per_simd"] = int(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024)
    if waves and "SQ_INSTS_SALU" in c:
        e["salu_instructions_per_wave"] = int(c["SQ_INSTS_SALU"] / waves)
        e["valu_instructions_per_wave"] = int(c["SQ_INSTS_VALU"] / waves)
    out[case] = e
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "r02_pmc_conv_halo_vs_igemm.json"), "w"), indent=1)
for k, v in out.items():
    if not k.startswith("_"):
        print(k, v if isinstance(v, str) else {a: b for a, b in v.items() if a != "counters"})
