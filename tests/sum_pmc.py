import csv, glob, os, sys
root, case, needle = sys.argv[1], sys.argv[2], sys.argv[3]
c = {}; waves=None
for part in ("a", "b"):
    for f in glob.glob(os.path.join(root, f"{case}_{part}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if needle in r["Kernel_Name"]:
                c.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                waves = int(r["Grid_Size"]) // 64
c = {k: sum(v) / len(v) for k, v in c.items()}
dur=None
for f in glob.glob(os.path.join(root, f"{case}_t", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if needle in r["Name"]: dur = float(r["AverageNs"]) / 1e3
print(case, "avg us", dur, "waves", waves)
wc=c.get("SQ_WAVE_CYCLES",1)
for k,v in sorted(c.items()):
    print(f"  {k:28s} {v:14.0f}  /wave {v/waves:10.1f}  share_of_wave_cycles {v/wc:6.3f}")
