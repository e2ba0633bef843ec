"""ONE captured step cut out of a rocprofv3 kernel trace of the headline bench command (between two `cfg_ddim_kernel` launches of the
timed replay loop): launches and kernel time per family.

    python profiles/make_step_breakdown.py <kernel_trace.csv> <out.json>
"""
import csv
import json
import sys


def family(n):
    if n.startswith("igemm_x3p_kernel"):
        return "conv (implicit GEMM on planes)" if n.split("(")[0].rstrip(">").endswith("true") else "linear (igemm_x3p)"
    if n.startswith("conv3x3_halo_x3p"):
        return "conv (halo on planes)"
    if n.startswith("attn_flash"):
        return "self-attention (flash)"
    if n.startswith("attn_cross_p2p"):
        return "cross-attention (fused edit)"
    if "gn3_" in n or "groupnorm" in n:
        return "GroupNorm"
    if "layernorm" in n:
        return "LayerNorm"
    if "reduce" in n:
        return "split-K reducer"
    return "other"


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "cfg_ddim" in r["Kernel_Name"]]
    # the timed replays are the longest run of equally spaced marks: take a step from the middle of the trace
    a, b = marks[len(marks) // 2] + 1, marks[len(marks) // 2 + 1] + 1
    step = rows[a:b]
    out = {"_what": "one captured step of the headline command (f16x3, SD1.5 512x512, batch 4) from profiles' kernel trace: launches and "
                    "kernel time per family", "launches": len(step)}
    fam = {}
    tot = 0.0
    for r in step:
        n = r["Kernel_Name"].replace("void ", "")
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        tot += d
        f = fam.setdefault(family(n), {"launches": 0, "ms": 0.0})
        f["launches"] += 1
        f["ms"] += d
    out["kernel_ms"] = round(tot, 3)
    out["wall_ms"] = round((int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e6, 3)
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
        out[k] = {"launches": v["launches"], "ms": round(v["ms"], 3)}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
