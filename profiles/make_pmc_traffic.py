"""profiles/<tag>_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tests/profile_round.sh.

    python profiles/make_pmc_traffic.py gpurun_out/prof_r03 [r03]

Per kernel of interest (tests/one_kernel.py launches it 4 times): the mean counter value of its dispatches, in KB as rocprofv3
reports them.  HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: on gfx950 FETCH_SIZE tallies the 128-byte requests of wide
coalesced reads at 64 bytes (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-byte streaming stores.
"""
import csv
import glob
import json
import os
import sys

CASES = {
    "gemmsq": ("igemm_f16_kernel", "igemm_f16_kernel<.., CONV=false> (linear / 1x1)",
               "square projection with bias + residual, B=4 64x64 tokens, 320->320 (M=16384 N=320 K=320): the most frequent linear "
               "of the step (to_out / proj_out / to_q at the 64x64 level)", 2 * (16384 * 320 * 3 + 320 * 320)),
    "conv64": ("conv3x3_halo_kernel", "conv3x3_halo_kernel<..> (3x3 convolution, input tile resident in LDS across the taps)",
               "conv3x3 B=4 64x64 320->320 (M=16384 N=320 K=2880): the most frequent large convolution of the step",
               2 * (16384 * 320 * 2 + 320 * 2880)),
    "conv32": ("conv3x3_halo_kernel", "-", "conv3x3 B=4 32x32 640->640 (M=4096 N=640 K=5760), split-K: the kernel alone; the reducer "
               "launch is listed beside it", 2 * (4096 * 640 * 2 + 640 * 5760)),
    "conv64igemm": ("igemm_f16_kernel", "igemm_f16_kernel<.., CONV=true> (3x3 implicit-GEMM convolution)",
                    "the same convolution on the implicit-GEMM kernel (tile 7), for comparison with the halo kernel",
                    2 * (16384 * 320 * 2 + 320 * 2880)),
    "attn40": ("attn_flash", "attn_flash_sp_kernel<..> (self-attention, all head dims)",
               "B=4 heads=8 N=L=4096 d=40 (SD1.5 64x64 self-attention)", 2 * 4 * 8 * 4096 * 40 * 4),
    # split-operand mode (fp32 activations; weights read as pre-split fp16 planes = 4 bytes per weight as well)
    "conv64x3": ("igemm_x3_kernel", "igemm_x3_kernel<.., conv> (3x3 implicit-GEMM convolution)",
                 "f16x3: conv3x3 B=4 64x64 320->320 (M=16384 N=320 K=2880), fp32 in / out", 4 * (16384 * 320 * 2 + 320 * 2880)),
    "conv32x3": ("igemm_x3_kernel", "-", "f16x3: conv3x3 B=4 32x32 640->640 (M=4096 N=640 K=5760), split-K 2: the kernel alone; "
                 "the reducer launch is listed beside it", 4 * (4096 * 640 * 2 + 640 * 5760)),
    "gemmsqx3": ("igemm_x3_kernel", "igemm_x3_kernel<.., linear> (linear / 1x1)",
                 "f16x3: square projection with bias + residual (M=16384 N=320 K=320), fp32 in / out",
                 4 * (16384 * 320 * 3 + 320 * 320)),
    "gemmffx3": ("igemm_x3_kernel", "-", "f16x3: FeedForward.net[0] shape without the fused GEGLU (M=4096 N=5120 K=640)",
                 4 * (4096 * 640 + 5120 * 640 + 4096 * 5120)),
    "attn40x3": ("attn_flash_x3", "attn_flash_x3_kernel<..> (self-attention, all head dims)",
                 "f16x3: B=4 heads=8 N=L=4096 d=40, fp32 q / k / v / out", 4 * 4 * 8 * 4096 * 40 * 4),
    # round 4: operand planes (4 bytes per element: hi + lo fp16 planes), both operands staged by LDS-DMA
    "conv64x3p": ("conv3x3_halo_x3p", "conv3x3 on operand planes (conv3x3_halo_x3p_kernel + igemm_x3p_kernel<.., conv>)",
                  "f16x3 planes: conv3x3 B=4 64x64 320->320 (M=16384 N=320 K=2880) on the halo kernel, planes in, fp32 out",
                  4 * (16384 * 320 * 2 + 320 * 2880)),
    "conv32x3p": ("conv3x3_halo_x3p", "-", "f16x3 planes: conv3x3 B=4 32x32 640->640 (M=4096 N=640 K=5760) as the plan table runs it "
                  "(halo kernel, split-K over channel blocks); the reducer launch is listed beside it", 4 * (4096 * 640 * 2 + 640 * 5760)),
    "gemmsqx3p": ("igemm_x3p_kernel", "igemm_x3p_kernel<.., linear> (linear / 1x1 on operand planes, LDS-DMA staged)",
                  "f16x3 planes: square projection with bias + residual (M=16384 N=320 K=320), planes in, fp32 residual and out",
                  4 * (16384 * 320 * 3 + 320 * 320)),
    "gemmffx3p": ("igemm_x3p_kernel", "-", "f16x3 planes: FeedForward.net[0] shape without the fused GEGLU (M=4096 N=5120 K=640)",
                  4 * (4096 * 640 + 5120 * 640 + 4096 * 5120)),
    "attn40x3p": ("attn_flash_x3p", "attn_flash_x3p_kernel<..> (self-attention on operand planes)",
                  "f16x3 planes: B=4 heads=8 N=L=4096 d=40, planes q / k / v in, planes out", 4 * 4 * 8 * 4096 * 40 * 4),
}


def mean_counter(d, needle):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                if needle in name or "splitk_epilogue" in name or "splitk_reduce" in name or "x3p_reduce" in name:
                    key = name.replace("void ", "").split("(")[0]
                    out.setdefault(key, []).append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}


def main():
    root = sys.argv[1]
    tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
    res = {"_how": __doc__.strip()}
    for case, (needle, family, shape, alg) in CASES.items():
        if not os.path.isdir(os.path.join(root, f"pmc_{case}_fetch")):
            continue
        fetch = mean_counter(os.path.join(root, f"pmc_{case}_fetch"), needle)
        write = mean_counter(os.path.join(root, f"pmc_{case}_write"), needle)
        for k in fetch:
            red = "splitk" in k or "x3p_reduce" in k
            e = {"family": family if not red else "-", "shape": shape if not red else shape + " — its slab reducer",
                 "fetch_size_kb": round(fetch[k], 1), "write_size_kb": round(write.get(k, 0.0), 1),
                 "hbm_bytes": int((2 * fetch[k] + write.get(k, 0.0)) * 1024)}
            if not red:
                e["algorithmic_bytes"] = alg
                e["hbm_over_algorithmic"] = round(e["hbm_bytes"] / alg, 2)
            res[f"{k} [{case}]"] = e
    here = os.path.dirname(os.path.abspath(__file__))
    json.dump(res, open(os.path.join(here, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    for k, v in res.items():
        if not k.startswith("_"):
            print(k, {a: b for a, b in v.items() if a not in ("shape", "family")})


if __name__ == "__main__":
    main()
