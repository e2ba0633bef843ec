# re-tune the other shape families with the current kernels and merge everything into one table (run on the GPU box)
set -e
IEF_TUNE_FORWARD_ONLY=1 IEF_TUNE_LATENT=128 python tests/tune_plans.py gpurun_out/tuned_sd15_l128.json sd15 > gpurun_out/retune_l128.log 2>&1; echo "sd15@128 done"
python tests/tune_plans.py gpurun_out/tuned_sd21_new.json sd21 > gpurun_out/retune_sd21.log 2>&1; echo "sd21 done"
python tests/tune_plans.py gpurun_out/tuned_sdxl_new.json sdxl > gpurun_out/retune_sdxl.log 2>&1; echo "sdxl done"
python - <<'PY'
import json
merged = json.load(open("image-editing-framework_amd/tuned_plans.json"))
sd15 = set(json.load(open("gpurun_out/tuned_sd15_new.json"))) if __import__("os").path.exists("gpurun_out/tuned_sd15_new.json") else set()
for f in ("gpurun_out/tuned_sd15_l128.json", "gpurun_out/tuned_sd21_new.json", "gpurun_out/tuned_sdxl_new.json"):
    new = json.load(open(f))
    n = 0
    for k, v in new.items():
        if k in sd15:          # shapes the SD1.5 512x512 step uses keep the plan tuned on that step
            continue
        if merged.get(k) != v:
            n += 1
        merged[k] = v
    print(f, len(new), "shapes,", n, "changed")
json.dump({k: merged[k] for k in sorted(merged)}, open("gpurun_out/tuned_plans_all.json", "w"), indent=0)
PY
