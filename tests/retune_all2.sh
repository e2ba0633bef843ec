# re-tune every shape family with the current kernels (run on the GPU box); the SD1.5 512 px step's shapes win key collisions
set -e
python tests/tune_plans.py gpurun_out/tuned_sd15_new.json sd15 > gpurun_out/retune.log 2>&1; echo "sd15 done"
IEF_TUNE_FORWARD_ONLY=1 IEF_TUNE_LATENT=128 python tests/tune_plans.py gpurun_out/tuned_sd15_l128.json sd15 > gpurun_out/retune_l128.log 2>&1; echo "sd15@128 done"
python tests/tune_plans.py gpurun_out/tuned_sd21_new.json sd21 > gpurun_out/retune_sd21.log 2>&1; echo "sd21 done"
python tests/tune_plans.py gpurun_out/tuned_sdxl_new.json sdxl > gpurun_out/retune_sdxl.log 2>&1; echo "sdxl done"
python - <<'PY'
import json
merged = json.load(open("image-editing-framework_amd/tuned_plans.json"))
sd15 = json.load(open("gpurun_out/tuned_sd15_new.json"))
for f in ("gpurun_out/tuned_sd15_l128.json", "gpurun_out/tuned_sd21_new.json", "gpurun_out/tuned_sdxl_new.json"):
    for k, v in json.load(open(f)).items():
        if k not in sd15:
            merged[k] = v
merged.update(sd15)
json.dump({k: merged[k] for k in sorted(merged)}, open("gpurun_out/tuned_plans_all.json", "w"), indent=0)
print(len(merged), "plans;", sum(1 for k in merged if k.endswith("|u")), "upsample keys")
PY
run() {
  python bench.py --no-cpu-baseline --pie-images 0 --steps-1024 0 --exact-steps 0 --in-flight "" --steps 100 2>/dev/null \
    | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['ms_per_step'])"
}
for i in 1 2; do
  run "committed"
  IEF_PLAN_FILE=$(pwd)/gpurun_out/tuned_plans_all.json run "retuned  "
done
