# re-tune the SD1.5 layer shapes with the current kernels, merge into the committed table, and A/B the two tables on this box
set -e
python tests/tune_plans.py gpurun_out/tuned_sd15_new.json sd15 > gpurun_out/retune.log 2>&1
python - <<'PY'
import json
old = json.load(open("image-editing-framework_amd/tuned_plans.json"))
new = json.load(open("gpurun_out/tuned_sd15_new.json"))
changed = {k: (old.get(k), v) for k, v in new.items() if old.get(k) != v}
print(len(new), "shapes tuned,", len(changed), "plans differ from the committed table")
merged = dict(old); merged.update(new)
json.dump({k: merged[k] for k in sorted(merged)}, open("gpurun_out/tuned_plans_merged.json", "w"), indent=0)
PY
run() {
  python bench.py --no-cpu-baseline --pie-images 0 --steps-1024 0 --exact-steps 0 --in-flight "" --steps 100 2>/dev/null \
    | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['ms_per_step'])"
}
for i in 1 2; do
  IEF_PLAN_FILE=image-editing-framework_amd/tuned_plans.json run "committed"
  IEF_PLAN_FILE=gpurun_out/tuned_plans_merged.json run "retuned  "
done
