# the other shape families on one box: plan table of the round's start vs the committed one (SD2.1 768x768, SDXL 1024x1024; P2P edit step, batch 4)
run() {
  python bench.py --config $2 --no-cpu-baseline --pie-images 0 --steps-1024 0 --exact-steps 0 --in-flight "" --steps 20 2>/dev/null \
    | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$2', '$1', d['ms_per_step'], 'ms', d['value'], 'steps/s')"
}
for cfg in sd21 sdxl; do
  IEF_HALO_HEURISTIC=0 IEF_PLAN_FILE=$(pwd)/tests/plans/tuned_plans_r02start.json run "round-start plans" $cfg
  run "committed plans  " $cfg
done
