# same-box A/B of the headline step: the plan table of the start of round 2 (implicit-GEMM tiles only) vs the committed one
run() {
  python bench.py --no-cpu-baseline --pie-images 0 --steps-1024 0 --exact-steps 0 --in-flight "" --steps 100 2>/dev/null \
    | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['ms_per_step'])"
}
for i in 1 2 3; do
  IEF_HALO_HEURISTIC=0 IEF_PLAN_FILE=$(pwd)/tests/plans/tuned_plans_r02start.json run "round-start plans"
  run "committed plans  "
done
