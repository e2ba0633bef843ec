# same-box A/B of the GroupNorm forms (environment switches of csrc/norm.hip); prints ms per step and GroupNorm ms per step
run() {
  python bench.py --no-cpu-baseline --pie-images 0 --steps-1024 0 --exact-steps 0 --in-flight "" --steps 100 2>/dev/null \
    | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['ms_per_step'], d['roofline']['per_kernel_ms'].get('groupnorm(stats+apply)'))"
}
for i in 1 2; do
  IEF_GN_KS_MAX=1 IEF_GN_CSTAT_FUSED=0 run "base      "
  IEF_GN_KS_MAX=8 IEF_GN_CSTAT_FUSED=0 run "ks8       "
  IEF_GN_KS_MAX=8 IEF_GN_CSTAT_FUSED=1 run "ks8+fused "
  IEF_GN_KS_MAX=2 IEF_GN_CSTAT_FUSED=1 run "ks2+fused "
done
