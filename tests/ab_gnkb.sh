run() {
  python bench.py --no-cpu-baseline --pie-images 0 --steps-1024 0 --exact-steps 0 --in-flight "" --steps 100 2>/dev/null \
    | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['ms_per_step'])"
}
for i in 1 2; do
  IEF_GN_CSTAT_FUSED_KB=16 run "16KB"
  IEF_GN_CSTAT_FUSED_KB=32 run "32KB"
  IEF_GN_CSTAT_FUSED_KB=64 run "64KB"
done
