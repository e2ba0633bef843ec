# same-box A/B of two builds of the library on the headline step: gpurun_ab/libief_old.so vs the in-tree one
run() {
  python bench.py --no-cpu-baseline --pie-images 0 --steps-1024 0 --exact-steps 0 --in-flight "" --steps 100 2>/dev/null \
    | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['ms_per_step'])"
}
for i in 1 2; do
  IEF_HIP_LIB=$(pwd)/gpurun_ab/libief_old.so run "old"
  run "new"
done
