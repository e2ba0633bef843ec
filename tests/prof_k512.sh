ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_mid; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k512 -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --pie-images 0 --steps-1024 0 --exact-steps 0 --in-flight "" > $OUT/k512.log 2>&1
echo "k512 rc=$?"; tail -c 300 $OUT/k512.log
