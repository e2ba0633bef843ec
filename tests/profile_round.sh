#!/bin/bash
# One round's profiles on the GPU box (run through gpurun from the repo root): rocprofv3 kernel-trace summaries of the
# headline bench command at 512x512 and 1024x1024, FETCH_SIZE / WRITE_SIZE passes of the dominant kernels, and the default
# `python bench.py` line.  Everything lands under gpurun_out/prof_<tag>/ ; copy what is to be judged into profiles/.
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HEAD="--steps 20 --warmup 5 --no-cpu-baseline --pie-images 0 --steps-1024 0 --exact-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k512 -- python3 $ROOT/bench.py $HEAD --in-flight "" > $OUT/k512.log 2>&1
echo "k512 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k1024 -- python3 $ROOT/bench.py $HEAD --latent 128 --steps 10 --in-flight "" > $OUT/k1024.log 2>&1
echo "k1024 rc=$?"
for k in gemmsq conv64 conv64igemm conv32 attn40; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${k}_fetch -- python3 $ROOT/tests/one_kernel.py $k > $OUT/pmc_${k}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${k}_write -- python3 $ROOT/tests/one_kernel.py $k > $OUT/pmc_${k}_write.log 2>&1
  echo "pmc $k rc=$?"
done
cd $ROOT
find $OUT -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head -20
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc=$?"
tail -c 400 $OUT/bench_default.json
