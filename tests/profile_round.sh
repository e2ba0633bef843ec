#!/bin/bash
# One round's profiles on the GPU box (run through gpurun from the repo root): rocprofv3 kernel-trace summaries of the
# headline bench command in every precision mode at 512x512 (and the headline mode at 1024x1024), FETCH_SIZE / WRITE_SIZE
# passes of the dominant kernels, and the default `python bench.py` line.  Everything lands under gpurun_out/prof_<tag>/ ;
# copy what is to be judged into profiles/.     usage: bash tests/profile_round.sh r03 [trace|nti|pmc|bench ...]
TAG=${1:-r03}; shift
WHAT=${*:-trace nti pmc bench}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HEAD="--steps 20 --warmup 5 --no-cpu-baseline --pie-images 0 --steps-1024 0 --other-modes= --in-flight="
if [[ " $WHAT " == *" trace "* ]]; then
  for mode in f16x3 f16 f32; do
    steps=20; [ $mode = f32 ] && steps=6
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k512_$mode -- python3 $ROOT/bench.py $HEAD --precision $mode --steps $steps > $OUT/k512_$mode.log 2>&1
    echo "k512 $mode rc=$?"
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k1024_f16x3 -- python3 $ROOT/bench.py $HEAD --precision f16x3 --latent 128 --steps 6 > $OUT/k1024_f16x3.log 2>&1
  echo "k1024 f16x3 rc=$?"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k1024_f16 -- python3 $ROOT/bench.py $HEAD --precision f16 --latent 128 --steps 10 > $OUT/k1024_f16.log 2>&1
  echo "k1024 f16 rc=$?"
fi
if [[ " $WHAT " == *" nti "* ]]; then      # the null-text inner iteration (forward + reverse pass + Adam), SD1.5 64x64, UNet batch 1
  for mode in f16 f16x3; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nti_$mode -- python3 $ROOT/tests/bench_nti.py --skip-full --iters 20 --precision $mode > $OUT/nti_$mode.log 2>&1
    echo "nti $mode rc=$?"
    tail -n 1 $OUT/nti_$mode.log
  done
fi
if [[ " $WHAT " == *" pmc "* ]]; then
  for k in conv64x3 conv32x3 gemmsqx3 gemmffx3 attn40x3 gemmsq conv64 conv32 attn40; do
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${k}_fetch -- python3 $ROOT/tests/one_kernel.py $k > $OUT/pmc_${k}_fetch.log 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${k}_write -- python3 $ROOT/tests/one_kernel.py $k > $OUT/pmc_${k}_write.log 2>&1
    echo "pmc $k rc=$?"
  done
fi
cd $ROOT
if [[ " $WHAT " == *" bench "* ]]; then
  python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
  echo "bench rc=$?"
  tail -c 300 $OUT/bench_default.json
fi
