#!/bin/bash
# Round-4 profiles on the GPU box (through gpurun, from the repo root): rocprofv3 kernel-trace summaries of the headline bench
# command in the f16x3 mode at 512x512 and 1024x1024 (and the fp16 path for context), the null-text inner iteration, FETCH_SIZE /
# WRITE_SIZE passes and SQ counter passes of the dominant planes kernels, the default `python bench.py` line.
# usage: bash tests/profile_r04.sh [trace|nti|pmc|sq|bench ...]   -> gpurun_out/prof_r04/ ; copy what is to be judged into profiles/
WHAT=${*:-trace nti pmc sq bench}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_r04; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HEAD="--steps 20 --warmup 5 --no-cpu-baseline --pie-images 0 --nti-images 0 --steps-1024 0 --other-modes= --in-flight="
if [[ " $WHAT " == *" trace "* ]]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k512_f16x3 -- python3 $ROOT/bench.py $HEAD --precision f16x3 > $OUT/k512_f16x3.log 2>&1; echo "k512 f16x3 rc=$?"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k1024_f16x3 -- python3 $ROOT/bench.py $HEAD --precision f16x3 --latent 128 --steps 6 > $OUT/k1024_f16x3.log 2>&1; echo "k1024 f16x3 rc=$?"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k512_f16 -- python3 $ROOT/bench.py $HEAD --precision f16 > $OUT/k512_f16.log 2>&1; echo "k512 f16 rc=$?"
fi
if [[ " $WHAT " == *" nti "* ]]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nti_f16x3 -- python3 $ROOT/tests/bench_nti.py --skip-full --iters 20 --precision f16x3 > $OUT/nti_f16x3.log 2>&1; echo "nti rc=$?"; tail -n 1 $OUT/nti_f16x3.log
fi
if [[ " $WHAT " == *" pmc "* ]]; then
  for k in conv64x3p conv32x3p gemmsqx3p gemmffx3p attn40x3p; do
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${k}_fetch -- python3 $ROOT/tests/one_kernel.py $k > $OUT/pmc_${k}_fetch.log 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${k}_write -- python3 $ROOT/tests/one_kernel.py $k > $OUT/pmc_${k}_write.log 2>&1
    echo "pmc $k rc=$?"
  done
fi
cd $ROOT
if [[ " $WHAT " == *" sq "* ]]; then
  bash tests/pmc_x3.sh conv64x3p gemmsqx3p attn40x3p > $OUT/sq.log 2>&1
  for k in conv64x3p gemmsqx3p attn40x3p; do
    needle=x3p; python3 tests/sum_pmc.py gpurun_out/pmc_x3 $k $needle > $OUT/sq_$k.txt 2>&1
  done
  echo "sq done"
fi
if [[ " $WHAT " == *" bench "* ]]; then
  python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -c 400 $OUT/bench_default.json
fi
