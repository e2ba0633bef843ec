# ablation builds of the halo convolution on operand planes (gpurun_ab/libief_abl{bits}.so: 1 no LDS-DMA in the loop, 2 no fragment reads, 4 one MFMA of three)
for a in "" 1 2 4 3 7; do
  if [ -z "$a" ]; then echo "== full"; python tests/bench_x3p.py --conv --tiles 11 2>&1 | grep "t11\|t12"
  else echo "== ablation $a"; IEF_HIP_LIB=$(pwd)/gpurun_ab/libief_abl$a.so python tests/bench_x3p.py --conv --tiles 11 2>&1 | grep "t11\|t12"; fi
done
