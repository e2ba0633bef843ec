# ablation builds of the planes GEMM (gpurun_ab/libief_abl{bits}.so: 1 no LDS-DMA, 2 no fragment reads, 4 one MFMA per block, 8 no barrier)
for a in "" 1 2 4 8 3 7; do
  if [ -z "$a" ]; then echo "== full"; python tests/bench_x3p.py --tiles ${TILES:-1,2} $1
  else echo "== ablation $a"; IEF_HIP_LIB=$(pwd)/gpurun_ab/libief_abl$a.so python tests/bench_x3p.py --tiles ${TILES:-1,2} $1; fi
done
