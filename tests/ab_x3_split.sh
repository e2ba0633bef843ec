for cfg in "384 512" "256 512" "384 768" "512 768" "200 384"; do
  set -- $cfg
  IEF_X3_SPLIT_BELOW=$1 IEF_X3_SPLIT_TARGET=$2 python tests/prof_exact.py sd15 f16x3 2>&1 | grep -E "whole eager|timed kernels" | tr '\n' ' '
  echo " <- below=$1 target=$2"
done
