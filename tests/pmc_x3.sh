# SQ counter passes of the split-operand kernels (usage: bash tests/pmc_x3.sh conv64x3 gemmsqx3 ...), run through gpurun
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_x3; mkdir -p $O
for k in "$@"; do
  rm -rf $O/${k}_a $O/${k}_b $O/${k}_t
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/${k}_a -- python3 $R/tests/one_kernel.py $k > $O/${k}_a.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_FLAT --output-format csv -d $O/${k}_b -- python3 $R/tests/one_kernel.py $k > $O/${k}_b.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${k}_t -- python3 $R/tests/one_kernel.py $k > $O/${k}_t.log 2>&1
  echo "$k done"
done
