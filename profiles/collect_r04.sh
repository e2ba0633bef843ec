#!/bin/bash
# profiles/r04_* from gpurun_out/prof_r04 (what `tests/profile_r04.sh` wrote on the GPU box): run from the repo root.
set -e
P=gpurun_out/prof_r04
one() { ls -t $1/*/*_kernel_stats.csv | head -n 1; }
cp "$(one $P/k512_f16x3)" profiles/r04_bench_kernel_stats_f16x3.csv
cp "$(one $P/k1024_f16x3)" profiles/r04_bench_1024_kernel_stats_f16x3.csv
cp "$(one $P/k512_f16)" profiles/r04_bench_kernel_stats_f16.csv
cp "$(one $P/nti_f16x3)" profiles/r04_nti_kernel_stats_f16x3.csv
python3 profiles/make_pmc_traffic.py $P r04 > /dev/null
rm -f profiles/r04_roofline_inputs.json
python3 profiles/make_roofline_inputs.py profiles/r04_bench_kernel_stats_f16x3.csv sd15 64 profiles/r04_pmc_traffic.json --precision f16x3 --tag r04
python3 profiles/make_roofline_inputs.py profiles/r04_bench_1024_kernel_stats_f16x3.csv sd15 128 profiles/r04_pmc_traffic.json --precision f16x3 --tag r04
python3 profiles/make_roofline_inputs.py profiles/r04_bench_kernel_stats_f16.csv sd15 64 --precision f16 --tag r04
for k in conv64x3p gemmsqx3p attn40x3p; do cp $P/sq_$k.txt profiles/r04_sq_$k.txt; done
python3 profiles/make_step_breakdown.py "$(ls -t $P/k512_f16x3/*/*_kernel_trace.csv | head -n 1)" profiles/r04_step_breakdown_f16x3.json
