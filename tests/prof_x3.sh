#!/bin/bash
# rocprofv3 kernel-trace summary of the headline bench command in the f16x3 mode (usage: bash tests/prof_x3.sh <tag> [latent]); the
# stats CSV lands in gpurun_out/prof_<tag>/ and its top rows are printed
TAG=${1:-r04}; LAT=${2:-0}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HEAD="--steps 20 --warmup 5 --no-cpu-baseline --pie-images 0 --steps-1024 0 --other-modes= --in-flight= --precision f16x3"
[ "$LAT" != "0" ] && HEAD="$HEAD --latent $LAT --steps 6"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k_$LAT -- python3 $ROOT/bench.py $HEAD > $OUT/k_$LAT.log 2>&1
echo "trace rc=$?"
cd $ROOT
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/k_$LAT/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {int(r["Calls"]):6d} x {float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][:110]}')
PY
python3 bench.py --no-cpu-baseline --pie-images 0 --steps-1024 0 --other-modes= --in-flight= --steps 50 2>/dev/null | tail -1 | cut -c1-400
