#!/bin/bash
# rocprofv3 --kernel-trace --stats of the headline bench (7 steps incl. warm-up), compact per-step table.
# usage (on the GPU box): tools/prof_step.sh <tag> [bench args...]   -> gpurun_out/<tag>_kernel_stats.csv + table on stdout
set -e
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer "$@" > $OUT/bench.log 2>&1
CSV=$(find $OUT -name '*kernel_stats.csv' | head -1)
cp $CSV gpurun_out/${TAG}_kernel_stats.csv
tail -1 $OUT/bench.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')" || tail -5 $OUT/bench.log
python3 - "$CSV" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'total kernel time per step {tot / 7e6:.2f} ms ({len(rows)} kernels, {sum(int(r["Calls"]) for r in rows) / 7:.0f} launches/step)')
for r in rows[:22]:
    name = re.sub(r'\(.*', '', r['Name'])[:70]
    print(f'{float(r["TotalDurationNs"]) / 7e6:7.2f} ms/step {int(r["Calls"]) / 7:7.1f} calls/step {float(r["AverageNs"]) / 1e3:8.1f} us  {name}')
PY
