#!/bin/bash
# round 3, GPU call 2: kernel trace of the step with the gradient exchange issued at world size 1
set -e
export TMPDIR=/tmp
for R in 0 16; do
  OUT=gpurun_out/r3b_red$R
  rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-isolated --force-reducer --reserve-cus $R > $OUT/bench.log 2>&1 || { tail -20 $OUT/bench.log; exit 1; }
  grep '^{' $OUT/bench.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench reserve $R', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')"
  T=$(find $OUT -name '*kernel_trace.csv' | head -1)
  python3 tools/trace_reducer.py $T
  S=$(find $OUT -name '*kernel_stats.csv' | head -1)
  cp $S gpurun_out/r3b_red${R}_kernel_stats.csv
  head -c 3000000 $T > gpurun_out/r3b_red${R}_kernel_trace_head.csv
done
b() { name=$1; shift; timeout -k 10 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/r3b_$name.json 2> gpurun_out/r3b_$name.err || { echo "$name failed"; tail -5 gpurun_out/r3b_$name.err; return 1; }
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r3b_$name.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}
print('$name', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms nt_us', r.get('avg_launch_us') and round(r['avg_launch_us'],1), 'loss', round(d['loss'],4))"
}
b default
b reducer_r0 --force-reducer --reserve-cus 0
b reducer_r16 --force-reducer --reserve-cus 16
b reducer_r32 --force-reducer --reserve-cus 32
b reducer_r16_bf16 --force-reducer --reserve-cus 16 --comm-dtype bf16
b default_r16 --reserve-cus 16
