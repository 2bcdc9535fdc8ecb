#!/bin/bash
# round 3, GPU call 4 (re-entry): the GPU suite on the restored tree, the default line, the dropout 0.1 row, a kernel profile and PMC traffic
set -e
export TMPDIR=/tmp
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3n_pytest.log 2>&1 || { tail -30 gpurun_out/r3n_pytest.log; exit 1; }
tail -3 gpurun_out/r3n_pytest.log
b() { name=$1; shift; timeout -k 10 300 python bench.py "$@" > gpurun_out/r3n_$name.json 2> gpurun_out/r3n_$name.err || { echo "$name failed"; tail -5 gpurun_out/r3n_$name.err; return 1; }
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r3n_$name.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}
print('$name', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms nt_us', r.get('avg_launch_us') and round(r['avg_launch_us'],1), 'loss', round(d['loss'],4))"
}
b default
b dropout01 --dropout 0.1 --no-cpu-baseline --steps 10 --warmup 3
tools/prof_step.sh r3n_headline
tools/collect_pmc.sh $1 r3n_headline
