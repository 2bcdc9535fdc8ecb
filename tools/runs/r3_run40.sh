#!/bin/bash
export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-isolated --no-kernel-timer "$@" > gpurun_out/sw_$name.json 2> gpurun_out/sw_$name.err || { echo "$name failed"; tail -3 gpurun_out/sw_$name.err; return 0; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/sw_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],2), d['config'].get('reserved_cus'))"; }
run base
run rc8 --reserve-cus 8
run rc16 --reserve-cus 16
run rc32 --reserve-cus 32
run rc64 --reserve-cus 64
run base2
