#!/bin/bash
# round 3, GPU call 13: full-size parity of attention_lai.hip; run-to-run spread of the w = 8 bench loss; PMC traffic
set -e
export TMPDIR=/tmp
timeout -k 10 400 python -u -m pytest tests/test_gpu_attention.py -x -q --timeout 200 --timeout-method thread -k "lai_dis" > gpurun_out/r3l_pytest.log 2>&1 || { tail -40 gpurun_out/r3l_pytest.log; exit 1; }
tail -2 gpurun_out/r3l_pytest.log
b() { name=$1; shift; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-isolated --no-kernel-timer "$@" > gpurun_out/r3l_$name.json 2> gpurun_out/r3l_$name.err || { echo "$name failed"; tail -5 gpurun_out/r3l_$name.err; return 1; }
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r3l_$name.json').read().strip().splitlines()[-1])
print('$name', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms loss', round(d['loss'],4))"
}
VITK_ATTN_LAI=0 b old_w8_a --band 8
VITK_ATTN_LAI=0 b old_w8_b --band 8
b lai_w8_a --band 8
b lai_w8_b --band 8
b lai_w12 --band 12
VITK_ATTN_LAI=0 b old_w12 --band 12
tools/collect_pmc.sh $1 r3l_headline
