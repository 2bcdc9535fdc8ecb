#!/bin/bash
export TMPDIR=/tmp
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-isolated --no-kernel-timer > gpurun_out/sw_$name.json 2> gpurun_out/sw_$name.err || { echo "$name failed"; tail -3 gpurun_out/sw_$name.err; return 0; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/sw_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],2))"; }
run base A=1
run dw_16_4 VITK_DW_SPLITK=16:4,64:2
run dw_16_3 VITK_DW_SPLITK=16:3,64:2
run dw_16_6 VITK_DW_SPLITK=16:6,64:2
run dw_16_8 VITK_DW_SPLITK=16:8,64:2
run dw_16_4_64_3 VITK_DW_SPLITK=16:4,64:3
run dw_16_4_48_3 VITK_DW_SPLITK=16:4,64:2,48:3
run dw_16_4_48_4 VITK_DW_SPLITK=16:4,64:2,48:4
run dw_16_4_48_5 VITK_DW_SPLITK=16:4,64:2,48:5
run base2 A=1
run dw_16_4b VITK_DW_SPLITK=16:4,64:2
