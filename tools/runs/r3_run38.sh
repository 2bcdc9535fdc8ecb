#!/bin/bash
export TMPDIR=/tmp
run() { wl=$1; name=$2; shift 2; env "$@" timeout -k 10 300 python bench.py --workload $wl --band 8 --steps 10 --warmup 3 --no-cpu-baseline --no-isolated --no-kernel-timer > gpurun_out/sw_$name.json 2> gpurun_out/sw_$name.err || { echo "$name failed"; tail -3 gpurun_out/sw_$name.err; return 0; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/sw_$name.json').read().strip().splitlines()[-1]); print('$wl $name', round(d['value'],1), round(d['ms_per_step'],2))"; }
wl=dislai_l16
run $wl base A=1
run $wl p25_4 VITK_DW_SPLITK=25:4
run $wl p25_6 VITK_DW_SPLITK=25:6
run $wl p25_2 VITK_DW_SPLITK=25:2
run $wl q75_2 VITK_DW_SPLITK=75:2
run $wl q75_4 VITK_DW_SPLITK=75:4
run $wl f100_1 VITK_DW_SPLITK=100:1
run $wl f100_3 VITK_DW_SPLITK=100:3
run $wl base2 A=1
