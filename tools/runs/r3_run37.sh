#!/bin/bash
export TMPDIR=/tmp
run() { wl=$1; name=$2; shift 2; env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 12 --warmup 4 --no-cpu-baseline --no-isolated --no-kernel-timer > gpurun_out/sw_$name.json 2> gpurun_out/sw_$name.err || { echo "$name failed"; tail -3 gpurun_out/sw_$name.err; return 0; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/sw_$name.json').read().strip().splitlines()[-1]); print('$wl $name', round(d['value'],1), round(d['ms_per_step'],2))"; }
for wl in performer_s16_384 lai_s16; do
run $wl base A=1
run $wl p9_4 VITK_DW_SPLITK=16:4,64:2,9:4
run $wl p9_8 VITK_DW_SPLITK=16:4,64:2,9:8
run $wl f36_2 VITK_DW_SPLITK=16:4,64:2,36:2
run $wl f36_4 VITK_DW_SPLITK=16:4,64:2,36:4
run $wl q27_3 VITK_DW_SPLITK=16:4,64:2,27:3
run $wl q27_5 VITK_DW_SPLITK=16:4,64:2,27:5
run $wl all VITK_DW_SPLITK=16:4,64:2,9:8,36:4,27:5
run $wl base2 A=1
done
