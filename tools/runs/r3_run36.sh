#!/bin/bash
export TMPDIR=/tmp
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-isolated --no-kernel-timer > gpurun_out/sw_$name.json 2> gpurun_out/sw_$name.err || { echo "$name failed"; tail -3 gpurun_out/sw_$name.err; return 0; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/sw_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],2))"; }
run base A=1
run sg2 VITK_GEMM_STAGGER_GROUPS=2
run sg8 VITK_GEMM_STAGGER_GROUPS=8
run rc8 VITK_RESERVED_CUS=8
run rc16 VITK_RESERVED_CUS=16
run rc32 VITK_RESERVED_CUS=32
run lai4 VITK_LAI_HEADS_PER_WG=4
run lai16 VITK_LAI_HEADS_PER_WG=16
run nocolsum VITK_SIDE_COLSUM=0
run base2 A=1
