#!/bin/bash
# env-level sweeps of the headline step on one box: GEMM start stagger, dW K splits
export TMPDIR=/tmp
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-isolated --no-kernel-timer > gpurun_out/sw_$name.json 2> gpurun_out/sw_$name.err || { echo "$name failed"; tail -3 gpurun_out/sw_$name.err; return 0; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/sw_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],2))"; }
run base A=1
run stag0 VITK_GEMM_STAGGER=0
run stag12 VITK_GEMM_STAGGER=12
run stag50 VITK_GEMM_STAGGER=50
run base2 A=1
run dw_16_4 VITK_DW_SPLITK=16:4,64:2
run dw_64_4 VITK_DW_SPLITK=16:2,64:4
run dw_16_1 VITK_DW_SPLITK=16:1,64:2
run dw_48 VITK_DW_SPLITK=16:2,64:2,48:2
run base3 A=1
