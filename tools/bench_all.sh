#!/bin/bash
# Every workload of DESIGN.md section 5, one JSON line each into gpurun_out/all_<name>.json.
set -e
mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/all_$name.json 2> gpurun_out/all_$name.err; python - "$name" <<'PY'
import json, sys
name = sys.argv[1]
d = json.loads(open(f'gpurun_out/all_{name}.json').read().strip().splitlines()[-1])
print(f"{name:24s} {d['value']:10.1f} img/s  {d['ms_per_step']:8.2f} ms/step  {d.get('model_tflops_per_gpu', 0):7.1f} TFLOP/s", flush=True)
PY
}
run dislai_b16_w20 --workload dislai_b16 --band 20
run dislai_b16_w8 --workload dislai_b16 --band 8
run dislai_b16_unmasked --workload dislai_b16 --band -1
run dis_b16 --workload dis_b16
run lai_s16_w8 --workload lai_s16 --band 8
run lai_s16_unmasked --workload lai_s16 --band -1
run dislai_l16_w8 --workload dislai_l16 --band 8
run regular_cifar --workload regular_cifar
run performer_s16_384 --workload performer_s16_384
run softmax_s16_384 --workload softmax_s16_384
run disperf_s16 --workload disperf_s16
