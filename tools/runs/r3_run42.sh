#!/bin/bash
set -e
export TMPDIR=/tmp
timeout -k 10 400 python -u -m pytest tests/test_gpu_dropout.py -x -q --timeout 120 --timeout-method thread -k "prob_dropout" 2>&1 | tail -3
timeout -k 10 400 python -u -m pytest tests/test_gpu_dropout.py tests/test_gpu_attention.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2
for e in 1 0; do
VITK_ATTN_FULL=$e timeout -k 10 300 python bench.py --workload dis_b16 --dropout 0.1 --steps 10 --warmup 3 --no-cpu-baseline --no-isolated > gpurun_out/sw_drop_full$e.json 2> gpurun_out/sw_drop_full$e.err || { tail -5 gpurun_out/sw_drop_full$e.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/sw_drop_full$e.json').read().strip().splitlines()[-1]); print('dis_b16 p=0.1 full=$e', round(d['value'],1), round(d['ms_per_step'],2), [(k['kernel'], round(k['avg_launch_us'],1)) for k in d['timed_kernels'] if 'attn' in k['kernel']])"
done
