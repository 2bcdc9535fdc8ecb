#!/bin/bash
# round 3, GPU call 15: dropout streams inside the fused disentangled layer node: parity with exported masks, the p = 0.1 row
set -e
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests/test_gpu_dropout.py tests/test_gpu_modules.py tests/test_gpu_eval.py tests/test_gpu_graph.py -x -q --timeout 200 --timeout-method thread > gpurun_out/r3o_pytest.log 2>&1 || { tail -60 gpurun_out/r3o_pytest.log; exit 1; }
tail -2 gpurun_out/r3o_pytest.log
b() { name=$1; shift; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-isolated "$@" > gpurun_out/r3o_$name.json 2> gpurun_out/r3o_$name.err || { echo "$name failed"; tail -5 gpurun_out/r3o_$name.err; return 1; }
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r3o_$name.json').read().strip().splitlines()[-1])
tk={k['kernel']: round(k['avg_launch_us'],1) for k in d.get('timed_kernels', [])}
print('$name', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms', tk, 'loss', round(d['loss'],4))"
}
b p0
b p01 --dropout 0.1
