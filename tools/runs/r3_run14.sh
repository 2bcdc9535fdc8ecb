#!/bin/bash
# round 3, GPU call 14: polynomial GELU in the bf16 GEMM epilogues: GEMM / module parity, isolated GEMM times, the step
set -e
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests/test_gpu_gemm_norm.py tests/test_gpu_modules.py tests/test_gpu_full_depth.py -x -q --timeout 200 --timeout-method thread > gpurun_out/r3m_pytest.log 2>&1 || { tail -40 gpurun_out/r3m_pytest.log; exit 1; }
tail -2 gpurun_out/r3m_pytest.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3m_default.json 2> gpurun_out/r3m_default.err || { tail -5 gpurun_out/r3m_default.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r3m_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print('default', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms loss', round(d['loss'],4), 'in-step NT us', round(r['avg_launch_us'],1), 'iso frac', round(r['isolated']['frac'],4))
for s in r['isolated']['per_shape']: print('  ', s['gemm'], round(s['us'],1), 'us', round(s['tflops'],1), 'TF/s')
"
