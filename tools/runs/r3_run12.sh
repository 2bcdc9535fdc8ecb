#!/bin/bash
# round 3, GPU call 12: attention_lai.hip (forward with lane-constant addressing, backward v2): parity, timing, step A/B
set -e
export TMPDIR=/tmp
timeout -k 10 400 python -u -m pytest tests/test_gpu_attention.py tests/test_gpu_modules.py tests/test_gpu_lai_mask_kernel.py tests/test_gpu_dropout.py tests/test_gpu_graph.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r3k_pytest.log 2>&1 || { tail -40 gpurun_out/r3k_pytest.log; exit 1; }
tail -2 gpurun_out/r3k_pytest.log
timeout -k 10 120 python tools/bench_attn.py --w 20
timeout -k 10 120 python tools/bench_attn.py --w 8
b() { name=$1; shift; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-isolated "$@" > gpurun_out/r3k_$name.json 2> gpurun_out/r3k_$name.err || { echo "$name failed"; tail -5 gpurun_out/r3k_$name.err; return 1; }
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r3k_$name.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}
tk={k['kernel']: round(k['avg_launch_us'],1) for k in d.get('timed_kernels', [])}
print('$name', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms', tk, 'loss', round(d['loss'],4))"
}
VITK_ATTN_LAI=0 b old_attn
b lai
VITK_ATTN_LAI=0 b old_attn_w8 --band 8
b lai_w8 --band 8
