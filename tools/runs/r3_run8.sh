#!/bin/bash
# round 3, GPU call 8: the banded disentangled kernels (attention_lai.hip): parity tests, then old / new timings
set -e
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests/test_gpu_attention.py tests/test_gpu_lai_mask_kernel.py tests/test_gpu_modules.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r3g_pytest.log 2>&1 || { tail -40 gpurun_out/r3g_pytest.log; exit 1; }
tail -3 gpurun_out/r3g_pytest.log
for w in 20 8; do
  VITK_ATTN_LAI=0 timeout -k 10 120 python tools/bench_attn.py --w $w
  timeout -k 10 120 python tools/bench_attn.py --w $w
done
timeout -k 10 300 python -u -m pytest tests/test_gpu_full_depth.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r3g_depth.log 2>&1 || { tail -40 gpurun_out/r3g_depth.log; exit 1; }
tail -2 gpurun_out/r3g_depth.log
