#!/bin/bash
set -e
export TMPDIR=/tmp
timeout -k 10 200 python -u -m pytest tests/test_gpu_attention.py -x -q --timeout 120 --timeout-method thread -k "test_full_backward and 2-197" 2>&1 | tail -3
timeout -k 10 400 python -u -m pytest tests/test_gpu_attention.py -x -q --timeout 120 --timeout-method thread -k "test_full" 2>&1 | tail -3
echo "== new";  timeout -k 10 120 python tools/bench_attn.py --w -1 --iters 10
echo "== relds";  VITK_ATTN_FULL_TABGRAD=0 timeout -k 10 120 python tools/bench_attn.py --w -1 --iters 10
