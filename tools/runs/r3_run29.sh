#!/bin/bash
set -e
export TMPDIR=/tmp
timeout -k 10 200 python -u -m pytest tests/test_gpu_attention.py -x -q --timeout 120 --timeout-method thread -k "test_full and 2-197" 2>&1 | tail -2
timeout -k 10 400 python -u -m pytest tests/test_gpu_attention.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2
echo "== plain new";  timeout -k 10 120 python tools/bench_attn.py --plain --heads 12 --w -1 --iters 10
echo "== plain old";  VITK_ATTN_FULL=0 timeout -k 10 120 python tools/bench_attn.py --plain --heads 12 --w -1 --iters 10
