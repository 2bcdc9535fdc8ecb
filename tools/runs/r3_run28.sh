#!/bin/bash
set -e
export TMPDIR=/tmp
timeout -k 10 300 python -u -m pytest tests/test_gpu_attention.py -x -q --timeout 120 --timeout-method thread -k "full_dis" 2>&1 | tail -2
echo "== full"; timeout -k 10 120 python tools/bench_attn.py --w -1 --iters 10
