#!/bin/bash
set -e
export TMPDIR=/tmp
timeout -k 10 400 python -u -m pytest tests/test_gpu_attention.py tests/test_gpu_dropout.py -x -q --timeout 120 --timeout-method thread -k "lai or dropout or drop" 2>&1 | tail -2
for w in 20 8; do echo "== w $w"; timeout -k 10 120 python tools/bench_attn.py --w $w --iters 20; done
