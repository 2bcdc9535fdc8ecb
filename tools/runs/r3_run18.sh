#!/bin/bash
# unmasked disentangled forward: parity (smallest case first), timing old / new
set -e
export TMPDIR=/tmp
OUT=gpurun_out/r3s
mkdir -p $OUT
timeout -k 10 200 python -u -m pytest tests/test_gpu_attention.py -x -q --timeout 120 --timeout-method thread -k "full_dis_forward and 2-197" > $OUT/a0.log 2>&1 || { tail -30 $OUT/a0.log; exit 1; }
tail -2 $OUT/a0.log
timeout -k 10 300 python -u -m pytest tests/test_gpu_attention.py -x -q --timeout 120 --timeout-method thread -k "full_dis_forward" > $OUT/a1.log 2>&1 || { tail -30 $OUT/a1.log; exit 1; }
tail -2 $OUT/a1.log
echo "== new"; timeout -k 10 120 python tools/bench_attn.py --w -1
echo "== old"; VITK_ATTN_FULL=0 timeout -k 10 120 python tools/bench_attn.py --w -1
timeout -k 10 900 python -u -m pytest tests/test_gpu_attention.py tests/test_gpu_modules.py tests/test_gpu_dropout.py -x -q --timeout 120 --timeout-method thread > $OUT/attn_all.log 2>&1 || { tail -30 $OUT/attn_all.log; exit 1; }
tail -2 $OUT/attn_all.log
