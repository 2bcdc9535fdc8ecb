#!/bin/bash
# unmasked disentangled backward: smallest case first, then the others, then timing old / new
set -e
export TMPDIR=/tmp
OUT=gpurun_out/r3u
mkdir -p $OUT
timeout -k 10 200 python -u -m pytest tests/test_gpu_attention.py -x -q --timeout 120 --timeout-method thread -k "full_dis_backward and 2-197" > $OUT/a0.log 2>&1 || { tail -40 $OUT/a0.log; exit 1; }
tail -2 $OUT/a0.log
timeout -k 10 300 python -u -m pytest tests/test_gpu_attention.py -x -q --timeout 120 --timeout-method thread -k "full_dis" > $OUT/a1.log 2>&1 || { tail -40 $OUT/a1.log; exit 1; }
tail -2 $OUT/a1.log
echo "== new"; timeout -k 10 120 python tools/bench_attn.py --w -1 --iters 10
echo "== old bwd"; VITK_ATTN_FULL_BWD=0 timeout -k 10 120 python tools/bench_attn.py --w -1 --iters 10
