#!/bin/bash
# plain banded attention on the LAI kernels: parity + timings old / new
set -e
export TMPDIR=/tmp
OUT=gpurun_out/r3r
mkdir -p $OUT
timeout -k 10 600 python -u -m pytest tests/test_gpu_attention.py -x -q --timeout 120 --timeout-method thread -k "lai" > $OUT/attn.log 2>&1 || { tail -30 $OUT/attn.log; exit 1; }
tail -2 $OUT/attn.log
for w in 8 20; do
  echo "== plain w=$w H=12 new"; timeout -k 10 120 python tools/bench_attn.py --plain --heads 12 --w $w
  echo "== plain w=$w H=12 old"; VITK_ATTN_LAI=0 timeout -k 10 120 python tools/bench_attn.py --plain --heads 12 --w $w
done
timeout -k 10 900 python -u -m pytest tests/test_gpu_attention.py tests/test_gpu_modules.py tests/test_gpu_dropout.py -x -q --timeout 120 --timeout-method thread > $OUT/attn_all.log 2>&1 || { tail -30 $OUT/attn_all.log; exit 1; }
tail -2 $OUT/attn_all.log
