#!/bin/bash
# drop-path tests + FAVOR+ error report
set -e
export TMPDIR=/tmp
OUT=gpurun_out/r3q
mkdir -p $OUT
timeout -k 10 300 python -u -m pytest tests/test_gpu_dropout.py -x -q --timeout 120 --timeout-method thread > $OUT/dropout.log 2>&1 || { tail -30 $OUT/dropout.log; exit 1; }
tail -3 $OUT/dropout.log
VITK_REPORT_ERRORS=$OUT/favor_errs.jsonl timeout -k 10 400 python -u -m pytest tests/test_gpu_performer.py -x -q --timeout 120 --timeout-method thread > $OUT/performer.log 2>&1 || { tail -30 $OUT/performer.log; exit 1; }
tail -3 $OUT/performer.log
wc -l $OUT/favor_errs.jsonl
