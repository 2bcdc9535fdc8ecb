#!/bin/bash
# the GPU suite twice in a row on one box (stability check; stops at the first failure)
export TMPDIR=/tmp
for i in 1 2; do
  timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3z_suite$i.log 2>&1 || { echo "run $i failed"; tail -30 gpurun_out/r3z_suite$i.log; exit 1; }
  tail -1 gpurun_out/r3z_suite$i.log
done
