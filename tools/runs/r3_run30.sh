#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 300 python -u -m pytest tests/test_gpu_ddp_nccl.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3y_nccl.log 2>&1; echo "nccl alone rc=$?"; tail -3 gpurun_out/r3y_nccl.log | cut -c1-200
[ $? -eq 0 ] || exit 1
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3y_pytest.log 2>&1; echo "suite rc=$?"; grep -n "passed\|failed\|Fatal" gpurun_out/r3y_pytest.log | head -5
