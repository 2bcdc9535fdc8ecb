#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests/test_gpu_full_depth.py -x -q -s --timeout 400 --timeout-method thread -k "oracle" > gpurun_out/r3z_fd.log 2>&1; echo "rc=$?"; grep -n "passed\|failed\|logits" gpurun_out/r3z_fd.log | cut -c1-400 | tail -8
