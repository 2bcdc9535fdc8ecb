#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests/test_gpu_full_depth.py -x -q -s --timeout 400 --timeout-method thread -k "timm_lai" > gpurun_out/r3z_tl.log 2>&1; echo "rc=$?"; grep -n "passed\|failed\|logits\|Error\|error" gpurun_out/r3z_tl.log | cut -c1-300 | tail -8
