#!/bin/bash
# round 3, GPU call 9: phase timing of the attention_lai.hip kernels (early-exit variants), new parity cases, in-process reducer at steady state
set -e
export TMPDIR=/tmp
timeout -k 10 300 python -u -m pytest tests/test_gpu_attention.py -x -q --timeout 120 --timeout-method thread -k lai_dis > gpurun_out/r3h_pytest.log 2>&1 || { tail -40 gpurun_out/r3h_pytest.log; exit 1; }
tail -2 gpurun_out/r3h_pytest.log
for k in 1 2 3 4 5; do
  echo "stop after phase $k:"
  VITK_LIB=$PWD/tools/_lai/libvitk_stop$k.so timeout -k 10 120 python tools/bench_attn.py --w 20
done
echo full:
timeout -k 10 120 python tools/bench_attn.py --w 20
DIAG_STEPS=8 DIAG_WARMUP=4 timeout -k 10 200 python -u tools/diag_reducer_hang.py auto side 2>&1 | grep -v "^  File\|^    " | grep "ok\|DONE\|WATCHDOG" 
