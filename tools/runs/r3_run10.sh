#!/bin/bash
# round 3, GPU call 10: attention_lai.hip with batched loads: parity, phase timing
set -e
export TMPDIR=/tmp
timeout -k 10 300 python -u -m pytest tests/test_gpu_attention.py -x -q --timeout 120 --timeout-method thread -k "lai_dis or (dis and bf16)" > gpurun_out/r3i_pytest.log 2>&1 || { tail -40 gpurun_out/r3i_pytest.log; exit 1; }
tail -2 gpurun_out/r3i_pytest.log
for k in 2 3 4 5; do
  echo "stop after phase $k:"
  VITK_LIB=$PWD/tools/_lai/libvitk_stop$k.so timeout -k 10 120 python tools/bench_attn.py --w 20
done
echo full:
timeout -k 10 120 python tools/bench_attn.py --w 20
timeout -k 10 120 python tools/bench_attn.py --w 8
