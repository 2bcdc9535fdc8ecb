#!/bin/bash
# round 3, GPU call 11: attention_lai.hip v3 (lane-constant addressing) parity + timing alone and in the step (same box A/B)
set -e
export TMPDIR=/tmp
timeout -k 10 300 python -u -m pytest tests/test_gpu_attention.py tests/test_gpu_modules.py -x -q --timeout 120 --timeout-method thread -k "lai_dis or dis or golden" > gpurun_out/r3j_pytest.log 2>&1 || { tail -40 gpurun_out/r3j_pytest.log; exit 1; }
tail -2 gpurun_out/r3j_pytest.log
for hp in 8 2 1; do
  echo "heads per workgroup $hp:"
  VITK_LAI_HEADS_PER_WG=$hp timeout -k 10 120 python tools/bench_attn.py --w 20
done
for k in 3 5; do
  echo "stop after phase $k (8 heads per workgroup):"
  VITK_LAI_HEADS_PER_WG=8 VITK_LIB=$PWD/tools/_lai/libvitk_stop$k.so timeout -k 10 120 python tools/bench_attn.py --w 20
done
timeout -k 10 120 python tools/bench_attn.py --w 8
b() { name=$1; shift; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-isolated "$@" > gpurun_out/r3j_$name.json 2> gpurun_out/r3j_$name.err || { echo "$name failed"; tail -5 gpurun_out/r3j_$name.err; return 1; }
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r3j_$name.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}
tk={k['kernel']: round(k['avg_launch_us'],1) for k in d.get('timed_kernels', [])}
print('$name', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms', tk, 'loss', round(d['loss'],4))"
}
VITK_ATTN_LAI=0 b old_attn
VITK_LAI_HEADS_PER_WG=8 b lai_8
VITK_LAI_HEADS_PER_WG=2 b lai_2
VITK_LAI_HEADS_PER_WG=1 b lai_1
