#!/bin/bash
set -e
export TMPDIR=/tmp
OUT=gpurun_out/r3v
mkdir -p $OUT
timeout -k 10 900 python -u -m pytest tests/test_gpu_attention.py tests/test_gpu_modules.py tests/test_gpu_dropout.py tests/test_gpu_full_depth.py -x -q --timeout 300 --timeout-method thread > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
timeout -k 10 300 python bench.py --workload dis_b16 --steps 10 --warmup 3 --no-cpu-baseline --no-isolated > $OUT/dis_b16_new.json 2> $OUT/dis_b16_new.err || { tail -5 $OUT/dis_b16_new.err; exit 1; }
VITK_ATTN_FULL=0 timeout -k 10 300 python bench.py --workload dis_b16 --steps 10 --warmup 3 --no-cpu-baseline --no-isolated > $OUT/dis_b16_old.json 2> $OUT/dis_b16_old.err || { tail -5 $OUT/dis_b16_old.err; exit 1; }
python - <<'PY'
import json
for t in ('new','old'):
    d=json.load(open(f'gpurun_out/r3v/dis_b16_{t}.json'))
    print(t, round(d['value'],1), round(d['ms_per_step'],2), [ (k['kernel'], round(k['avg_launch_us'],1)) for k in d['timed_kernels']])
PY
cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o attn -- python3 $GRAFT_REPO_ROOT/tools/bench_attn.py --w -1 --iters 10 > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT && find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/attn_kernel_stats.csv && head -8 $OUT/attn_kernel_stats.csv | cut -c1-150
