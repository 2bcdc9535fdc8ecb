#!/bin/bash
# round 3, GPU call 1: full GPU suite, then the headline bench under the schedules to compare.
set -e
O=gpurun_out/r3a
mkdir -p $O
rm -f $O/favor_errs.jsonl
VITK_REPORT_ERRORS=$O/favor_errs.jsonl timeout -k 10 420 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
b() { name=$1; shift; timeout -k 10 240 python bench.py --steps 20 --warmup 5 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; return 1; }
  python - $O/$name.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get('roofline') or {}
iso = (r.get('isolated') or {})
print(sys.argv[1], round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms  in-step frac', r.get('frac') and round(r['frac'], 4),
      'nt_us', r.get('avg_launch_us') and round(r['avg_launch_us'], 1), 'iso frac', iso.get('frac') and round(iso['frac'], 4),
      'all_gemm', r.get('all_gemm_tflops') and round(r['all_gemm_tflops'], 1), 'loss', round(d['loss'], 4), flush=True)
PY
}
b default
b nooverlap --no-max-overlap --no-cpu-baseline
VITK_SIDE_COLSUM=0 b splitk_only --no-cpu-baseline
VITK_DW_SPLITK= b colsum_only --no-cpu-baseline
b dropout01 --dropout 0.1 --no-cpu-baseline
b reducer_r0 --force-reducer --reserve-cus 0 --no-cpu-baseline
b reducer_r8 --force-reducer --reserve-cus 8 --no-cpu-baseline
b reducer_r16 --force-reducer --reserve-cus 16 --no-cpu-baseline
b reducer_r16_bf16 --force-reducer --reserve-cus 16 --comm-dtype bf16 --no-cpu-baseline
b default2 --no-cpu-baseline
set +e
timeout -k 10 120 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > $O/gpus2.out 2> $O/gpus2.err
echo "gpus2 rc=$?"; tail -3 $O/gpus2.err
