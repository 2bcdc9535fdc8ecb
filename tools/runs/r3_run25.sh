#!/bin/bash
# round 3, artifact pass on the tree with the unmasked per-head kernels: GPU suite, PMC traffic, default line, kernel stats,
# every workload, dropout row, per-kernel traffic + kernel stats of the unmasked config
set -e
export TMPDIR=/tmp
T=r3x
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/${T}_pytest.log 2>&1 || { tail -30 gpurun_out/${T}_pytest.log; exit 1; }
tail -3 gpurun_out/${T}_pytest.log
tools/collect_pmc.sh $1 ${T}_headline
cp gpurun_out/${T}_headline_pmc_traffic.json profiles/pmc_gemm_traffic.json
timeout -k 10 400 python bench.py > gpurun_out/${T}_default.json 2> gpurun_out/${T}_default.err || { tail -5 gpurun_out/${T}_default.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/${T}_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print('default', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms frac', round(r['frac'],3), 'traffic', r.get('traffic'), 'iso', round(r['isolated']['frac'],3), 'cpu', d['cpu_baseline']['value'])"
tools/prof_step.sh ${T}_headline
tools/bench_all.sh
timeout -k 10 300 python bench.py --dropout 0.1 --no-cpu-baseline --no-isolated --steps 10 --warmup 3 > gpurun_out/${T}_dropout01.json 2> gpurun_out/${T}_dropout01.err
tools/prof_step.sh ${T}_dis_b16 --workload dis_b16
tools/collect_pmc.sh $1 ${T}_dis_b16 --workload dis_b16
