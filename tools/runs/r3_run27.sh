#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 300 python -c "
import __graft_entry__ as g
g.smoke()
print('smoke ok')
" 2>&1 | tail -5
echo "== --gpus 2 on one GPU (expected: clean error)"
timeout -k 10 200 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-isolated 2>&1 | tail -6
echo "rc=$?"
