#!/bin/bash
# SQ counters of the kernels of one command, per kernel (averages over its dispatches).
# usage (on the GPU box): tools/pmc_kernel.sh <tag> <counters, space separated, <= 8 SQ> -- python3 script.py args...
set -e
TAG=$1; shift
CTRS=()
while [ "$1" != "--" ]; do CTRS+=("$1"); shift; done
shift
OUT=gpurun_out/pmck_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "${CTRS[@]}" --output-format csv -d $OUT -o run -- "$@" > $OUT/log.txt 2>&1
F=$(find $OUT -name '*counter_collection.csv' | head -1)
python3 - "$F" <<'PY'
import csv, sys, re
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r'\(.*', '', r['Kernel_Name'])
    tot[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[(k, r['Counter_Name'])] += 1
for k in tot:
    if 'vitk' not in k: continue
    print(k[:90])
    for c, v in tot[k].items():
        print(f'    {c:28s} {v / cnt[(k, c)]:16.0f}')
PY
