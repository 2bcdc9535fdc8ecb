#!/bin/bash
# HBM traffic per launch of the hot kernels from two rocprofv3 --pmc passes over the bench command (FETCH_SIZE and
# WRITE_SIZE each in a run of its own, nothing else traced), summarised by tools/pmc_traffic.py.
# usage (on the GPU box): tools/collect_pmc.sh <git-head> <out-tag> [bench args...]
#   -> gpurun_out/<out-tag>_pmc_fetch.csv, _pmc_write.csv, _pmc_traffic.json   (copy into profiles/ to have bench.py use it)
set -e
HEAD=$1; TAG=$2; shift 2
export TMPDIR=/tmp
# a counter pass serialises every kernel and prints nothing until it ends: a line a minute shows that it is alive
( while true; do sleep 50; echo "[collect_pmc] $(date +%T) still collecting"; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
# one stream: bytes per launch do not depend on what runs beside a kernel, and the side stream's overlap probe (spin kernels
# on two streams, functional.streams_overlap) did not come back under counter collection (killed after 180 s of silence)
export VITK_SIDE_DW=0
for c in FETCH_SIZE WRITE_SIZE; do
    OUT=gpurun_out/pmc_${TAG}_$c
    rm -rf $OUT; mkdir -p $OUT
    timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d $OUT -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-isolated "$@" > $OUT/bench.log 2>&1
done
F=$(find gpurun_out/pmc_${TAG}_FETCH_SIZE -name '*counter_collection.csv' | head -1)
W=$(find gpurun_out/pmc_${TAG}_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python3 tools/pmc_traffic.py "$F" "$W" gpurun_out/${TAG}_pmc_traffic.json --head "$HEAD" --command "bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-isolated $*" \
    --per-kernel gpurun_out/${TAG}_pmc_per_kernel.csv
