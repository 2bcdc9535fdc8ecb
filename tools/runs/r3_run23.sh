#!/bin/bash
set -e
export TMPDIR=/tmp
OUT=gpurun_out/r3w
mkdir -p $OUT
echo "== full"; timeout -k 10 120 python tools/bench_attn.py --w -1 --iters 10
for k in 1 2 4 8 15; do
  echo "== stop $k"; VITK_LIB=tools/_lai/libvitk_stop$k.so timeout -k 10 120 python tools/bench_attn.py --w -1 --iters 10
done
cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o attn -- python3 $GRAFT_REPO_ROOT/tools/bench_attn.py --w -1 --iters 10 > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT && find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/attn_kernel_stats.csv && head -8 $OUT/attn_kernel_stats.csv | cut -c1-160
