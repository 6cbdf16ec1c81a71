#!/bin/bash
# usage (on the GPU box): tools/prof_stats.sh <tag> [n]  -> n kernel-trace runs of the default bench; prints the decode kernel's stats line of each
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for i in $(seq 1 ${2:-3}); do
  O=$R/gpurun_out/$1/stats$i; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --no-cpu > $O.log 2>&1
  grep "k_decode_blocks" $O/*/*_kernel_stats.csv | cut -d, -f2- | awk -F, '{printf "run '$i': calls %s avg %.4f ms min %.4f max %.4f | ", $(NF-6), $(NF-4)/1e6, $(NF-2)/1e6, $(NF-1)/1e6}'
  grep -o '"kernel_ms": [0-9.]*' $O.log
done
