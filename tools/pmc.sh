#!/bin/bash
# usage (on the GPU box): tools/pmc.sh <tag> "<counters group 1>" "<group 2>" ...  -> gpurun_out/<tag>/summary.txt (bench.py, 8 GiB decode)
# every pass runs under its own timeout (a counter group that the profiler cannot schedule must not eat the GPU budget)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift; mkdir -p $O; cd $R
for grp in "$@"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 5 ${PMC_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_$n -- python3 bench.py --steps 3 --warmup 1 --no-cpu $BENCH_ARGS > $O/pmc_$n.log 2>&1 || tail -3 $O/pmc_$n.log
done
python3 tools/prof_summary.py $O > $O/summary.txt 2>&1
grep -A40 "k_decode_blocks" $O/summary.txt | head -60
