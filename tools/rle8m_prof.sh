#!/bin/bash
# usage (on the GPU box via gpurun): bash tools/rle8m_prof.sh [section_bytes ...]  -> gpurun_out/r8m/: bench + rocprofv3 kernel stats per kind
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r8m; mkdir -p $O; cd $R
for sec in ${@:-4096}; do for k in 0 1; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s${sec}k$k -- python3 tools/rle8m_bench.py 1024 $sec $k > $O/bench_s${sec}k$k.txt 2> $O/err_s${sec}k$k.txt
  tail -1 $O/bench_s${sec}k$k.txt; f=$(find $O/s${sec}k$k -name "*kernel_stats.csv" | head -1); grep -i "rle8m" $f
done; done
