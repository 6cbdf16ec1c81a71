cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r8m; mkdir -p $O; cd $R
for k in 0 1; do timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k$k -- python3 tools/rle8m_bench.py 1024 4096 $k > $O/bench$k.txt 2> $O/err$k.txt; tail -1 $O/bench$k.txt; f=$(find $O/k$k -name "*kernel_stats.csv" | head -1); grep -i "rle8m" $f; done
