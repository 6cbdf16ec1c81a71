#!/bin/bash
# usage (on the GPU box): tools/prof_script.sh <tag> <python script and args...>  -> rocprofv3 kernel stats of the script, head printed
R=$GRAFT_REPO_ROOT; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$tag; mkdir -p $O
( cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 "$@" > $O.log 2>&1 )
f=$(ls $O/*/*_kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && python3 $R/tools/show_stats.py $f ${ROWS:-10} || tail -5 $O.log
