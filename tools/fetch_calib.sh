#!/bin/bash
# usage (GPU box, via gpurun): bash tools/fetch_calib.sh  -> gpurun_out/fetch_calib/summary.txt
# raw FETCH_SIZE (and its request counters) of the replayed decoder read mix over a buffer of known size (tools/ubench/fetch_calib.hip)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fetch_calib; mkdir -p $O; cd $R
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $O/fetch_calib tools/ubench/fetch_calib.hip || exit 1
for mode in 0 1 2 3; do
  $O/fetch_calib $mode 4096 > $O/plain_$mode.txt
  for grp in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum"; do
    n=$(echo $grp | tr ' ' '_' | cut -c1-30)
    timeout -k 5 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/m${mode}_$n -o p -- $O/fetch_calib $mode 4096 > $O/m${mode}_$n.log 2>&1 || echo "pass failed: $mode $grp"
  done
done
python3 tools/fetch_calib_summary.py $O > $O/summary.txt
cat $O/summary.txt
