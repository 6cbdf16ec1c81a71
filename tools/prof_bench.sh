#!/bin/bash
# usage (on the GPU box via gpurun): tools/prof_bench.sh <tag>   -> gpurun_out/<tag>/{bench.json, stats/, pmc_*/, summary.txt}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
python3 bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu > $O/stats.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_$n -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $O/pmc_$n.log 2>&1
done
python3 tools/prof_summary.py $O > $O/summary.txt 2>&1
cat $O/bench.json; head -12 $O/summary.txt
