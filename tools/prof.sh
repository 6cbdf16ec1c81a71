#!/bin/bash
# usage: tools/prof.sh <outdir-name> [probe args...]   (run on the GPU box via gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/probe_profile_run.py "$@" > $O/trace.log 2>&1
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT" "GRBM_GUI_ACTIVE"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_$n -- python3 tools/probe_profile_run.py "$@" > $O/pmc_$n.log 2>&1
done
python3 tools/prof_summary.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
