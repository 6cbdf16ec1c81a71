#!/bin/bash
# usage (on the GPU box via gpurun): tools/prof_bench.sh <tag>   -> gpurun_out/<tag>/{bench.json, stats/, pmc_*/, summary.txt}
# every profiler pass runs under its own timeout (a counter group the profiler cannot schedule must not eat the GPU budget)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
timeout -k 5 400 python3 bench.py > $O/bench.json 2> $O/bench.err
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu > $O/stats_bench.json 2> $O/stats.log
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_$n -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $O/pmc_$n.log 2>&1 || echo "pass failed: $grp"
done
python3 tools/prof_summary.py $O > $O/summary.txt 2>&1
cat $O/bench.json; cat $O/stats_bench.json; grep -A30 "k_decode_blocks" $O/summary.txt | head -50
