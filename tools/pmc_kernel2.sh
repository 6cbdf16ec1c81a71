#!/bin/bash
# usage (GPU box, via gpurun): bash tools/pmc_kernel2.sh <tag> <kernel substring> -- <python script and args>
# like pmc_kernel.sh, plus the HBM traffic counters (FETCH_SIZE / WRITE_SIZE in KiB, request counts); HSRLE_LIB is passed through
tag=$1; kern=$2; shift 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$tag; mkdir -p $O; cd $R
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_IFETCH" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/$n -o p -- python3 "$@" > $O/$n.log 2>&1 || echo "pass failed: $grp"
done
python3 - <<PY > $O/summary.txt
import csv, glob, collections
vals = collections.defaultdict(list)
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$kern" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(vals.items()):
    print("%-28s avg per launch %16.1f  launches %d" % (c, sum(v) / len(v), len(v)))
PY
cat $O/summary.txt
