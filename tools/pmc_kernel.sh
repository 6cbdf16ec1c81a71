#!/bin/bash
# usage (GPU box, via gpurun): bash tools/pmc_kernel.sh <tag> <kernel substring> -- <python script and args>
# PMC passes (one counter group per pass) of one kernel -> gpurun_out/pmc_<tag>/summary.txt
tag=$1; kern=$2; shift 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$tag; mkdir -p $O; cd $R
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_IFETCH"; do
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
    print(c, "avg per launch", sum(v) / len(v), "launches", len(v))
PY
cat $O/summary.txt
