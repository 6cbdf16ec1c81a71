#!/bin/bash
# usage (GPU box, via gpurun): bash tools/traffic_r02.sh -> gpurun_out/traffic_r02/: the decoder's request statistics (diagnostic build) and the
# PMC passes of the SHIPPED build on the headline workload (one counter group per pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/traffic_r02; mkdir -p $O; cd $R
HSRLE_LIB=$R/variants/libhsrle_reqstats.so python3 tools/req_stats.py > $O/req_stats.json 2> $O/req_stats.err; cat $O/req_stats.json
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-30)
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_$n -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $O/pmc_$n.log 2>&1 || echo "pass failed: $grp"
done
python3 - <<PY
import csv, glob, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "decode" if "k_decode_blocks" in k else ("encode8" if "k_encode8_blocks" in k else ("compact" if "k_compact" in k else None))
        if name: vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, d in vals.items():
    for c, v in sorted(d.items()):
        print(name, c, "avg per launch", sum(v) / len(v), "launches", len(v))
PY
