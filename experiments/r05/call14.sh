#!/bin/bash
# instruction counts of the sizes pass (MODE 0) and the emission pass (MODE 1) of the current build
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
bash tools/pmc_kernel2.sh pp_m0 "k_encode8_pp<1, 0" -- tools/enc_time.py rle8_packed_multi 0 8 2>&1 | grep -E "INSTS|WAVE_CYCLES|WAIT|ACTIVE|THREAD|BUSY"
python3 - <<'PY'
import csv, glob, collections
vals = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_pp_m0/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_encode8_pp<1, 1" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("---- MODE 1")
for c, v in sorted(vals.items()):
    if any(k in c for k in ("INSTS", "WAVE_CYCLES", "WAIT", "ACTIVE", "THREAD", "BUSY")):
        print("%-28s avg per launch %16.1f  launches %d" % (c, sum(v) / len(v), len(v)))
PY
