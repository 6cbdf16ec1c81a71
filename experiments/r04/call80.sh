#!/bin/bash
# where the 8 GiB encode of the Single codecs goes (pick / encode / compaction), both data kinds
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4sb
cd /tmp && export TMPDIR=/tmp
for kind in 0 1; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r4sb/k$kind" -o f -- python3 "$R/tools/enc_time.py" rle8_single $kind 8 > "$R/gpurun_out/r4sb/k$kind.log" 2>&1
tail -1 "$R/gpurun_out/r4sb/k$kind.log"
python3 - "$R/gpurun_out/r4sb/k$kind" <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/**/f_kernel_stats.csv',recursive=True):
    rows=sorted(csv.DictReader(open(f)), key=lambda r:-float(r['TotalDurationNs']))
    for r in rows[:7]: print('   %-80s calls %4s avg %10.1f us'%(r['Name'][:80], r['Calls'], float(r['AverageNs'])/1000))
PY
done
