#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4pl
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r4pl/prof" -o s -- python3 "$GRAFT_REPO_ROOT/tools/split_bench.py" --subs 1024,1 > "$GRAFT_REPO_ROOT/gpurun_out/r4pl/prof.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r4pl/prof8" -o s -- python3 "$GRAFT_REPO_ROOT/tools/split_bench.py" --codec rle8_packed_multi --synth runs --size 67108864 --subs 1 > "$GRAFT_REPO_ROOT/gpurun_out/r4pl/prof8.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r4pl/prof*/**/s_kernel_stats.csv',recursive=True):
    print(f)
    for r in csv.DictReader(open(f)):
        if 'hsrle' in r['Name']: print(' ', r['Name'][:100], r['Calls'], r['AverageNs'])
PY
