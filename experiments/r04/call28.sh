#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4pl2
timeout 900 python tools/ab_codecs.py 4096 rle24_byte,rle24_sym_packed,rle24_sym,rle32_sym_packed 0 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r4pl2/prof" -o s -- python3 "$GRAFT_REPO_ROOT/tools/split_bench.py" --subs 1 > "$GRAFT_REPO_ROOT/gpurun_out/r4pl2/prof.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r4pl2/prof8" -o s -- python3 "$GRAFT_REPO_ROOT/tools/split_bench.py" --codec rle8_packed_multi --synth runs --size 67108864 --subs 1 > "$GRAFT_REPO_ROOT/gpurun_out/r4pl2/prof8.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r4pl2/prof*/**/s_kernel_stats.csv',recursive=True):
    print(f)
    for r in csv.DictReader(open(f)):
        if 'packets' in r['Name']: print(' ', r['Name'][:100], r['Calls'], r['AverageNs'])
PY
