#!/bin/bash
# packet-list decode of small containers: parity tests + the config-3 frame with every split mode, kernel trace
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4pl
( timeout 900 python -m pytest tests/test_gpu_split.py -x -q -k "not wave" 2>&1 | tail -6 ) 
timeout 300 python tools/split_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4pl/split.txt
timeout 300 python tools/split_bench.py --codec rle8_packed_multi --synth runs --size 67108864 --subs 4096,1024,1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4pl/split.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r4pl/prof" -o s -- python3 "$GRAFT_REPO_ROOT/tools/split_bench.py" --subs 1024,1 > "$GRAFT_REPO_ROOT/gpurun_out/r4pl/prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r4pl/prof/**/s_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'hsrle' in r['Name']: print(r['Name'][:100], r['Calls'], r['AverageNs'])
PY
