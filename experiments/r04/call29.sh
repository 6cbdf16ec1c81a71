#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4pl3
( timeout 1500 python -m pytest tests/test_gpu_split.py tests/test_gpu_mono.py -x -q -k "not wave" 2>&1 | tail -3 )
timeout 300 python tools/split_bench.py --subs 1 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/split_bench.py --codec rle8_packed_multi --synth runs --size 67108864 --subs 1 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/split_bench.py --size 268435456 --subs 4096,1 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r4pl3/prof" -o s -- python3 "$GRAFT_REPO_ROOT/tools/split_bench.py" --subs 1 > "$GRAFT_REPO_ROOT/gpurun_out/r4pl3/prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r4pl3/prof*/**/s_kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'packets' in r['Name']: print(' ', r['Name'][:100], r['Calls'], r['AverageNs'])
PY
# first-codec effect of the list / ring choice: per-launch encode times of the first codec of a process
python - <<'PY'
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import torch, hsrle
size=4<<30; bs=4096
dst = torch.empty(hsrle.container_bound(size, bs), dtype=torch.uint8, device='cuda'); ws = torch.empty(hsrle.workspace_size(size, bs), dtype=torch.uint8, device='cuda')
for k in ('rle24_byte','rle24_sym','rle24_byte'):
    src = hsrle.synth(0, 3, 5, size)
    ts=[]
    for i in range(8):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(); hsrle.compress_async(k, src, dst, bs, workspace=ws); e1.record(); torch.cuda.synchronize()
        ts.append(round(e0.elapsed_time(e1),3))
    print(k, ts)
PY
