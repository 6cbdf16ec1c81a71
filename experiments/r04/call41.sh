#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4single
cat > /tmp/one.py <<'PY'
import sys; sys.path.insert(0,sys.argv[1]+'/tests'); sys.path.insert(0,sys.argv[1]+'/hypersonic-rle-kit_amd/python')
import torch, hsrle
size=4<<30; bs=4096
dst = torch.empty(hsrle.container_bound(size, bs), dtype=torch.uint8, device='cuda'); ws = torch.empty(hsrle.workspace_size(size, bs), dtype=torch.uint8, device='cuda')
for k,S in (('rle8_single',1),('rle128_byte_packed',16),('rle32_3symlut_byte_short_greedy',4)):
  for kind in (0,1):
    src = hsrle.synth(kind, S, 5, size)
    for i in range(3): hsrle.compress_async(k, src, dst, bs, workspace=ws)
    torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r4single/p" -o s -- python3 /tmp/one.py "$R" > "$R/gpurun_out/r4single/log.txt" 2>&1
cd "$R"
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r4single/p/**/s_kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'hsrle' in r['Name'] and 'synth' not in r['Name']: print(' ', r['Name'][:120], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
