#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4e
( time timeout 900 python bench.py > gpurun_out/r4e/bench.json 2> gpurun_out/r4e/bench.err ) 2>&1 | tail -3
tail -c 600 gpurun_out/r4e/bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r4e/bench.json').read())
print('dec', j['ms_per_step'], j['roofline']['frac'], 'enc', j['encode']['ms'], j['encode']['roofline']['frac'], 'ok', j['bit_exact'])
print('cpu', j.get('cpu_baseline',{}).get('value'), j['encode'].get('cpu_baseline',{}).get('value'))
e=j.get('extras',{})
print({k:v for k,v in e.items() if k!='config5'})
c5=e.get('config5',{})
print('config5', {k:v for k,v in c5.items() if k!='rows'})
for r in c5.get('rows',[]): print(r)
PY
timeout 900 python -m pytest tests/test_gpu_dist_nccl.py -x -q 2>&1 | tail -5
