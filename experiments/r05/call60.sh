#!/bin/bash
# round 5 call 60: the bench line with the video-shaped rows (extras.video_shaped)
cd /root/repo
( time timeout 900 python bench.py > gpurun_out/call60_bench.json 2> gpurun_out/call60_bench.err ) 2>&1 | tail -3
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/call60_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['traffic'])
v=d['extras']['video_shaped']; print({k:v[k] for k in v if k!='rows'})
for r in v.get('rows',[]): print(r)
PY
