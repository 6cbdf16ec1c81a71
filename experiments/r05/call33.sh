#!/bin/bash
# round 5 call 33: gated asynchronous monolithic decode: tests + bench extras
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_mono_async.py tests/test_gpu_mono.py tests/test_gpu_big.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/call33_tests.log
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/call33_bench.json 2> gpurun_out/call33_bench.err
tail -3 gpurun_out/call33_tests.log
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/call33_bench.json').read().strip().splitlines()[-1])
print(json.dumps(d['extras']['mono_1GiB'],indent=1))
print(d['value'], d['roofline'])
PY
