#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r04_again
timeout 900 python bench.py 2> gpurun_out/r04_again/bench.err | tail -1 > gpurun_out/r04_again/bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_again/bench.json').read())
print('decode', d['value'], d['ms_per_step'], d['roofline']['frac'], 'traffic', d['roofline']['traffic'], '| encode', d['encode']['value'], d['encode']['ms'], d['encode']['roofline']['frac'])
e=d['extras']; print({k:e['config3_frame'][k] for k in ('encode_us','split_decode_us')}, e['mono_1GiB']['decode_ms'], e['mono_1GiB']['encode_ms'])
PY
