#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
( timeout 2400 python -m pytest tests/test_gpu_mono.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -4 )
timeout 300 python tools/mono_bench.py --reps 5 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tail -8
