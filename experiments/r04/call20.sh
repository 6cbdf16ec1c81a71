#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 900 python -m pytest tests/test_gpu_mono.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python tools/mono_bench.py --reps 4 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tail -8
