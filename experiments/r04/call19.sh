#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 900 python -m pytest tests/test_gpu_mono.py -x -q -m gpu 2>&1 | tail -4
for g in 0 4096 2048 1024; do echo "== region $g"; timeout 300 python tools/mono_bench.py --cases packed8_runs_1g,lut8_runs_256m,lut64_video_88m --reps 4 --region $g 2>&1 | grep -v amdgpu.ids | tail -3; done
