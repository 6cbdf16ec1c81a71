#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r04_sweep
timeout 3000 python tools/sweep.py 8192 4096 video 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_sweep/sweep.md
tail -3 gpurun_out/r04_sweep/sweep.md
