#!/bin/bash
# the round's closing measurement: full GPU suite, final_measure (bench line, kernel stats, PMC traffic, config 3, small containers), low-entropy bench, 110-codec sweep
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
( time timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 ) 2>&1 | tail -8
bash tools/final_measure.sh r04 2>&1 | tail -6
timeout 600 python tools/low_entropy_bench.py 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_final/low_entropy.txt; tail -3 gpurun_out/r04_final/low_entropy.txt
mkdir -p gpurun_out/r04_sweep
timeout 3000 python tools/sweep.py 8192 4096 video 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_sweep/sweep.md
tail -2 gpurun_out/r04_sweep/sweep.md
