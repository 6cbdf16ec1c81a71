#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
( time timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 ) 2>&1 | tail -8
bash tools/final_measure.sh r04 2>&1 | tail -14
timeout 600 python tools/low_entropy_bench.py 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_final/low_entropy.txt; cat gpurun_out/r04_final/low_entropy.txt | tail -9
