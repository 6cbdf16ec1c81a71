#!/bin/bash
# intermediate closing measurement of the build with the position-parallel encoder: tests touched this round, then tools/final_measure.sh
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist_nccl.py -x -q -m gpu -k "fuzzer or distributed_leg or graph or single_blocks or rle128_blocks or greedy_small" 2>&1 | tail -4
bash tools/final_measure.sh r05a
