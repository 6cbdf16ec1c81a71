#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 1500 python -m pytest tests/test_gpu_pp.py -x -q 2>&1 | tail -4
for c in rle16_byte_packed rle64_byte_packed rle24_sym rle48_sym_packed; do for k in 0 1; do timeout 300 python tools/enc_time.py $c $k 8 2>&1 | tail -1; done; done
