#!/bin/bash
# the whole GPU suite after the memset -> kernel change and the cut finder's straight-line path; times of what the finder feeds
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
( time timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^Extension modules" | tail -4 ) 2>&1 | tail -8
for k in rle128_sym rle8_single; do timeout 120 python tools/frame_enc_time.py $k 2>&1 | grep -v amdgpu.ids; done
