#!/bin/bash
# first run of the position-parallel encoder: parity (new tests + the 8 bit rows of the old ones), then time at 8 GiB
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 900 python -m pytest tests/test_gpu_pp.py -x -q 2>&1 | tail -15
timeout 300 python tools/enc_time.py rle8_packed_multi 0 8 2>&1 | tail -1
timeout 300 python tools/enc_time.py rle8_multi 0 8 2>&1 | tail -1
timeout 300 python tools/enc_time.py rle8_packed_multi 1 8 2>&1 | tail -1
