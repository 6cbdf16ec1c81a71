#!/bin/bash
# single pass with look-back groups kept on one XCD (group g on XCD g % 8) and longer sleeps between polls
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 600 python -m pytest tests/test_gpu_pp.py -x -q 2>&1 | tail -3
timeout 300 python tools/enc_time.py rle8_packed_multi 0 8 2>&1 | tail -1
timeout 300 python tools/enc_time.py rle8_packed_multi 1 8 2>&1 | tail -1
HSRLE_LIB=$PWD/variants/libhsrle_st1.so timeout 300 python tools/probe_pp_stamps.py rle8_packed_multi 0 2>&1 | grep -E "call|copy out|look-back"
