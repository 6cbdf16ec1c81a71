#!/bin/bash
# emission v3: literals and run symbols from an LDS copy of the block's input (no L2 gathers in the chain)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 900 python -m pytest tests/test_gpu_pp.py -x -q 2>&1 | tail -3
timeout 300 python tools/enc_time.py rle8_packed_multi 0 8 2>&1 | tail -1
timeout 300 python tools/enc_time.py rle8_packed_multi 1 8 2>&1 | tail -1
timeout 300 python tools/enc_time.py rle8_multi 0 8 2>&1 | tail -1
bash tools/prof_script.sh r05_pp_v3 tools/enc_time.py rle8_packed_multi 0 8 | head -4
HSRLE_LIB=$PWD/variants/libhsrle_ppst.so timeout 300 python tools/probe_pp_stamps.py rle8_packed_multi 0
