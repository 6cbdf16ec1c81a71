#!/bin/bash
# emission pass as resident workgroups with the next blocks' input prefetched into registers: parity, prefetch depth 0 .. 3
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 900 python -m pytest tests/test_gpu_pp.py -x -q 2>&1 | tail -3
for pf in 0 1 3; do HSRLE_LIB=$PWD/variants/libhsrle_pf$pf.so timeout 300 python tools/enc_time.py rle8_packed_multi 0 8 2>&1 | tail -1; done
timeout 300 python tools/enc_time.py rle8_packed_multi 0 8 2>&1 | tail -1
timeout 300 python tools/enc_time.py rle8_packed_multi 1 8 2>&1 | tail -1
bash tools/prof_script.sh r05_pp_v4 tools/enc_time.py rle8_packed_multi 0 8 | head -4
HSRLE_LIB=$PWD/variants/libhsrle_ppst.so timeout 300 python tools/probe_pp_stamps.py rle8_packed_multi 0
