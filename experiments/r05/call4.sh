#!/bin/bash
# emission with G chunks per lane in flight: parity of the default (G = 3), then encode time for G = 1, 2, 3, 4
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 900 python -m pytest tests/test_gpu_pp.py -x -q 2>&1 | tail -3
for g in 1 2 4; do HSRLE_LIB=$PWD/variants/libhsrle_g$g.so timeout 300 python tools/enc_time.py rle8_packed_multi 0 8 2>&1 | tail -1; done
timeout 300 python tools/enc_time.py rle8_packed_multi 0 8 2>&1 | tail -1
timeout 300 python tools/enc_time.py rle8_packed_multi 1 8 2>&1 | tail -1
bash tools/prof_script.sh r05_pp_g3 tools/enc_time.py rle8_packed_multi 0 8
