#!/bin/bash
# emission v2: headers and literal pieces OR-ed into a zeroed LDS image by the packet lanes, image copied out once
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 900 python -m pytest tests/test_gpu_pp.py -x -q 2>&1 | tail -5
timeout 300 python tools/enc_time.py rle8_packed_multi 0 8 2>&1 | tail -1
timeout 300 python tools/enc_time.py rle8_packed_multi 1 8 2>&1 | tail -1
timeout 300 python tools/enc_time.py rle8_multi 0 8 2>&1 | tail -1
bash tools/prof_script.sh r05_pp_v2 tools/enc_time.py rle8_packed_multi 0 8 | head -4
bash tools/pmc_kernel2.sh pp_v2 "k_encode8_pp<1, 1" -- tools/enc_time.py rle8_packed_multi 0 8 2>&1 | grep -E "INSTS|WAVE_CYCLES|WAIT|ACTIVE|THREAD|FETCH|WRITE_SIZE|BANK"
