#!/bin/bash
# instruction counts and waits of the north-star decode kernel on the headline buffer (8 GiB rle8_packed_multi)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
bash tools/pmc_kernel2.sh dec_r05 "k_decode_blocks<1, 1, 0" -- tools/probe_profile_run.py 8192 4096 rle8_packed_multi 3 2>&1 | grep -E "INSTS|WAVE_CYCLES|WAIT|ACTIVE|THREAD|BUSY|SQ_WAVES|BANK|IDX"
