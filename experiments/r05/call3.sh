#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
bash tools/prof_script.sh r05_pp_first tools/enc_time.py rle8_packed_multi 0 8
