#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
ROWS=14 bash tools/prof_script.sh r05_mono_pp tools/mono_enc_bench.py rle8_packed_multi 1
