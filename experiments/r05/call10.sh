#!/bin/bash
# what does the emission pass wait for?  timing-only ablations of the copy-out: no stores / plain stores / no edge pieces
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
for a in 1 2 3; do echo "== ablate $a"; HSRLE_LIB=$PWD/variants/libhsrle_ab$a.so bash tools/prof_script.sh r05_pp_ab$a tools/enc_time.py rle8_packed_multi 0 8 | grep "pp<1, 1"; done
echo "== default"; bash tools/prof_script.sh r05_pp_v3b tools/enc_time.py rle8_packed_multi 0 8 | grep "pp<1, [01]"
