#!/bin/bash
# resident workgroups WITHOUT payload stores (timing only): is the input load the wave's life, or the wait for its stores?  prefetch 2 / 0 / 4
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
for a in p1 p2 p3; do echo "== $a"; HSRLE_LIB=$PWD/variants/libhsrle_$a.so bash tools/prof_script.sh r05_pp_$a tools/enc_time.py rle8_packed_multi 0 8 | grep "pp<1, 1"; done
