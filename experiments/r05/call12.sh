#!/bin/bash
# resident workgroups: is the grid what the occupancy query says it should be?  grid = 50 .. 800 % of (query x CUs), without (q1) and with (q2) stores
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
for a in q1 q2; do for pct in 50 100 200 400 1600; do echo "== $a grid $pct %"; HSRLE_PP_GRID_PCT=$pct HSRLE_LIB=$PWD/variants/libhsrle_$a.so timeout 300 python tools/enc_time.py rle8_packed_multi 0 8 2>&1 | tail -1; done; done
