#!/bin/bash
# round 5 pricing: the headline encoder with header assembly + literal emission compiled out (decisions and sizes only),
# with and without one 8-byte record store per packet, at both ring sizes (a decide-only kernel needs no history: 128-byte ring = 16 waves per CU)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
for lib in exp d1 d2; do
  for ring in 256 128; do
    echo "== $lib ring $ring"
    HSRLE_LIB=$PWD/variants/libhsrle_$lib.so HSRLE_ENC_RING=$ring timeout 300 python tools/enc_time.py rle8_packed_multi 0 8 2>&1 | tail -1
  done
done
echo "== kernel trace d1 ring 128"
cd /tmp && export TMPDIR=/tmp
HSRLE_LIB=$GRAFT_REPO_ROOT/variants/libhsrle_d1.so HSRLE_ENC_RING=128 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_d1 -o d1 -- python3 $GRAFT_REPO_ROOT/tools/enc_time.py rle8_packed_multi 0 8 > /tmp/d1.log 2>&1
find /tmp/prof_d1 -name "*kernel_stats*" | head -1 | xargs -r head -8
HSRLE_LIB=$GRAFT_REPO_ROOT/variants/libhsrle_exp.so HSRLE_ENC_RING=256 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_exp -o exp -- python3 $GRAFT_REPO_ROOT/tools/enc_time.py rle8_packed_multi 0 8 > /tmp/exp.log 2>&1
find /tmp/prof_exp -name "*kernel_stats*" | head -1 | xargs -r head -8
