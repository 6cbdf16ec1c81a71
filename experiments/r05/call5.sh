#!/bin/bash
# why is the emission pass 4.6 - 5.3 ms?  counters of k_encode8_pp<PACKED, 1> (G = 3) and encode time for G = 1, 2
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
for g in 1 2; do HSRLE_LIB=$PWD/variants/libhsrle_g$g.so timeout 300 python tools/enc_time.py rle8_packed_multi 0 8 2>&1 | tail -1; done
bash tools/pmc_kernel2.sh pp_g3 "k_encode8_pp<1, 1" -- tools/enc_time.py rle8_packed_multi 0 8
