#!/bin/bash
# where does the position-parallel encoder take over?  container sizes 1 MiB .. 1 GiB, 4 KiB blocks, against the ring / run list encoders
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
for pp in 1 2; do HSRLE_PP=$pp HSRLE_LIB=$PWD/variants/libhsrle_exp.so timeout 600 python tools/pp_threshold.py; done
