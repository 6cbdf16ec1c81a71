#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
for a in st1 st2; do echo "== $a"; HSRLE_LIB=$PWD/variants/libhsrle_$a.so timeout 300 python tools/probe_pp_stamps.py rle8_packed_multi 0; done
