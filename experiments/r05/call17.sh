#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
HSRLE_LIB=$PWD/variants/libhsrle_st1.so timeout 300 python tools/probe_pp_stamps.py rle8_packed_multi 0 2>&1 | tail -8
