#!/bin/bash
# headline encoder with a 128-byte window per lane and step (two words of match bits, one run-end loop): parity, then same-box A/B
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
echo "== default build (window 64, refactored scan)"; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "rle8_ and (blocks_bit_exact or small_containers or dropin or long_literal)" 2>&1 | tail -2
export HSRLE_LIB=$PWD/variants/libhsrle_q128.so
echo "== window 128"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mono.py -x -q -k "rle8_" 2>&1 | tail -2
unset HSRLE_LIB
REPS=2 bash tools/ab.sh q128 2>&1 | grep -E "^==|dec "
