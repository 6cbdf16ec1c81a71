#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -v -k "single or 128" 2>&1 | grep -E "PASSED|FAILED|ERROR|Fatal|fault|Abort|core|test_" | tail -25
