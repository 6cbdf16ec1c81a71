#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 1500 python -X faulthandler -m pytest tests/test_gpu_parity.py -x -v 2>&1 | grep -v "^Extension modules" | grep -v "PASSED\|SKIPPED" | tail -60
