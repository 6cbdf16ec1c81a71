#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 900 python -X faulthandler -m pytest tests/test_gpu_parity.py -x -q -k "need_no_alignment" 2>&1 | grep -v "^Extension modules" | tail -25 | cut -c1-300
