#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
( timeout 600 python -m pytest tests/test_gpu_split.py -x -q -k "aligned or range or more_packets" 2>&1 | tail -8 )
