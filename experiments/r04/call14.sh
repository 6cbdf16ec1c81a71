#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "low_entropy or rle8m" 2>&1 | tail -15
