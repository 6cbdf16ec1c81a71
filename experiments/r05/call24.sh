#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzzer" 2>&1 | tail -40
