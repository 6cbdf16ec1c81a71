#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 3000 python -m pytest tests -q -m gpu --deselect "tests/test_gpu_parity.py::test_reference_fuzzer_runs_on_the_gpu_library" 2>&1 | tail -6
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k fuzzer 2>&1 | tail -3
