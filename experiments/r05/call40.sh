#!/bin/bash
# round 5 call 40: the low-entropy split-phase helpers against the oracle's streams and the compiled reference's own functions
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_low_entropy_helpers.py tests/test_capi_symbols.py -x -q 2>&1 | tail -25
