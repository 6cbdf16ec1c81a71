#!/bin/bash
# round 5 call 38: Short family (no list / one-symbol list) on the position-parallel encoder: parity
cd /root/repo
timeout 1700 python -m pytest tests/test_gpu_pp.py -x -q -m gpu -k "short or path_is" 2>&1 | tail -15
