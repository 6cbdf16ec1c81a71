#!/bin/bash
# round 5 call 36: async mono decode tests + the mono / big / graph suites
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_mono_async.py tests/test_gpu_mono.py tests/test_gpu_big.py -q -m gpu -k "not fuzzer" 2>&1 | tail -15
