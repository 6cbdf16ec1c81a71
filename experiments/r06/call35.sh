#!/bin/bash
# round 6, call 35: windowed encoders with blocks of 16 MiB (positions far beyond 16 bits)
mkdir -p gpurun_out/r06_c35
timeout 900 python -m pytest tests/test_gpu_pp.py -q -m gpu -k "many_mebibytes" -x 2>&1 | tail -12 | tee gpurun_out/r06_c35/log.txt
