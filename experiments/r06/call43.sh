#!/bin/bash
# round 6, call 43: differential stress of the final build (e39f25c2323390f1), all 110 codecs, block sizes 128 B .. 64 KiB, 10 minutes
mkdir -p gpurun_out/r06_c43
timeout 800 python tools/gpu_stress.py 600 97 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/r06_c43/log.txt
