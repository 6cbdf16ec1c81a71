#!/bin/bash
# round 6, call 41: differential stress of the final build (457c0ea725f74870), all 110 codecs, block sizes 128 B .. 64 KiB, 10 minutes
mkdir -p gpurun_out/r06_c41
timeout 800 python tools/gpu_stress.py 600 83 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/r06_c41/log.txt
