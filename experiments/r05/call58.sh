#!/bin/bash
# round 5 call 58: differential stress of the final kernels against the oracle (capped decoders, position-parallel Short encoders): three seeds x 5 minutes
cd /root/repo
for seed in 51 52 53; do timeout 400 python tools/gpu_stress.py 300 $seed 2>&1 | grep -v amdgpu | tail -3; done
