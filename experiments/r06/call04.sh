#!/bin/bash
# round 6, call 4: the whole GPU suite on the build with all 50 north-star encoders position-parallel
mkdir -p gpurun_out/r06_c04
python -m pytest tests -m gpu -q -x > gpurun_out/r06_c04/gpu_suite.log 2>&1; echo "suite rc=$?"
tail -8 gpurun_out/r06_c04/gpu_suite.log
