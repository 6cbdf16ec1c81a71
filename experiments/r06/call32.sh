#!/bin/bash
# round 6, call 32: the whole GPU suite on the build with the windowed encoders
mkdir -p gpurun_out/r06_c32
python -m pytest tests -m gpu -q -x > gpurun_out/r06_c32/gpu_suite.log 2>&1; echo "suite rc=$?"
tail -8 gpurun_out/r06_c32/gpu_suite.log
