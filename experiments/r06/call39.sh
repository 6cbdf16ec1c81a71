#!/bin/bash
# round 6, call 39: the final build (windowed encoders for 86 codecs): the whole GPU suite, then -- if it is green -- the closing measurement (call 33's recipe)
mkdir -p gpurun_out/r06_c39
python -m pytest tests -m gpu -q -x > gpurun_out/r06_c39/gpu_suite.log 2>&1; rc=$?; echo "suite rc=$rc"
tail -4 gpurun_out/r06_c39/gpu_suite.log
[ $rc -eq 0 ] || exit 1
bash experiments/r06/call33.sh
