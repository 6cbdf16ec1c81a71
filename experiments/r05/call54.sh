#!/bin/bash
# round 5 call 54: full GPU suite, then the closing measurement of this build (tools/final_measure.sh r05) and the 110-codec sweep
cd /root/repo
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -4
bash tools/final_measure.sh r05
timeout 2400 python tools/sweep.py 8192 4096 video > gpurun_out/r05_final/codec_sweep_8GiB.md 2> gpurun_out/r05_final/sweep.err
tail -2 gpurun_out/r05_final/codec_sweep_8GiB.md
