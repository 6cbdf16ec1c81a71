#!/bin/bash
# round 6, call 9: the closing measurement of this build -- PMC traffic, bench line, rocprofv3 kernel stats, config 3, small containers (tools/final_measure.sh), then the 220-row sweep
bash tools/final_measure.sh r06
timeout 2400 python tools/sweep.py 8192 4096 video > gpurun_out/r06_final/codec_sweep_8GiB.md 2> gpurun_out/r06_final/sweep.err; echo "sweep rc=$?"
tail -3 gpurun_out/r06_final/codec_sweep_8GiB.md
