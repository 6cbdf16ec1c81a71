#!/bin/bash
# round 6, call 33: the closing measurement of the build with the windowed encoders -- PMC traffic, bench line (with extras.large_blocks), rocprofv3 kernel stats, config 3,
# small containers (tools/final_measure.sh), the 220-row sweep at 4 KiB blocks, and the 110 codecs at 64 KiB blocks (run-distributed)
bash tools/final_measure.sh r06
timeout 2400 python tools/sweep.py 8192 4096 video > gpurun_out/r06_final/codec_sweep_8GiB.md 2> gpurun_out/r06_final/sweep.err; echo "sweep rc=$?"
tail -3 gpurun_out/r06_final/codec_sweep_8GiB.md
timeout 1500 python tools/sweep.py 8192 65536 > gpurun_out/r06_final/codec_sweep_8GiB_64KiB_blocks.md 2> gpurun_out/r06_final/sweep64.err; echo "sweep64 rc=$?"
tail -3 gpurun_out/r06_final/codec_sweep_8GiB_64KiB_blocks.md
