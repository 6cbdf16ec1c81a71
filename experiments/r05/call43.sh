#!/bin/bash
# round 5 call 43: async mono tests again, then the closing measurement of this build (tools/final_measure.sh r05) and the 110-codec sweep
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_mono_async.py -q -m gpu 2>&1 | tail -3
bash tools/final_measure.sh r05
timeout 2400 python tools/sweep.py 8192 4096 video > gpurun_out/r05_final/codec_sweep_8GiB.md 2> gpurun_out/r05_final/sweep.err
tail -3 gpurun_out/r05_final/codec_sweep_8GiB.md
