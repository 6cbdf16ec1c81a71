#!/bin/bash
# round 5 call 50: the 110-codec sweep at 8 GiB with the capped decoder rounds
cd /root/repo
timeout 2400 python tools/sweep.py 8192 4096 video > gpurun_out/sweep_cap.md 2> gpurun_out/sweep_cap.err
tail -2 gpurun_out/sweep_cap.md
