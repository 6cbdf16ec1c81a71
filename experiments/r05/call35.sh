#!/bin/bash
# round 5 call 35: kernel split of the monolithic decode, 1 GiB rle8_packed_multi (walk / resolve / records / decode)
cd /root/repo
ROWS=14 bash tools/prof_script.sh mono_dec_split tools/mono_bench.py --cases packed8_runs_1g --reps 5
tail -3 gpurun_out/mono_dec_split.log
