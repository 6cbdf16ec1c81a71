#!/bin/bash
# usage (GPU box, via gpurun): bash tools/prof_mono.sh CASE [CASE ...]   -> gpurun_out/prof_mono/<case>/: rocprofv3 kernel stats of tools/mono_bench.py
for c in "$@"; do
  O=gpurun_out/prof_mono/$c; mkdir -p $O
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o mono -- python3 tools/mono_bench.py --cases $c --reps 3 $MONO_ARGS > $O/bench.txt 2>&1
  tail -1 $O/bench.txt
done
