#!/bin/bash
# round 6, call 18: the position-parallel encode slab by slab (HSRLE_PP_SLAB_BLOCKS: pass 1, scan, pass 2 per slab of 32 / 64 / 128 MiB on one stream) --
# does the second read of the input come from the memory-side cache?
mkdir -p gpurun_out/r06_c18
REPS=2 bash tools/ab.sh slab8192 slab16384 slab32768 2>&1 | tee gpurun_out/r06_c18/ab.log
cd /tmp && export TMPDIR=/tmp
for v in slab8192 slab16384; do
  HSRLE_LIB=$GRAFT_REPO_ROOT/variants/libhsrle_$v.so rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06_c18/prof_$v -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r06_c18/bench_$v.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/r06_c18/prof_$v -name '*kernel_stats.csv' | head -1)
  head -12 "$f" | cut -c1-220 > $GRAFT_REPO_ROOT/gpurun_out/r06_c18/stats_$v.txt
  find $GRAFT_REPO_ROOT/gpurun_out/r06_c18/prof_$v -type f ! -name '*kernel_stats.csv' -delete
done
