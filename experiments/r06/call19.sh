#!/bin/bash
# round 6, call 19: kernel times of the slab-by-slab encode (call 18 measured whole-call rates only: the launch gaps dominate those)
mkdir -p gpurun_out/r06_c19
O=$GRAFT_REPO_ROOT/gpurun_out/r06_c19
cd /tmp && export TMPDIR=/tmp
for v in default slab8192 slab16384 slab32768; do
  if [ $v = default ]; then unset HSRLE_LIB; else export HSRLE_LIB=$GRAFT_REPO_ROOT/variants/libhsrle_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extras --steps 5 --warmup 2 > $O/bench_$v.log 2>&1
  f=$(find $O/prof_$v -name '*kernel_stats.csv' | head -1)
  head -10 "$f" | cut -c1-200 > $O/stats_$v.txt
  rm -rf $O/prof_$v
done
