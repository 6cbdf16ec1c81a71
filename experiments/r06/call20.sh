#!/bin/bash
# round 6, call 20: kernel times of the monolithic encode / decode of the 1 GiB rle8_packed stream (where do 1.20 / 1.32 ms go?)
O=$GRAFT_REPO_ROOT/gpurun_out/r06_c20; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -o e -- python3 $GRAFT_REPO_ROOT/tools/mono_enc_bench.py rle8_packed_multi 1 > $O/enc.log 2>&1
head -16 $(find $O/enc -name '*kernel_stats.csv' | head -1) | cut -c1-230 > $O/enc_stats.txt
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dec -o d -- python3 tools/mono_bench.py --cases rle8_packed_multi --reps 3 > $O/dec.log 2>&1
head -16 $(find $O/dec -name '*kernel_stats.csv' | head -1) | cut -c1-230 > $O/dec_stats.txt
rm -rf $O/enc $O/dec
