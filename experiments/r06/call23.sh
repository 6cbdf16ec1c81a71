#!/bin/bash
# round 6, call 23: kernel times of the windowed monolithic encode (1 GiB rle8_packed), and the cut finder's piece size
O=$GRAFT_REPO_ROOT/gpurun_out/r06_c23; mkdir -p $O
cd $GRAFT_REPO_ROOT
for g in 2048 4096 16384; do echo "== MONO_G $g"; MONO_G=$g timeout 300 python tools/mono_enc_bench.py rle8_packed_multi 1; done 2>&1 | grep -v amdgpu.ids | tee $O/g.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -o e -- python3 $GRAFT_REPO_ROOT/tools/mono_enc_bench.py rle8_packed_multi 1 > $O/enc.log 2>&1
f=$(find $O/enc -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $O/enc_kernel_stats.csv
rm -rf $O/enc
