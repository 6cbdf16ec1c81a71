#!/bin/bash
# round 6, call 24: windowed monolithic encode, the cut finder's piece size G against the input size
O=$GRAFT_REPO_ROOT/gpurun_out/r06_c24; mkdir -p $O
for gib in 1 0.25 0.0824 0.015625; do
  for g in 4096 8192 16384 32768 65536 131072; do echo "== GiB $gib MONO_G $g"; MONO_G=$g timeout 300 python tools/mono_enc_bench.py rle8_packed_multi $gib; done
done 2>&1 | grep -v amdgpu.ids | tee $O/g.log
