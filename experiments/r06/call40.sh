#!/bin/bash
# round 6, call 40: kernel times of the windowed encoders on the final build (8 GiB in 64 KiB blocks: rle8_packed_multi, rle32_byte, rle64_3symlut_byte; the 1 GiB monolithic encode)
O=$GRAFT_REPO_ROOT/gpurun_out/r06_c40; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # name, script args...
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o p -- python3 "$@" > $O/$name.log 2>&1
  f=$(find $O/$name -name '*kernel_stats.csv' 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp "$f" $O/${name}_kernel_stats.csv; fi
  rm -rf $O/$name
}
run packed8_64k $GRAFT_REPO_ROOT/tools/enc_time.py rle8_packed_multi 0 8 65536
run byte32_64k $GRAFT_REPO_ROOT/tools/enc_time.py rle32_byte 0 8 65536
run lut64_64k $GRAFT_REPO_ROOT/tools/enc_time.py rle64_3symlut_byte 0 8 65536
run lut7_32_64k $GRAFT_REPO_ROOT/tools/enc_time.py rle32_7symlut_byte 0 8 65536
run lut7_8_64k $GRAFT_REPO_ROOT/tools/enc_time.py rle8_7symlut 0 8 65536
run mono $GRAFT_REPO_ROOT/tools/mono_enc_bench.py rle8_packed_multi 1
grep -h "encode ms\|kind" $O/*.log | grep -v amdgpu
