cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3c
( HSRLE_ENC_RING=256 HSRLE_LIB=$PWD/variants/libhsrle_e8stats.so timeout 200 python tools/e8_stats.py rle8_packed_multi 0 2 2>&1 | grep -v amdgpu.ids
  HSRLE_ENC_RING=256 HSRLE_LIB=$PWD/variants/libhsrle_e8stats.so timeout 200 python tools/e8_stats.py rle8_packed_multi 1 2 2>&1 | grep -v amdgpu.ids
) > gpurun_out/r3c/log.txt 2>&1
cat gpurun_out/r3c/log.txt
