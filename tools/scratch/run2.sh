cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3b
( for v in default xhot xnost xboth r02; do
    if [ $v = default ]; then unset HSRLE_LIB; else export HSRLE_LIB=$PWD/variants/libhsrle_$v.so; fi
    timeout 200 python tools/enc_time.py rle8_packed_multi 0 8 2>&1 | tail -1
  done
  unset HSRLE_LIB
  HSRLE_ENC_RING=128 timeout 200 python tools/enc_time.py rle8_packed_multi 0 8 2>&1 | tail -1
  HSRLE_ENC_RING=256 timeout 200 python tools/enc_time.py rle8_packed_multi 0 8 2>&1 | tail -1
  HSRLE_ENC_RING=128 timeout 200 python tools/enc_time.py rle8_packed_multi 1 8 2>&1 | tail -1
  HSRLE_ENC_RING=256 timeout 200 python tools/enc_time.py rle8_packed_multi 1 8 2>&1 | tail -1
  bash tools/pmc_kernel.sh r3b k_encode8_blocks -- tools/enc_time.py rle8_packed_multi 0 8
) > gpurun_out/r3b/log.txt 2>&1
cat gpurun_out/r3b/log.txt
