cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3q
( for v in default xntst r02 default xntst r02; do
    if [ $v = default ]; then unset HSRLE_LIB; else export HSRLE_LIB=$PWD/variants/libhsrle_$v.so; fi
    HSRLE_ENC_RING=256 timeout 200 python tools/enc_time.py rle8_packed_multi 0 8 2>&1 | tail -1
  done
) > gpurun_out/r3q/log.txt 2>&1
cat gpurun_out/r3q/log.txt
