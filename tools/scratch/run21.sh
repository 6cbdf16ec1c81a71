cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3I
( HSRLE_LIB=$PWD/variants/libhsrle_exp.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wave_per_block" 2>&1 | tail -3
  for v in exp0 exp exp0 exp; do
    HSRLE_ENCODE_WAVE=1 HSRLE_LIB=$PWD/variants/libhsrle_$v.so timeout 200 python tools/enc_time.py rle8_packed_multi 0 8 2>&1 | tail -1
  done
  HSRLE_ENCODE_WAVE=1 HSRLE_LIB=$PWD/variants/libhsrle_exp.so timeout 200 python tools/enc_time.py rle8_packed_multi 1 8 2>&1 | tail -1
  timeout 200 python tools/enc_time.py rle8_packed_multi 0 8 2>&1 | tail -1
) > gpurun_out/r3I/log.txt 2>&1
cat gpurun_out/r3I/log.txt
