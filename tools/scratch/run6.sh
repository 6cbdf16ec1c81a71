cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3k
( STRESS_KEYS=rle8_multi,rle8_packed_multi,rle8_3symlut,rle8_7symlut,rle8_1symlut timeout 400 python tools/gpu_stress.py 30 12 2>&1 | grep -v amdgpu.ids | tail -4
  HSRLE_ENC_RING=256 HSRLE_LIB=$PWD/variants/libhsrle_e8stats.so timeout 200 python tools/e8_stats.py rle8_packed_multi 0 2 2>&1 | grep -v amdgpu.ids | head -10
  for v in default; do
    if [ $v = default ]; then unset HSRLE_LIB; else export HSRLE_LIB=$PWD/variants/libhsrle_$v.so; fi
    HSRLE_ENC_RING=256 timeout 200 python tools/enc_time.py rle8_packed_multi 0 8 2>&1 | tail -1
    HSRLE_ENC_RING=256 timeout 200 python tools/enc_time.py rle8_packed_multi 1 8 2>&1 | tail -1
  done
) > gpurun_out/r3k/log.txt 2>&1
cat gpurun_out/r3k/log.txt
