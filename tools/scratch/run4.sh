cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3h
( HSRLE_ENC_RING=256 HSRLE_LIB=$PWD/variants/libhsrle_e8stats.so timeout 200 python tools/e8_stats.py rle8_packed_multi 0 2 2>&1 | grep -v amdgpu.ids
  timeout 300 python tools/probe_correctness.py rle8_packed_multi,rle8_multi,rle8_3symlut,rle8_7symlut,rle8_multi_short,rle8_1symlut_short,rle8_3symlut_short,rle8_7symlut_short 2>&1 | grep -v amdgpu.ids | tail -5
  STRESS_KEYS=rle8_multi,rle8_packed_multi,rle8_3symlut,rle8_7symlut,rle8_1symlut timeout 400 python tools/gpu_stress.py 40 11 2>&1 | grep -v amdgpu.ids | tail -5
  for i in 1 2; do
  timeout 200 python tools/enc_time.py rle8_packed_multi 0 8 2>&1 | tail -1
  done
  HSRLE_LIB=$PWD/variants/libhsrle_r02.so timeout 200 python tools/enc_time.py rle8_packed_multi 0 8 2>&1 | tail -1
  timeout 200 python tools/enc_time.py rle8_packed_multi 1 8 2>&1 | tail -1
  HSRLE_ENC_RING=256 timeout 200 python tools/enc_time.py rle8_packed_multi 1 8 2>&1 | tail -1
  HSRLE_LIB=$PWD/variants/libhsrle_r02.so timeout 200 python tools/enc_time.py rle8_packed_multi 1 8 2>&1 | tail -1
) > gpurun_out/r3h/log.txt 2>&1
cat gpurun_out/r3h/log.txt
