cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3p
( timeout 300 python tools/probe_correctness.py rle8_packed_multi,rle8_multi,rle8_3symlut,rle8_7symlut,rle8_multi_short,rle8_1symlut_short,rle16_sym,rle8_single 2>&1 | grep -v amdgpu.ids | tail -5
  STRESS_KEYS=rle8_multi,rle8_packed_multi,rle8_3symlut,rle8_7symlut,rle8_1symlut timeout 400 python tools/gpu_stress.py 40 13 2>&1 | grep -v amdgpu.ids | tail -4
  for v in default r02 default r02; do
    if [ $v = default ]; then unset HSRLE_LIB; else export HSRLE_LIB=$PWD/variants/libhsrle_$v.so; fi
    HSRLE_ENC_RING=256 timeout 200 python tools/enc_time.py rle8_packed_multi 0 8 2>&1 | tail -1
  done
  unset HSRLE_LIB
  timeout 200 python tools/enc_time.py rle8_packed_multi 1 8 2>&1 | tail -1
  HSRLE_LIB=$PWD/variants/libhsrle_r02.so timeout 200 python tools/enc_time.py rle8_packed_multi 1 8 2>&1 | tail -1
  timeout 200 python tools/enc_time.py rle8_3symlut 0 8 2>&1 | tail -1
  HSRLE_LIB=$PWD/variants/libhsrle_r02.so timeout 200 python tools/enc_time.py rle8_3symlut 0 8 2>&1 | tail -1
  timeout 300 python bench.py --no-cpu --no-extras --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-1500
) > gpurun_out/r3p/log.txt 2>&1
cat gpurun_out/r3p/log.txt
