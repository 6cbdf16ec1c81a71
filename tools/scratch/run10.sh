cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3r
( timeout 300 python tools/probe_correctness.py rle8_packed_multi,rle8_multi,rle8_3symlut,rle8_single,rle8_packed_single,rle16_sym,rle24_byte,rle32_3symlut_sym,rle48_byte_packed,rle64_3symlut_byte,rle128_sym,rle128_byte_packed 2>&1 | grep -v amdgpu.ids | tail -5
  timeout 400 python tools/gpu_stress.py 60 21 2>&1 | grep -v amdgpu.ids | tail -4
  for key in rle8_packed_multi rle8_3symlut rle16_sym rle64_3symlut_byte rle8_single rle128_sym; do for kind in 0 1; do for v in default r02; do
    if [ $v = default ]; then unset HSRLE_LIB; else export HSRLE_LIB=$PWD/variants/libhsrle_$v.so; fi
    timeout 200 python tools/enc_time.py $key $kind 8 2>&1 | tail -1
  done; done; done
) > gpurun_out/r3r/log.txt 2>&1
cat gpurun_out/r3r/log.txt
