cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3v
( STRESS_KEYS=rle8_m,rle8_p,rle8_3,rle16_3,rle32_,rle64_ timeout 200 python tools/gpu_stress.py 40 191 2>&1 | grep -v amdgpu.ids | tail -3
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "small_containers or overflow or graph_capturable" 2>&1 | tail -3
  for key in rle64_3symlut_byte rle8_packed_multi rle16_7symlut_byte; do python tools/frame_enc_time.py $key 2>&1 | grep -v amdgpu.ids; HSRLE_LIB=$PWD/variants/libhsrle_exp.so python tools/frame_enc_time.py $key 2>&1 | grep -v amdgpu.ids; done
) > gpurun_out/r3v/log.txt 2>&1
cat gpurun_out/r3v/log.txt
