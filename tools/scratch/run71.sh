cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3x
( for i in 1 2; do for key in rle64_3symlut_byte rle8_packed_multi; do python tools/frame_enc_time.py $key 2>&1 | grep -v amdgpu.ids; HSRLE_LIB=$PWD/variants/libhsrle_exp.so python tools/frame_enc_time.py $key 2>&1 | grep -v amdgpu.ids; done; done
) > gpurun_out/r3x/log.txt 2>&1
cat gpurun_out/r3x/log.txt
