cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3ad
( export HSRLE_LIB=$PWD/variants/libhsrle_exp.so
  for key in rle64_3symlut_byte rle64_sym rle32_sym_packed rle16_sym rle8_multi rle64_7symlut_byte_short; do for kind in 0 1; do for rl in 2 1; do echo "$key kind $kind runlist $rl: $(HSRLE_RUNLIST=$rl python tools/enc_time.py $key $kind 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/.*encode ms/ms/')"; done; done; done
) > gpurun_out/r3ad/log.txt 2>&1
cat gpurun_out/r3ad/log.txt
