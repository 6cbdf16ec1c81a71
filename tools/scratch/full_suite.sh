cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3ai
( timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
  HSRLE_LIB=$PWD/variants/libhsrle_exp.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_split.py -x -q -m gpu -k "wave_per_block or wave_decode or knob or experiment" 2>&1 | tail -4
) > gpurun_out/r3ai/log.txt 2>&1
cat gpurun_out/r3ai/log.txt
