cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3s
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
  echo "== experiments build"
  HSRLE_LIB=$PWD/variants/libhsrle_exp.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_split.py -x -q -m gpu 2>&1 | tail -4
  HSRLE_LIB=$PWD/variants/libhsrle_exp.so HSRLE_ENC_RING=128 HSRLE_DEC_RING=64 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "blocks_bit_exact or stress or long_literal" 2>&1 | tail -3
  echo "== bench"
  timeout 600 python bench.py 2>gpurun_out/r3s/bench.err | tee gpurun_out/r3s/bench.json | cut -c1-3000
) > gpurun_out/r3s/log.txt 2>&1
bash tools/traffic.sh r3 >> gpurun_out/r3s/log.txt 2>&1
cat gpurun_out/r3s/log.txt
