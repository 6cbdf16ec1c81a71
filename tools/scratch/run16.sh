cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3A
( timeout 300 python tools/probe_correctness.py rle8_packed_multi,rle8_multi,rle8_3symlut,rle16_sym,rle64_3symlut_byte,rle24_byte_packed,rle32_7symlut_sym,rle8_multi_short,rle16_1symlut_sym_short 2>&1 | grep -v amdgpu.ids | tail -8
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "blocks_bit_exact or graph_capturable or stress or long_literal" 2>&1 | tail -8
  timeout 600 python -m pytest tests/test_gpu_big.py -x -q -m gpu -k "config3_video" 2>&1 | tail -4
) > gpurun_out/r3A/log.txt 2>&1
cat gpurun_out/r3A/log.txt
