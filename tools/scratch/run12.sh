cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3t
( timeout 1200 python -m pytest tests/test_gpu_mono.py -x -q -m gpu -k "test_mono_encode_is_the_reference_stream and (single or rle128 or rle8_packed_multi or rle16_sym)" 2>&1 | tail -15
  timeout 600 python tools/mono_enc_bench.py 2>&1 | grep -v amdgpu.ids | tail -12
) > gpurun_out/r3t/log.txt 2>&1
cat gpurun_out/r3t/log.txt
