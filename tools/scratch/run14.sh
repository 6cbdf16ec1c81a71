cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3z
( timeout 900 python -m pytest tests/test_gpu_mono.py -x -q -m gpu -k "test_mono_encode_is_the_reference_stream and single" 2>&1 | tail -3
  for g in ""; do echo "G=$g"; MONO_G=$g timeout 600 python tools/mono_enc_bench.py rle8_single,rle128_byte_packed 2>&1 | grep -v amdgpu.ids | tail -4; done
) > gpurun_out/r3z/log.txt 2>&1
cat gpurun_out/r3z/log.txt
