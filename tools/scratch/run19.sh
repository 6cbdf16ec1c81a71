cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3F
( timeout 1500 python -m pytest tests/test_gpu_mono.py -x -q -m gpu -k "test_mono_encode" 2>&1 | tail -4
  timeout 400 python tools/gpu_stress.py 60 41 2>&1 | grep -v amdgpu.ids | tail -4
  bash tools/scratch/run18.sh 2>&1 | head -6
  for key in rle64_3symlut_byte rle16_sym rle32_byte_packed; do python tools/scratch/frame_time.py $key 2>&1 | grep -v amdgpu.ids; done
  timeout 300 python tools/mono_enc_bench.py rle16_sym,rle64_3symlut_byte,rle128_sym 2>&1 | grep -v amdgpu.ids | tail -6
) > gpurun_out/r3F/log.txt 2>&1
cat gpurun_out/r3F/log.txt
