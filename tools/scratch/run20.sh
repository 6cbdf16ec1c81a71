cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3G
( timeout 300 python tools/split_bench.py --subs 4096,2048,1024,512,256 2>&1 | grep -v amdgpu.ids | tail -8
  timeout 300 python tools/split_bench.py --codec rle8_packed_multi --subs 4096,1024,512,256 2>&1 | grep -v amdgpu.ids | tail -6
) > gpurun_out/r3G/log.txt 2>&1
cat gpurun_out/r3G/log.txt
