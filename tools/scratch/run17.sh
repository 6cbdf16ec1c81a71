cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3E
( timeout 400 python tools/gpu_stress.py 40 32 2>&1 | grep -v amdgpu.ids | tail -6
  for key in rle64_3symlut_byte rle8_packed_multi rle16_sym rle32_byte_packed; do
  python - <<PY 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0,'hypersonic-rle-kit_amd/python'); sys.path.insert(0,'tests')
import torch, hsrle
from hsrle_testlib import CODEC_BY_KEY
key="$key"; S=CODEC_BY_KEY[key].S
for kind,size in ((1,88473600),(0,64<<20)):
    src=hsrle.synth(kind,S,2,size,device="cuda")
    dst=torch.empty(hsrle.container_bound(size,4096),dtype=torch.uint8,device="cuda"); ws=torch.empty(hsrle.workspace_size(size,4096),dtype=torch.uint8,device="cuda")
    for _ in range(3): hsrle.compress_async(key,src,dst,4096,workspace=ws)
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(20): hsrle.compress_async(key,src,dst,4096,workspace=ws)
    e1.record(); torch.cuda.synchronize()
    print(key,"kind",kind,"bytes",size,"encode us",round(e0.elapsed_time(e1)/20*1e3,1))
PY
  done
) > gpurun_out/r3E/log.txt 2>&1
cat gpurun_out/r3E/log.txt
