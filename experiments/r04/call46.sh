#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
( timeout 2400 python -m pytest tests/test_gpu_mono.py tests/test_gpu_parity.py -x -q 2>&1 | tail -6 )
python - <<'PY'
# how fast: greedy drop-in encode through the device API, 256 MiB run-distributed
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import torch, hsrle, time
from hsrle_testlib import CODEC_BY_KEY
for key in ("rle16_3symlut_byte_short_greedy","rle32_1symlut_byte_short_greedy","rle64_7symlut_byte_short_greedy"):
    c=CODEC_BY_KEY[key]
    src=hsrle.synth(0,c.S,5,256<<20,device="cuda")
    for i in range(2):
        torch.cuda.synchronize(); t0=time.time(); s,ch=hsrle.mono_compress_dev(key,src,return_chunks=True); torch.cuda.synchronize(); t=time.time()-t0
    print(key,'256 MiB runs: %.2f ms, %.0f GiB/s, chunks %d, stats %s'%(t*1e3,0.25/t,ch,hsrle.mono_encode_stats()))
PY
