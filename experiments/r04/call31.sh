#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
python - <<'PY'
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import torch, hsrle
size=4<<30; bs=4096
nb=size//bs; stride=(hsrle.lib().rle_compress_bounds(bs)+15)&~15
off=((nb*stride+255)//256)*256
dst = torch.empty(hsrle.container_bound(size, bs), dtype=torch.uint8, device='cuda'); ws = torch.empty(hsrle.workspace_size(size, bs), dtype=torch.uint8, device='cuda')
for kind in (0,1):
  for k in ('rle8_packed_multi','rle16_sym','rle24_byte','rle24_sym_packed','rle32_byte','rle32_3symlut_sym_short'):
    S = {'8':1,'16':2,'24':3,'32':4}[k[3:].split('_')[0]]
    src = hsrle.synth(kind, S, 5, size)
    ts=[]
    for i in range(5):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(); hsrle.compress_async(k, src, dst, bs, workspace=ws); e1.record(); torch.cuda.synchronize()
        ts.append(round(e0.elapsed_time(e1),3))
    sel=ws[off:off+16].view(torch.int32).cpu().tolist()
    print(('runs','video')[kind], k, 'ms', min(ts), 'GiB/s %.0f'%(4/min(ts)*1e3), 'sel', sel, 'e/n %.3f runs/KiB %.1f'%(sel[2]/max(sel[1],1), sel[3]*1024/max(sel[1],1)))
PY
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_big.py -x -q 2>&1 | tail -3 )
