#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
python - <<'PY'
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import torch, hsrle
size=4<<30; bs=4096
nb=size//bs; stride=(hsrle.lib().rle_compress_bounds(bs)+15)&~15
off=((nb*stride+255)//256)*256
dst = torch.empty(hsrle.container_bound(size, bs), dtype=torch.uint8, device='cuda'); ws = torch.empty(hsrle.workspace_size(size, bs), dtype=torch.uint8, device='cuda')
print('stride',stride,'off',off,'ws',ws.numel())
for k in ('rle24_byte','rle24_sym','rle24_byte','rle24_sym_packed','rle32_byte'):
    S = 3 if '24' in k else 4
    src = hsrle.synth(0, S, 5, size)
    ts=[]
    for i in range(5):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(); hsrle.compress_async(k, src, dst, bs, workspace=ws); e1.record(); torch.cuda.synchronize()
        ts.append(round(e0.elapsed_time(e1),3))
    sel=ws[off:off+16].view(torch.int32).cpu().tolist()
    print(k, ts, 'sel', sel, 'runs/KiB', sel[3]*1024/max(sel[1],1))
PY
echo "== packet walk stamps (variants/libhsrle_stamps.so)"
env HSRLE_LIB=variants/libhsrle_stamps.so python - <<'PY'
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import torch, hsrle
for key,kind,S,size in (("rle64_3symlut_byte",1,8,88473600),("rle8_packed_multi",0,1,67108864)):
    src = hsrle.synth(kind, S, 3, size, device="cuda")
    cont, info = hsrle.compress(key, src, block_size=4096)
    out = torch.empty(size, dtype=torch.uint8, device="cuda")
    st = torch.zeros(16, dtype=torch.int32, device="cuda")
    ws = torch.empty(hsrle.split_workspace_size(info, None, 1), dtype=torch.uint8, device="cuda")
    for i in range(3):
        st.zero_()
        hsrle.decompress_split_async(cont, info, out, ws, st, sub_block=1); torch.cuda.synchronize()
    v = st.cpu().numpy().view('uint32')
    stage = int(v[2]) | (int(v[3])<<32); walk = int(v[4]) | (int(v[5])<<32); waves=int(v[6])
    print(key, 'waves', waves, 'stage cycles/wave', stage//max(waves,1), 'walk cycles/wave', walk//max(waves,1), 'max entries', int(v[7]), 'mean entries', int(v[8])/info.blockCount, 'status', int(v[0]), 'ok', torch.equal(out,src))
PY
