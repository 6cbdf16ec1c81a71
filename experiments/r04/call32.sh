#!/bin/bash
# phase stamps of k_encodeS_runlist on the config-3 frame
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
env HSRLE_LIB=variants/libhsrle_rl.so python - <<'PY'
import sys, ctypes; sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import torch, hsrle
size=88473600
src = hsrle.synth(1, 8, 2, size, device="cuda")
dst = torch.empty(hsrle.container_bound(size, 4096), dtype=torch.uint8, device='cuda'); ws = torch.empty(hsrle.workspace_size(size, 4096), dtype=torch.uint8, device='cuda')
L = hsrle.lib()
buf=(ctypes.c_ulonglong*8)()
for i in range(3): hsrle.compress_async("rle64_3symlut_byte", src, dst, 4096, workspace=ws)
torch.cuda.synchronize(); L.hsrle_debug_rl_stamps(buf, 1)
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record(); hsrle.compress_async("rle64_3symlut_byte", src, dst, 4096, workspace=ws); e1.record(); torch.cuda.synchronize()
L.hsrle_debug_rl_stamps(buf, 1)
v=list(buf); w=max(v[3],1)
print('call us', e0.elapsed_time(e1)*1e3, 'waves', v[3], 'per wave cycles: A', v[0]//w, 'B', v[1]//w, 'C', v[2]//w, 'flushes/wave', v[4]/w, 'candidates/wave', v[5]/w)
PY
