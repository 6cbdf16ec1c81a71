# in-kernel phase stamps of k_encode8_pp<.., MODE 1> (diagnostic build -DHSRLE_PP_STAMPS=1|2; 2 = every stamp waits for all memory operations):
#   HSRLE_LIB=variants/libhsrle_ppst.so python tools/probe_pp_stamps.py [codec] [kind]
import sys, os, ctypes
sys.path.insert(0, 'tests'); sys.path.insert(0, 'hypersonic-rle-kit_amd/python')
import torch, hsrle
key = sys.argv[1] if len(sys.argv) > 1 else 'rle8_packed_multi'
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
size = 8 << 30
nb = size // 4096
src = hsrle.synth(kind, 1, 2, size)
dst = torch.empty(hsrle.container_bound(size, 4096), dtype=torch.uint8, device='cuda'); ws = torch.zeros(hsrle.workspace_size(size, 4096), dtype=torch.uint8, device='cuda')
hsrle.compress_async(key, src, dst, 4096, workspace=ws); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); hsrle.compress_async(key, src, dst, 4096, workspace=ws); e1.record(); torch.cuda.synchronize()
st = ws[(64 << 20):(64 << 20) + nb * 128].view(torch.int64).view(nb, 16)
d = st.sum(dim=0).tolist()
w = max(d[6], 1)
span = (st[:, 12].max() - st[:, 13].min()).item()
print('call %.3f ms; waves %d, candidates per block %.1f; first start -> last end %.0f ticks' % (e0.elapsed_time(e1), d[6], d[7] / w, span))
names = {0: 'input loads (issue + wait)', 1: '-', 2: 'masks, scans, candidate list, image zeroing', 3: 'rounds: loop ends', 8: 'rounds: candidate -> p, symbol', 9: 'rounds: decision chain', 10: 'rounds: sizes, scan, header', 11: 'rounds: short literals', 4: 'terminator + wave-wide stretches', 5: 'copy out'}
tot = sum(d[q] for q in names)
for q, n in names.items():
    print('%-44s %8.0f ticks per wave  %5.1f %%' % (n, d[q] / w, 100.0 * d[q] / tot))
print('total per wave %.0f ticks; mean life (end - start) %.0f' % (tot / w, (st[:, 12] - st[:, 13]).double().mean().item()))
l1, l2 = st[:, 14].double(), st[:, 15].double()
lead = (torch.arange(nb, device=st.device) % 64) == 0
sp1, sp2 = (st[:, 1] >> 32).double(), (st[:, 1] & 0xFFFFFFFF).double()
print('look-back: level 1 wait %.0f ticks (polls %.2f), level 2 wait: group leaders %.0f (polls %.2f), others %.0f (polls %.2f)' % (l1.mean().item(), sp1.mean().item(), l2[lead].mean().item(), sp2[lead].mean().item(), l2[~lead].mean().item(), sp2[~lead].mean().item()))
t0 = st[:, 13].double(); t0 = t0 - t0.min()
import numpy as np
q = t0.cpu().numpy()
print('start time by block index (ticks): block 0 %.0f, 25%% %.0f, 50%% %.0f, 75%% %.0f, last %.0f' % (q[0], q[nb // 4], q[nb // 2], q[3 * nb // 4], q[-1]))
d = np.diff(q.reshape(-1, 8)[:, :].mean(axis=1))
xs = q.reshape(-1, 8)
print('start spread inside 8 consecutive blocks (one per XCD): mean %.0f ticks, p99 %.0f' % ((xs.max(axis=1) - xs.min(axis=1)).mean(), np.percentile(xs.max(axis=1) - xs.min(axis=1), 99)))
