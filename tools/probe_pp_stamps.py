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
off = ((1028 * nb + 256) + 255) & ~255   # (behind the records: inst_pp8.hip)
st = ws[off:off + nb * 128].view(torch.int64).view(nb, 16)
d = st.sum(dim=0).tolist()
w = max(d[6], 1)
span = (st[:, 12].max() - st[:, 13].min()).item()
print('call %.3f ms; waves %d, candidates per block %.1f; first start -> last end %.0f ticks' % (e0.elapsed_time(e1), d[6], d[7] / w, span))
names = {0: 'input loads (issue + wait)', 1: '-', 2: 'masks, scans, candidate list, image zeroing', 3: 'rounds: loop ends', 8: 'rounds: candidate -> p, symbol', 9: 'rounds: decision chain', 10: 'rounds: sizes, scan, header', 11: 'rounds: short literals', 4: 'terminator + wave-wide stretches', 5: 'copy out'}
tot = sum(d[q] for q in names)
for q, n in names.items():
    print('%-44s %8.0f ticks per wave  %5.1f %%' % (n, d[q] / w, 100.0 * d[q] / tot))
print('total per wave %.0f ticks; mean life (end - start) %.0f' % (tot / w, (st[:, 12] - st[:, 13]).double().mean().item()))
