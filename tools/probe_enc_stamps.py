# in-kernel phase stamps of k_encode8_blocks (diagnostic build -DHSRLE_ENC_STAMPS):  HSRLE_LIB=variants/libhsrle_encst.so python tools/probe_enc_stamps.py [codec] [kind]
import sys, os, ctypes
sys.path.insert(0, 'tests'); sys.path.insert(0, 'hypersonic-rle-kit_amd/python')
import torch, hsrle
key = sys.argv[1] if len(sys.argv) > 1 else 'rle8_packed_multi'
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
size = 8 << 30
src = hsrle.synth(kind, 1, 2, size)
dst = torch.empty(hsrle.container_bound(size, 4096), dtype=torch.uint8, device='cuda'); ws = torch.empty(hsrle.workspace_size(size, 4096), dtype=torch.uint8, device='cuda')
hsrle.compress_async(key, src, dst, 4096, workspace=ws); torch.cuda.synchronize()
L = hsrle.lib()
buf = (ctypes.c_ulonglong * 8)()
L.hsrle_debug_enc_stamps(buf, 1)
hsrle.compress_async(key, src, dst, 4096, workspace=ws); torch.cuda.synchronize()
L.hsrle_debug_enc_stamps(buf, 0)
d = list(buf)
steps = d[5] / max(d[6], 1)
print('waves', d[6], 'steps/wave %.1f' % steps, '| run-end loop: %.2f wave trips per step, %.2f run ends per lane and step' % ((d[7] >> 32) / max(d[5], 1), (d[7] & 0xFFFFFFFF) / max(d[5], 1) / 64))
names = ['loop top', 'issue (exchange + loads)', 'scan masks', 'run ends (handle_run)', 'land (wait + ring writes)']
tot = sum(d[:5])
for n, v in zip(names, d[:5]):
    print('%-28s %8.0f cycles per step  %5.1f %%' % (n, v / max(d[5], 1), 100.0 * v / tot))
print('total per step %.0f' % (tot / max(d[5], 1)))
