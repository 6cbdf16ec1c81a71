"""Same-box A/B of decode / encode throughput per codec:  HSRLE_LIB=<lib> python tools/ab_codecs.py <size_mib> <codec,codec,...> [kinds]
Prints one line per codec and data kind (runs / video): decode and encode GiB/s, best of 3 batches of 5 launches each."""
import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, 'hypersonic-rle-kit_amd/python')
import torch, hsrle
from hsrle_testlib import CODEC_BY_KEY
size = int(sys.argv[1]) << 20
keys = sys.argv[2].split(',')
kinds = [int(k) for k in sys.argv[3].split(',')] if len(sys.argv) > 3 else [0, 1]
bs = 4096
def bench(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / n / 1e3
        best = t if best is None else min(best, t)
    return best
dst = torch.empty(hsrle.container_bound(size, bs), dtype=torch.uint8, device='cuda'); ws = torch.empty(hsrle.workspace_size(size, bs), dtype=torch.uint8, device='cuda')
out = torch.empty(size, dtype=torch.uint8, device='cuda'); st = torch.zeros(16, dtype=torch.int32, device='cuda')
tag = os.path.basename(os.environ.get('HSRLE_LIB', 'default'))
for kind in kinds:
    for k in keys:
        c = CODEC_BY_KEY[k]
        src = hsrle.synth(kind, c.S, 5, size)
        hsrle.compress_async(k, src, dst, bs, workspace=ws); torch.cuda.synchronize()
        info = hsrle.container_info(dst)
        td = bench(lambda: hsrle.decompress_async(dst, info, out, st))
        ok = int(st[0].item()) == 0 and torch.equal(out, src)
        te = bench(lambda: hsrle.compress_async(k, src, dst, bs, workspace=ws), 3)
        print('%-18s %-26s %-5s dec %6.0f enc %6.0f GiB/s  %s' % (tag, k, ('runs', 'video')[kind], size / td / 2**30, size / te / 2**30, 'ok' if ok else 'FAIL'), flush=True)
        del src
