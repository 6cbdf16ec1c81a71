"""Position-parallel encoder against the ring / run list encoders by container size (experiment build: HSRLE_PP=1 always, 2 never):
   HSRLE_LIB=variants/libhsrle_exp.so HSRLE_PP=1 python tools/pp_threshold.py"""
import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, 'hypersonic-rle-kit_amd/python')
import torch, hsrle
bs = 4096
def bench(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / n
        best = t if best is None else min(best, t)
    return best
for key in ('rle8_packed_multi', 'rle8_multi'):
    for kind in (0, 1):
        row = []
        for mib in (1, 4, 16, 64, 256, 1024):
            size = mib << 20
            src = hsrle.synth(kind, 1, 5, size)
            dst = torch.empty(hsrle.container_bound(size, bs), dtype=torch.uint8, device='cuda'); ws = torch.empty(hsrle.workspace_size(size, bs), dtype=torch.uint8, device='cuda')
            t = bench(lambda: hsrle.compress_async(key, src, dst, bs, workspace=ws))
            row.append('%4d MiB %8.1f us' % (mib, t * 1e3))
            del src, dst, ws
        print('PP=%s %-18s %-5s %s' % (os.environ.get('HSRLE_PP', '0'), key, ('runs', 'video')[kind], ' | '.join(row)), flush=True)
