# which codecs need a repair round on 3 MiB of run data at the default tuning?  (async status against the synchronous function's stats)
import sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/hypersonic-rle-kit_amd/python")
import torch, numpy as np, hsrle
from hsrle_testlib import CODECS, SYNTH_RUNS, Oracle
o = Oracle()
for c in CODECS[:40:3]:
    data = o.synth(SYNTH_RUNS, c.S, 5, (3 << 20) + 777)
    stream = o.compress(c, data.tobytes())
    t = torch.zeros(len(stream) + 64, dtype=torch.uint8, device="cuda"); t[:len(stream)] = torch.frombuffer(bytearray(stream), dtype=torch.uint8).cuda()
    out = torch.empty(data.size, dtype=torch.uint8, device="cuda")
    ws = torch.empty(max(hsrle.mono_decompress_workspace_size(c.key, data.size, len(stream)), 256), dtype=torch.uint8, device="cuda")
    st = torch.full((1,), 77, dtype=torch.int32, device="cuda")
    hsrle.mono_decompress_dev_async(c.key, t, stream[:16], out, ws, st, stream_size=len(stream)); torch.cuda.synchronize()
    ctrl_off = None
    got, stats = hsrle.mono_decompress_dev(c.key, t, return_stats=True)
    print(c.key, "async status", int(st.item()), "sync stats", stats, "ok", bool(torch.equal(got.cpu(), torch.from_numpy(data))), flush=True)
