"""Encode time of small containers for every codec: the 88 MB video-shaped frame of BASELINE config 3 and a 64 MiB run-distributed buffer, 4 KiB
blocks (21 600 / 16 384 blocks: fewer than one lane per block needs).  Markdown table on stdout; every container is decoded and compared.
  python tools/small_container_sweep.py > profiles/rNN_small_containers.md"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R + "/hypersonic-rle-kit_amd/python"); sys.path.insert(0, R + "/tests")
import torch, hsrle
from hsrle_testlib import CODECS

def encode_us(key, src, reps=20):
    size = src.numel()
    dst = torch.empty(hsrle.container_bound(size, 4096), dtype=torch.uint8, device="cuda")
    ws = torch.empty(hsrle.workspace_size(size, 4096, codec=key), dtype=torch.uint8, device="cuda")   # (8 bit Single / 128 bit: with the split encode regions)
    for _ in range(3): info = hsrle.compress_async(key, src, dst, 4096, workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): hsrle.compress_async(key, src, dst, 4096, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    container, cinfo = hsrle.compress(key, src, block_size=4096)
    ok = torch.equal(hsrle.decompress(container), src)
    return e0.elapsed_time(e1) / reps * 1e3, cinfo.totalSize / size, ok

print("Small containers, 4 KiB blocks, device resident, build", hsrle.build_id(), "\n")
print("| codec | 88 MB frame: encode us | GiB/s | ratio | 64 MiB runs: encode us | GiB/s | ratio | round trip |")
print("|---|---:|---:|---:|---:|---:|---:|---|")
data = {}
for c in CODECS:
    row = [c.key]; good = True
    for kind, size in ((1, 88473600), (0, 64 << 20)):
        if (kind, c.S) not in data: data[(kind, c.S)] = hsrle.synth(kind, c.S, 2, size, device="cuda")
        us, ratio, ok = encode_us(c.key, data[(kind, c.S)])
        good &= ok
        row += ["%.1f" % us, "%.0f" % (size / 2**30 / (us * 1e-6)), "%.4f" % ratio]
    print("| " + " | ".join(row + ["ok" if good else "FAIL"]) + " |", flush=True)
