"""First block of a synthetic buffer whose GPU stream differs from the oracle's:  python tools/find_mismatch.py <codec> <kind 0|1> <MiB>
(the block, both streams -> gpurun_out/mismatch.json; how the rle128 pair-search bug of round 2 was pinned)"""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hypersonic-rle-kit_amd", "python"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch, hsrle
from hsrle_testlib import CODEC_BY_KEY, Oracle
key, kind, size, bs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) << 20, 4096
codec = CODEC_BY_KEY[key]
ora = Oracle()
src = hsrle.synth(kind, codec.S, 2, size, device="cuda")
data = src.cpu().numpy().tobytes()
container, info = hsrle.compress(key, src, block_size=bs)
cinfo, streams = hsrle.split_container(container.cpu().numpy().tobytes())
bad = 0
for i, s in enumerate(streams):
    blk = data[i * bs:(i + 1) * bs]
    e = ora.compress(codec, blk)
    if s != e:
        bad += 1
        if bad == 1:
            k = next(j for j in range(min(len(s), len(e))) if s[j] != e[j]) if s[:min(len(s), len(e))] != e[:min(len(s), len(e))] else min(len(s), len(e))
            print("block", i, "sizes", len(s), len(e), "first diff at", k)
            os.makedirs("gpurun_out", exist_ok=True)
            json.dump({"block": blk.hex(), "gpu": s.hex(), "ref": e.hex()}, open("gpurun_out/mismatch.json", "w"))
print("blocks", len(streams), "bad", bad)
