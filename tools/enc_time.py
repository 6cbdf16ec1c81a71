"""Encode time of one codec on the 8 GiB run-distributed buffer (4 KiB blocks):  python tools/enc_time.py [codec] [kind] [GiB]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hypersonic-rle-kit_amd", "python"))
import torch, hsrle
key = sys.argv[1] if len(sys.argv) > 1 else "rle8_packed_multi"
kind = int(sys.argv[2]) if len(sys.argv) > 2 else hsrle.SYNTH_RUNS
gib = int(sys.argv[3]) if len(sys.argv) > 3 else 8
size = gib << 30
S = {"rle8": 1, "rle16": 2, "rle24": 3, "rle32": 4, "rle48": 6, "rle64": 8, "rle128": 16}[key.split("_")[0]]
src = hsrle.synth(kind, S, 2, size, device="cuda")
dst = torch.empty(hsrle.container_bound(size, 4096), dtype=torch.uint8, device="cuda")
ws = torch.empty(hsrle.workspace_size(size, 4096), dtype=torch.uint8, device="cuda")
for _ in range(2): hsrle.compress_async(key, src, dst, 4096, workspace=ws)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): hsrle.compress_async(key, src, dst, 4096, workspace=ws)
e1.record(); torch.cuda.synchronize()
print(os.environ.get("HSRLE_LIB", "default").split("/")[-1], key, "kind", kind, "encode ms", round(e0.elapsed_time(e1) / 5, 3), "GiB/s", round(gib / (e0.elapsed_time(e1) / 5e3), 1), flush=True)
