"""Encode time of one codec on the 8 GiB run-distributed buffer:  python tools/enc_time.py [codec] [kind] [GiB] [block size, 4096]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hypersonic-rle-kit_amd", "python"))
import torch, hsrle
key = sys.argv[1] if len(sys.argv) > 1 else "rle8_packed_multi"
kind = int(sys.argv[2]) if len(sys.argv) > 2 else hsrle.SYNTH_RUNS
gib = int(sys.argv[3]) if len(sys.argv) > 3 else 8
size = gib << 30
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
S = {"rle8": 1, "rle16": 2, "rle24": 3, "rle32": 4, "rle48": 6, "rle64": 8, "rle128": 16}[key.split("_")[0]]
src = hsrle.synth(kind, S, 2, size, device="cuda")
dst = torch.empty(hsrle.container_bound(size, B), dtype=torch.uint8, device="cuda")
ws = torch.empty(hsrle.workspace_size(size, B), dtype=torch.uint8, device="cuda")
for _ in range(2): hsrle.compress_async(key, src, dst, B, workspace=ws)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): hsrle.compress_async(key, src, dst, B, workspace=ws)
e1.record(); torch.cuda.synchronize()
print(os.environ.get("HSRLE_LIB", "default").split("/")[-1], key, "kind", kind, "B", B, "path", hsrle.lib().hsrle_encode_path(hsrle.codec_id(key), size, B), "encode ms", round(e0.elapsed_time(e1) / 5, 3), "GiB/s", round(gib / (e0.elapsed_time(e1) / 5e3), 1), flush=True)
