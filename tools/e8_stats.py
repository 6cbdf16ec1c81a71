"""What the waves of the 8 bit ring encoder execute (diagnostic build -DHSRLE_E8_STATS, variants/libhsrle_e8stats.so):
python tools/e8_stats.py [codec] [kind] [GiB]"""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hypersonic-rle-kit_amd", "python"))
import torch, hsrle
key = sys.argv[1] if len(sys.argv) > 1 else "rle8_packed_multi"
kind = int(sys.argv[2]) if len(sys.argv) > 2 else hsrle.SYNTH_RUNS
gib = int(sys.argv[3]) if len(sys.argv) > 3 else 2
size = gib << 30
src = hsrle.synth(kind, 1, 2, size, device="cuda")
dst = torch.empty(hsrle.container_bound(size, 4096), dtype=torch.uint8, device="cuda")
ws = torch.empty(hsrle.workspace_size(size, 4096), dtype=torch.uint8, device="cuda")
L = hsrle.lib()
out = (ctypes.c_ulonglong * 32)()
L.hsrle_debug_e8stats(out, 1)
hsrle.compress_async(key, src, dst, 4096, workspace=ws)
torch.cuda.synchronize()
L.hsrle_debug_e8stats(out, 1)
names = ["waves", "trips", "active lane-trips", "emit lane-trips", "trips with a global literal path", "literal loop passes (wave)", "literal chunks (lanes)",
         "header stores (lanes)", "loading lane-trips", "starved lane-trips", "mask passes (wave)", "mask chunks (lanes)"]
w = max(1, out[0])
for i, nme in enumerate(names):
    print(f"{nme:40s} {out[i]:14d}  per wave {out[i] / w:10.2f}")
