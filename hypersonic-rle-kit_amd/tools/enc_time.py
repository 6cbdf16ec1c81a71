import sys, os
sys.path.insert(0, "/root/repo/hypersonic-rle-kit_amd/python")
import torch, hsrle
size = 8 << 30
src = hsrle.synth(hsrle.SYNTH_RUNS, 1, 2, size, device="cuda")
dst = torch.empty(hsrle.container_bound(size, 4096), dtype=torch.uint8, device="cuda")
ws = torch.empty(hsrle.workspace_size(size, 4096), dtype=torch.uint8, device="cuda")
for _ in range(2): hsrle.compress_async("rle8_packed_multi", src, dst, 4096, workspace=ws)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): hsrle.compress_async("rle8_packed_multi", src, dst, 4096, workspace=ws)
e1.record(); torch.cuda.synchronize()
print(os.environ.get("HSRLE_LIB", "default").split("/")[-1], "encode ms", e0.elapsed_time(e1) / 5, "GiB/s", 8 / (e0.elapsed_time(e1) / 5e3))
