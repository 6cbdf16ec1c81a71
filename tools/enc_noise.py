"""Encode time on input without runs (random bytes) and on sparse runs (long literal gaps), 4 KiB blocks:  python tools/enc_noise.py [codec ...]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hypersonic-rle-kit_amd", "python"))
import torch, hsrle
size = 2 << 30
torch.manual_seed(1)
noise = torch.randint(0, 256, (size,), dtype=torch.uint8, device="cuda")
sparse = noise.clone()
v = sparse.view(-1, 1024)
v[:, 500:540] = 7                      # one run of 40 bytes per KiB: literal gaps of ~1000 bytes
for key in (sys.argv[1:] or ["rle8_packed_multi", "rle8_multi", "rle16_sym", "rle64_3symlut_byte"]):
    for name, src in (("noise", noise), ("sparse", sparse)):
        dst = torch.empty(hsrle.container_bound(size, 4096), dtype=torch.uint8, device="cuda")
        ws = torch.empty(hsrle.workspace_size(size, 4096), dtype=torch.uint8, device="cuda")
        for _ in range(2): hsrle.compress_async(key, src, dst, 4096, workspace=ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): hsrle.compress_async(key, src, dst, 4096, workspace=ws)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        out = hsrle.decompress(dst)
        print(os.environ.get("HSRLE_LIB", "default").split("/")[-1], key, name, "encode ms", round(ms, 3), "GiB/s", round(2 / (ms / 1e3), 1), "roundtrip", bool(torch.equal(out, src)), flush=True)
