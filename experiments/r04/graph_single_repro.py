# repro: rle8_single small container (split encode) under graph replay with a workspace full of garbage
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "hypersonic-rle-kit_amd", "python"))
import torch, hsrle as hs
key = sys.argv[1] if len(sys.argv) > 1 else "rle8_single"
block = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
fill = int(sys.argv[2]) if len(sys.argv) > 2 else 0xFF
size = (8 << 20) + 4096 * 3 + 77
src = hs.synth(hs.SYNTH_RUNS, 1, 9, size)
print(key, block, 'path', hs.lib().hsrle_encode_path(hs.codec_id(key), size, block), flush=True)
dst = torch.empty(hs.container_bound(size, block), dtype=torch.uint8, device="cuda")
ws = torch.full((hs.workspace_size(size, block, key),), fill, dtype=torch.uint8, device="cuda")
out = torch.zeros(size, dtype=torch.uint8, device="cuda")
status = torch.zeros(16, dtype=torch.int32, device="cuda")
for rep in range(3):
    hs.compress_async(key, src, dst, block, workspace=ws)
    torch.cuda.synchronize()
    print("eager", rep, "ok", flush=True)
info = hs.container_info(dst)
eager = dst[: info.totalSize].clone()
ws.fill_(fill)
hs.compress_async(key, src, dst, block, workspace=ws)
torch.cuda.synchronize()
print("eager after refill ok", torch.equal(dst[: info.totalSize], eager), flush=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        hs.compress_async(key, src, dst, block, workspace=ws)
        hs.decompress_async(dst, info, out, status)
torch.cuda.current_stream().wait_stream(side)
for rep in range(3):
    dst.zero_(); out.zero_(); status.zero_()
    if rep == 1: ws.fill_(fill)
    g.replay()
    torch.cuda.synchronize()
    print("replay", rep, torch.equal(dst[: info.totalSize], eager), int(status[0].item()), torch.equal(out, src), flush=True)
