# which of {graph replay, zeroed container, garbage workspace} breaks the rle8_single split encode?
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "hypersonic-rle-kit_amd", "python"))
import torch, hsrle as hs
key, mode = sys.argv[1], sys.argv[2]
block = 4096
size = (8 << 20) + 4096 * 3 + 77
src = hs.synth(hs.SYNTH_RUNS, 1, 9, size)
dst = torch.empty(hs.container_bound(size, block), dtype=torch.uint8, device="cuda")
ws = torch.full((hs.workspace_size(size, block, key),), 0xFF, dtype=torch.uint8, device="cuda")
hs.compress_async(key, src, dst, block, workspace=ws); torch.cuda.synchronize()
info = hs.container_info(dst)
eager = dst[: info.totalSize].clone()
if mode == "eager_zero_dst":
    for rep in range(3):
        ws.fill_(0xFF); dst.zero_()
        hs.compress_async(key, src, dst, block, workspace=ws); torch.cuda.synchronize()
        print(mode, rep, torch.equal(dst[: info.totalSize], eager), flush=True)
else:
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            hs.compress_async(key, src, dst, block, workspace=ws)
    torch.cuda.current_stream().wait_stream(side)
    for rep in range(3):
        ws.fill_(0xFF)
        if mode == "graph_zero_dst": dst.zero_()
        torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        print(mode, rep, torch.equal(dst[: info.totalSize], eager), flush=True)
