#!/usr/bin/env python3
"""Request statistics of the decoder's top-up (diagnostic build -DHSRLE_REQ_STATS, variants/libhsrle_reqstats.so): how many load
instructions x rows asked for a WHOLE 128-byte line and how many for a (part of a) 64-byte half, for the headline workload.
usage (GPU box): HSRLE_LIB=variants/libhsrle_reqstats.so python tools/req_stats.py"""
import json, os, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "hypersonic-rle-kit_amd", "python"))
import torch
import hsrle

size, block = 8 << 30, 4096
src = hsrle.synth(hsrle.SYNTH_RUNS, 1, 2, size, device="cuda")
container, info = hsrle.compress("rle8_packed_multi", src, block_size=block)
out = torch.empty(size, dtype=torch.uint8, device="cuda")
status = torch.zeros(64, dtype=torch.int32, device="cuda")
hsrle.decompress_async(container, info, out, status)
torch.cuda.synchronize()
dbg = status[16:].view(torch.int64).cpu().tolist()
full, half, both, one = dbg[0], dbg[1], dbg[2], dbg[3]
print(json.dumps({"container_bytes": int(info.totalSize), "payload_bytes": int(info.payloadSize), "full_line_requests": full, "half_line_requests": half,
                  "partial_requests_touching_both_halves": both, "partial_requests_in_one_half": one, "group_requests": full + both + one,
                  "group_model_fetch_bytes": 128 * (full + both) + 64 * one, "predicted_fabric_requests": full + half, "predicted_fetch_bytes": 128 * full + 64 * half, "predicted_FETCH_SIZE_raw_bytes": 64 * (full + half),
                  "status": int(status[0].item()), "exact": bool(torch.equal(out, src))}))
