#!/usr/bin/env python3
"""Device-side time of the many-lane monolithic ENCODE (hsrle_compress_mono_dev) of the 1 GiB synthetic buffers, and the stream against the
oracle's on a 64 MiB prefix:  python tools/mono_enc_bench.py [codec,...] [GiB]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "hypersonic-rle-kit_amd", "python"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
import hsrle
from hsrle_testlib import CODEC_BY_KEY, Oracle

keys = sys.argv[1].split(",") if len(sys.argv) > 1 else ["rle8_packed_multi", "rle8_single", "rle8_packed_single", "rle128_byte_packed", "rle128_sym"]
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
n = int(gib * (1 << 30))
ora = Oracle()
L = hsrle.lib()
if os.environ.get("MONO_G"):
    hsrle.mono_tuning(0, int(os.environ["MONO_G"]), 0)                  # piece size of the cut finder (test knob)
for key in keys:
    codec = CODEC_BY_KEY[key]
    for kind in (0, 1):
        src = hsrle.synth(kind, codec.S, 2, n, device="cuda")
        small = src[: 64 << 20]
        got = hsrle.mono_compress_dev(key, small).cpu().numpy().tobytes()
        want = ora.compress(codec, small.cpu().numpy().tobytes())
        same = got == want
        ews = torch.empty(L.hsrle_compress_mono_workspace_size(hsrle.codec_id(key), n), dtype=torch.uint8, device="cuda")
        edst = torch.empty(hsrle.compress_bounds(n) + 64, dtype=torch.uint8, device="cuda")
        best = None
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            stream = hsrle.mono_compress_dev(key, src, dst=edst, workspace=ews)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print(f"{key:24s} kind {kind}: {best * 1e3:8.3f} ms  {gib / best:8.1f} GiB/s  ratio {stream.numel() / n:.4f}  64 MiB prefix == oracle: {same}", flush=True)
        del src, ews, edst
