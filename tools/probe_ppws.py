#!/usr/bin/env python3
"""Windowed position-parallel encoder of the 1 .. 8 byte symbol codecs (csrc/hsrle_encodeSpw.hip.h) against the oracle: containers of blocks above 4 KiB, every block
stream, on the data shapes of tests/test_gpu_pp.py.  GPU box:  python tools/probe_ppws.py [key prefix,...] [MiB per case]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "hypersonic-rle-kit_amd", "python"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import torch
import hsrle
from hsrle_testlib import CODECS, CODEC_BY_KEY, Oracle
from test_gpu_pp import _cases, _periodic, _edge_runs

prefixes = sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] else None
mib = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
n = int(mib * (1 << 20))
ora = Oracle()
rng = np.random.default_rng(424242)
base = _cases()
cases = {
    "periods": _periodic(rng, n, [1, 2, 3, 4, 6, 8, 12, 16], 256, 40, [4, 5, 6, 7, 8, 9, 11, 12, 13, 16, 17, 18, 19, 23, 24, 25, 40, 64, 100, 300, 2000, 8000]),
    "butting": _periodic(rng, n, [2, 3, 4, 6, 8], 3, 0, [4, 6, 7, 8, 9, 12, 13, 14, 16, 17, 20, 24, 25, 33]),
    "far_apart": _periodic(rng, n, [2, 3, 4, 6, 8], 256, 700, [8, 12, 16, 19, 24, 36, 48]),
    "two_symbols": rng.integers(0, 2, n, dtype=np.uint8),
    "mixed": base["mixed"][:n],
    "zeros": np.zeros(n, dtype=np.uint8),
    "edges": _edge_runs(rng, n),
    "long_literals": _periodic(rng, n, [2, 3, 4, 6, 8], 256, 8000, [8, 12, 16, 19, 24, 36, 48, 7000]),
}
PATH_PP = 3
keys = [c.key for c in CODECS if hsrle.lib().hsrle_encode_path(hsrle.codec_id(c.key), 1 << 26, 8192) == PATH_PP]
if prefixes:
    keys = [k for k in keys if any(k.startswith(p) for p in prefixes)]
print(len(keys), "windowed codecs:", " ".join(keys), flush=True)
bad = 0
for key in keys:
    codec = CODEC_BY_KEY[key]
    for B, cut in ((4224, 0), (8192, 777), (65536, 4097), (1 << 19, 3)):
        for name, data in cases.items():
            part = data[: data.size - cut]
            src = torch.from_numpy(part).cuda()
            container, info = hsrle.compress(key, src, block_size=B)
            cinfo, streams = hsrle.split_container(container.cpu().numpy().tobytes())
            expect = ora.compress_blocks(codec, part, B)
            wrong = [i for i, (a, b) in enumerate(zip(streams, expect)) if a != b]
            rt = torch.equal(hsrle.decompress(container), src)
            ok = not wrong and len(streams) == len(expect) and rt
            if not ok:
                bad += 1
                i = wrong[0] if wrong else -1
                extra = ""
                if i >= 0:
                    a, b = streams[i], expect[i]
                    fd = next((k for k, (u, v) in enumerate(zip(a, b)) if u != v), min(len(a), len(b)))
                    extra = f" block {i} sizes {len(a)} {len(b)} first diff at {fd}"
                print(f"MISMATCH {key:28s} B {B:7d} {name:14s} wrong {len(wrong)} of {len(expect)} roundtrip {rt}{extra}", flush=True)
    print(key, "done, failures so far", bad, flush=True)
print("failures", bad)
sys.exit(1 if bad else 0)
