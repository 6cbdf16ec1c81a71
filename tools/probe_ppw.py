#!/usr/bin/env python3
"""Windowed position-parallel 8 bit encoder (csrc/hsrle_encode8pw.hip.h) against the oracle: containers of blocks above 4 KiB (every block stream) and monolithic
streams, on the data shapes of tests/test_gpu_pp.py.  GPU box:  python tools/probe_ppw.py [blocks|mono|all]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "hypersonic-rle-kit_amd", "python"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import torch
import hsrle
from hsrle_testlib import CODEC_BY_KEY, Oracle
from test_gpu_pp import _cases

what = sys.argv[1] if len(sys.argv) > 1 else "all"
ora = Oracle()
cases = _cases()
bad = 0
if what in ("blocks", "all"):
    for B in (4224, 8192, 12416, 65536):
        need = 4096 * B
        for name, data in cases.items():
            big = np.resize(data, need + 777)                  # (ragged last block)
            src = torch.from_numpy(big).cuda()
            for key in ("rle8_multi", "rle8_packed_multi"):
                codec = CODEC_BY_KEY[key]
                path = hsrle.lib().hsrle_encode_path(hsrle.codec_id(key), big.size, B)
                container, info = hsrle.compress(key, src, block_size=B)
                cinfo, streams = hsrle.split_container(container.cpu().numpy().tobytes())
                expect = ora.compress_blocks(codec, big, B)
                wrong = [i for i, (a, b) in enumerate(zip(streams, expect)) if a != b]
                rt = torch.equal(hsrle.decompress(container), src)
                ok = not wrong and len(streams) == len(expect) and rt
                bad += 0 if ok else 1
                print(f"B {B:6d} {name:14s} {key:18s} path {path} blocks {len(expect)} {'ok' if ok else 'MISMATCH ' + str(len(wrong)) + ' first ' + str(wrong[:6]) + ' roundtrip ' + str(rt)}", flush=True)
if what in ("mono", "all"):
    for name, data in cases.items():
        for n in (data.size, 1 << 20, 123457, 40000):
            part = data[:n]
            src = torch.from_numpy(part).cuda()
            for key in ("rle8_multi", "rle8_packed_multi"):
                got = hsrle.mono_compress_dev(key, src).cpu().numpy().tobytes()
                want = ora.compress(CODEC_BY_KEY[key], part.tobytes())
                ok = got == want
                bad += 0 if ok else 1
                first = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), min(len(got), len(want)))
                print(f"mono n {n:8d} {name:14s} {key:18s} {'ok' if ok else 'MISMATCH sizes %d %d first diff at %d' % (len(got), len(want), first)}", flush=True)
print("failures", bad)
sys.exit(1 if bad else 0)
