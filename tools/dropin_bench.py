#!/usr/bin/env python3
"""The rle.h drop-in path with HOST pointers, as an integrator sees it (PCIe included): decode of a reference-minted monolithic stream
through `rle8_packed_decompress` / `rle64_3symlut_byte_decompress`, next to the compiled reference's CPU decoder on one core of this host,
and the reference's own `hsrlekit` (src/main.c, unmodified) linked against the GPU library on a 64 MiB file.

    python tools/dropin_bench.py      -> one line per case (GPU box)"""
import ctypes
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "hypersonic-rle-kit_amd", "python"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import hsrle
from hsrle_testlib import CODEC_BY_KEY, REF_SO, Oracle

ora = Oracle()
ref = ctypes.CDLL(REF_SO) if os.path.exists(REF_SO) else None
for key, kind, size in (("rle8_packed_multi", 0, 1 << 30), ("rle8_packed_multi", 1, 88473600), ("rle64_3symlut_byte", 1, 88473600)):
    codec = CODEC_BY_KEY[key]
    data = ora.synth(kind, codec.S, 2, size)
    stream = ora.compress(codec, data.tobytes())
    out = ctypes.create_string_buffer(size + 256)
    f = getattr(hsrle.lib(), codec.dname)
    f.restype = ctypes.c_uint32
    f.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
    best = None
    for _ in range(4):
        t0 = time.perf_counter()
        got = f(stream, len(stream), out, size)
        dt = time.perf_counter() - t0
        assert got == size
        best = dt if best is None else min(best, dt)
    ok = out.raw[:size] == data.tobytes()
    line = f"{codec.dname:32s} U {size >> 20:5d} MiB  GPU library, host pointers (H2D + index + decode + D2H): {size / 2**30 / best:7.2f} GiB/s ({best * 1e3:.1f} ms) exact {ok}"
    if ref is not None:
        g = getattr(ref, codec.dname)
        g.restype = ctypes.c_uint32
        g.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
        src = ctypes.create_string_buffer(stream + bytes(256), len(stream) + 256)
        bc = None
        for _ in range(3):
            t0 = time.perf_counter()
            assert g(src, len(stream), out, size) == size
            dt = time.perf_counter() - t0
            bc = dt if bc is None else min(bc, dt)
        line += f" | compiled reference, 1 core: {size / 2**30 / bc:6.2f} GiB/s"
    print(line, flush=True)

exe = os.path.join(REPO, "oracle", "_ref", "hsrlekit_dropin")
if os.path.exists(exe):
    path = "/tmp/hsrle_sample_64m.bin"
    ora.synth(0, 1, 2, 64 << 20).tofile(path)
    r = subprocess.run([exe, path, "--extreme", "--x-size", "8", "--packed", "--not-short", "--multi", "--runs", "2", "--min-time", "0"], capture_output=True, text=True, timeout=900)
    for l in r.stdout.replace("\r", "\n").splitlines():
        if "Bit" in l and "|" in l or "Mode" in l:
            print("hsrlekit (reference main.c on the GPU library):", l.strip())
