#!/usr/bin/env python3
"""Throughput of the monolithic-stream decode (hsrle_decompress_mono_dev), stream and output resident in device memory.

    python tools/mono_bench.py [--cases NAME,...] [--reps N] [--block B --region G --lookback M]

The stream of every case is written by the oracle's (= the reference's) encoder on the host; the decode is checked against the input.
Prints one line per case: GiB/s of uncompressed bytes, ms, index statistics (regions, repair rounds, regions walked again, look-back)."""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "hypersonic-rle-kit_amd", "python"))
sys.path.insert(0, os.path.join(REPO, "tests"))

CASES = {
    "packed8_runs_1g": ("rle8_packed_multi", 0, 1 << 30),
    "packed8_runs_64m": ("rle8_packed_multi", 0, 64 << 20),
    "packed8_video_88m": ("rle8_packed_multi", 1, 88473600),
    "lut64_video_88m": ("rle64_3symlut_byte", 1, 88473600),
    "lut8_runs_256m": ("rle8_3symlut", 0, 256 << 20),
    "plain8_runs_256m": ("rle8_multi", 0, 256 << 20),
    "sympacked16_runs_256m": ("rle16_sym_packed", 0, 256 << 20),
    "short32_video_256m": ("rle32_7symlut_byte_short", 1, 256 << 20),
    "lut8_runs_1g": ("rle8_3symlut", 0, 1 << 30),
    "lut16_7_runs_256m": ("rle16_7symlut_sym", 0, 256 << 20),
    "short8_1_video_256m": ("rle8_1symlut_short", 1, 256 << 20),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="packed8_runs_64m,packed8_video_88m,lut64_video_88m,lut8_runs_256m,sympacked16_runs_256m,short32_video_256m,packed8_runs_1g")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--region", type=int, default=0)
    ap.add_argument("--lookback", type=int, default=0)
    args = ap.parse_args()
    import torch
    import hsrle
    from hsrle_testlib import CODEC_BY_KEY, Oracle

    ora = Oracle()
    hsrle.mono_tuning(args.block, args.region, args.lookback)
    for name in args.cases.split(","):
        key, kind, size = CASES[name]
        codec = CODEC_BY_KEY[key]
        data = ora.synth(kind, codec.S, 2, size)
        t0 = time.time()
        stream = ora.compress(codec, data.tobytes())
        enc_s = time.time() - t0
        t = torch.zeros(len(stream) + 64, dtype=torch.uint8, device="cuda")
        t[: len(stream)] = torch.frombuffer(bytearray(stream), dtype=torch.uint8).cuda()
        out = torch.empty(size, dtype=torch.uint8, device="cuda")
        ws = torch.empty(hsrle.lib().hsrle_decompress_mono_workspace_size(hsrle.codec_id(key), size, len(stream)), dtype=torch.uint8, device="cuda")
        best, stats = None, None
        for _ in range(args.reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res, stats = hsrle.mono_decompress_dev(key, t, dst=out, workspace=ws, return_stats=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        ok = torch.equal(res.cpu(), torch.from_numpy(data))
        enc = ""
        if hsrle.lib().hsrle_compress_mono_workspace_size(hsrle.codec_id(key), size):
            src = torch.from_numpy(data).cuda()
            ews = torch.empty(hsrle.lib().hsrle_compress_mono_workspace_size(hsrle.codec_id(key), size), dtype=torch.uint8, device="cuda")
            edst = torch.empty(hsrle.compress_bounds(size) + 64, dtype=torch.uint8, device="cuda")
            eb = None
            for _ in range(args.reps + 1):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                st, chunks = hsrle.mono_compress_dev(key, src, dst=edst, workspace=ews, return_chunks=True)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                eb = dt if eb is None else min(eb, dt)
            same = st.cpu().numpy().tobytes() == stream
            lst = f" list rounds/again/again/rejected {hsrle.mono_encode_stats()}" if "symlut" in key else ""
            enc = f"  | mono ENCODE {size / 2**30 / eb:7.1f} GiB/s {eb * 1e3:8.3f} ms chunks {chunks} identical {same}{lst}"
        print(f"{name:24s} {key:28s} U {size >> 20:5d} MiB C/U {len(stream) / size:.3f}  {size / 2**30 / best:8.1f} GiB/s  {best * 1e3:8.3f} ms  regions {stats[0]} rounds {stats[1]} rewalked {stats[2]} lookback {stats[3]}"
              f"  exact {ok}{enc}  (host encode {enc_s:.1f} s)", flush=True)


if __name__ == "__main__":
    main()
