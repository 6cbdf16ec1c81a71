#!/usr/bin/env python3
"""BASELINE config 3 (and friends): decode time of a block container with and without the split decode, device resident.

    python tools/split_bench.py [--codec rle64_3symlut_byte] [--synth video] [--size 88473600] [--block 4096] [--subs 4096,2048,1024,512,256]"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "hypersonic-rle-kit_amd", "python"))
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--codec", default="rle64_3symlut_byte")
    ap.add_argument("--synth", default="video")
    ap.add_argument("--size", type=int, default=88473600)
    ap.add_argument("--block", type=int, default=4096)
    ap.add_argument("--subs", default="4096,2048,1024,512,256,1,0")     # 1 = HSRLE_SPLIT_PACKET_LIST
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch
    import hsrle
    from hsrle_testlib import CODEC_BY_KEY

    codec = CODEC_BY_KEY[args.codec]
    src = hsrle.synth(hsrle.SYNTH_VIDEO if args.synth == "video" else hsrle.SYNTH_RUNS, codec.S, 3, args.size, device="cuda")
    container, info = hsrle.compress(args.codec, src, block_size=args.block)
    out = torch.empty(args.size, dtype=torch.uint8, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    print(f"{args.codec} {args.synth} {args.size} B, {info.blockCount} blocks of {args.block}, ratio {info.totalSize / args.size:.4f}")
    for sub in [int(x) for x in args.subs.split(",")]:
        ws = torch.empty(max(hsrle.split_workspace_size(info, None, sub), 16), dtype=torch.uint8, device="cuda")
        eff = hsrle.lib().hsrle_split_sub_block_size(info, sub)
        run = lambda: hsrle.decompress_split_async(container, info, out, ws, status, sub_block=sub)
        out.zero_()
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        ok = int(status.item()) == 0 and torch.equal(out, src)
        print(f"  sub-block {sub:5d} (-> {eff:5d}): {ms * 1e3:8.1f} us  {args.size / 2**30 / (ms * 1e-3):8.1f} GiB/s  {(args.size + info.totalSize) / (ms * 1e-3) / 1e12:6.3f} TB/s algorithmic  exact {ok}", flush=True)
    # one wave per block (hsrle_decompress_wave_dev_async)
    if args.block <= 16384 and hsrle.experiments_enabled():
        run = lambda: hsrle.decompress_wave_async(container, info, out, status)
        out.zero_()
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        ok = int(status.item()) == 0 and torch.equal(out, src)
        print(f"  wave per block          : {ms * 1e3:8.1f} us  {args.size / 2**30 / (ms * 1e-3):8.1f} GiB/s  {(args.size + info.totalSize) / (ms * 1e-3) / 1e12:6.3f} TB/s algorithmic  exact {ok}", flush=True)


if __name__ == "__main__":
    main()
