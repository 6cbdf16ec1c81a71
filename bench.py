#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size-gib G] [--block B] [--codec NAME]

Workload (N = 1): `rle8_packed_multi` -- 8 bit Packed, the north-star codec -- on an 8 GiB run-distributed(8) synthetic buffer
(BASELINE.json: "8-bit Packed decode of an 8 GiB synthetic buffer at 1 GPU"), cut into 4 KiB blocks that are each a
complete reference stream.  A STEP is one decode of the whole container, device resident (compressed container and output both in
HBM).  `value` = uncompressed GiB decoded per second (the reference's own unit, src/main.c:875,:1014); the encode throughput of
the same buffer is reported next to it ("encode").  For N > 1 every rank owns its own 8 GiB shard (weak scaling, block-sharded,
no data-path collective: SURVEY.md §8e); the RCCL gather of the compressed segments into one stream is timed separately
("gather_ms") because it is not part of the decode path.

The JSON line also carries
  roofline      HBM roofline of the decode kernel: algorithmic bytes (compressed + uncompressed) / average kernel
                duration measured with HIP events on the launch stream, against the 8 TB/s peak
  cpu_baseline  the same decode on the host CPU, one thread, on a bounded sample of the same workload: the compiled reference
                (oracle/_ref, kind "reference") when it is present, else the oracle's restatement (kind "port"); encode.cpu_baseline is
                the encode direction of the same sample (BASELINE metric: "decode + encode GiB/s ... vs CPU")
  bit_exact     block streams of a sample equal the CPU codec's streams and the full decode equals the input
"""
import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "hypersonic-rle-kit_amd", "python"))
sys.path.insert(0, os.path.join(REPO, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def measured_traffic(codec, size, block, build_id):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this workload AND THIS BUILD (profiles/r*_traffic.json, written by
    tools/traffic.sh, stamped with the library's build id; a file of another build or workload is refused: traffic = null).  FETCH_SIZE
    tallies every L2 -> fabric read request at 64 bytes whether it asks for 64 or 128 (calibrated with tools/ubench/fetch_calib.hip on a
    buffer of known size, profiles/r02_fetch_calibration.txt), so the read side is known between two bounds: every chunk at least once
    (the container) and 128 bytes per counted request.  Returns (decode upper bound, decode detail, encode upper bound)."""
    import glob

    stale = None
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_traffic.json")), reverse=True):
        try:
            t = json.load(open(path))
            if not (t["codec"] == codec and t["size"] == size and t["block"] == block):
                continue
            if t.get("library_build_id") != build_id:
                stale = stale or os.path.basename(path)
                continue
            detail = {"fetch_size_raw": t["fetch_size_raw_bytes"], "fabric_read_requests": t["fabric_read_requests"],
                      "fetch_bounds": [t["fetch_lower_bound_bytes"], t["fetch_upper_bound_bytes"]], "write_size": t["write_bytes"],
                      "traffic_bounds": [t["traffic_lower_bound_bytes"], t["traffic_upper_bound_bytes"]], "launches": t.get("launches"),
                      "library_build_id": build_id, "source": "profiles/" + os.path.basename(path)}
            enc = t.get("encode", {}).get("traffic_bounds")
            return int(t["traffic_upper_bound_bytes"]), detail, (int(enc[1]) if enc else None)
        except Exception:
            continue
    return None, ({"refused": f"profiles/{stale} was measured on another build of the library (this one: {build_id}); rerun tools/traffic.sh"} if stale else None), None


def cpu_baseline(container_prefix, n_blocks, block_size, codec_key, expect, budget_s=12.0, all_cores=True):
    """Decode the first n_blocks of the container on the host (bounded sample), single thread."""
    import numpy as np
    from hsrle_testlib import CODEC_BY_KEY, REF_SO, Oracle

    codec = CODEC_BY_KEY[codec_key]
    raw = container_prefix
    offs = np.frombuffer(raw, dtype=np.uint64, count=n_blocks + 1, offset=64).copy()
    p0 = 64 + 8 * (int(np.frombuffer(raw, dtype=np.uint32, count=1, offset=28)[0]) + 1)
    payload = np.frombuffer(raw, dtype=np.uint8, offset=p0)
    usize = n_blocks * block_size
    out = np.zeros(usize + 256, dtype=np.uint8)

    if os.path.exists(REF_SO):
        kind = "reference"
        lib = ctypes.CDLL(REF_SO)
        fn = ctypes.cast(getattr(lib, codec.dname), ctypes.c_void_p)
        lib.hsrle_ref_decode_blocks.restype = ctypes.c_uint64
        lib.hsrle_ref_decode_blocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64]
        run = lambda: lib.hsrle_ref_decode_blocks(fn, payload.ctypes.data, offs.ctypes.data, n_blocks, block_size, out.ctypes.data, usize)
    else:
        kind = "port"
        ora = Oracle()
        ora.lib.hso_decompress_blocks.restype = ctypes.c_uint64
        ora.lib.hso_decompress_blocks.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64]
        run = lambda: ora.lib.hso_decompress_blocks(codec.family, codec.S, codec.aligned, payload.ctypes.data, offs.ctypes.data, n_blocks, block_size, out.ctypes.data, usize)

    assert run() == usize  # warm-up run, discarded (reference protocol: src/main.c:822-887)
    best, total, reps = None, 0.0, 0
    t_end = time.time() + budget_s
    while reps < (3 if budget_s >= 4.0 else 1) or (time.time() < t_end and reps < 40):
        t0 = time.perf_counter()
        got = run()
        dt = time.perf_counter() - t0
        assert got == usize
        total += dt
        reps += 1
        best = dt if best is None else min(best, dt)
    ok = out[:usize].tobytes() == expect
    res = {"value": round(usize / 2**30 / (total / reps), 3), "best": round(usize / 2**30 / best, 3), "unit": "GiB/s", "cores": 1, "kind": kind,
           "sample": f"decode of the first {usize >> 20} MiB ({n_blocks} blocks) of the same container, {reps} runs, mean", "matches_gpu_input": bool(ok)}

    # context, not the baseline: the same decode spread over all host cores by the shim's pthread loop (one contiguous block range
    # per thread).  SURVEY.md §8d "CPU beside it": 1-thread and N-thread.
    if all_cores and kind == "reference" and hasattr(lib, "hsrle_ref_decode_blocks_mt"):
        nthreads = max(1, min(os.cpu_count() or 1, 1024))
        lib.hsrle_ref_decode_blocks_mt.restype = ctypes.c_uint64
        lib.hsrle_ref_decode_blocks_mt.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int]
        # the reference writes up to 128 bytes past a block's end: thread t's range sits at + 256 * t (own slack per range)
        out2 = np.zeros(usize + 256 * (nthreads + 1), dtype=np.uint8)
        runmt = lambda: lib.hsrle_ref_decode_blocks_mt(fn, payload.ctypes.data, offs.ctypes.data, n_blocks, block_size, out2.ctypes.data, nthreads)
        if runmt() == usize:  # warm-up
            tb, t_stop, runs = None, time.time() + 4.0, 0
            while runs < 3 or (time.time() < t_stop and runs < 50):
                t0 = time.perf_counter()
                got = runmt()
                dt = time.perf_counter() - t0
                runs += 1
                if got == usize:
                    tb = dt if tb is None else min(tb, dt)
            bounds = [n_blocks * t // nthreads for t in range(nthreads + 1)]
            same = all(out2[bounds[t] * block_size + 256 * t : bounds[t + 1] * block_size + 256 * t].tobytes() == expect[bounds[t] * block_size : bounds[t + 1] * block_size] for t in range(nthreads))
            if tb is not None and same:
                res["all_cores"] = {"value": round(usize / 2**30 / tb, 2), "unit": "GiB/s", "cores": nthreads, "note": "best of %d runs, one block range per POSIX thread" % runs}
    return res


def cpu_encode_baseline(sample, block_size, codec_key, gpu_payload_prefix, gpu_offsets, budget_s=10.0):
    """Encode the sample (the first bytes of the same input) block by block on the host, single thread: the compiled reference (kind
    "reference") when it is present, else the oracle's restatement ("port").  The streams must be the GPU's."""
    import numpy as np
    from hsrle_testlib import CODEC_BY_KEY, REF_SO, Oracle

    codec = CODEC_BY_KEY[codec_key]
    n = sample.size
    nb = (n + block_size - 1) // block_size
    src = np.zeros(n + 64, dtype=np.uint8)                                   # guard pad behind the sample: bytes that never match
    src[:n] = sample
    if os.path.exists(REF_SO):
        kind = "reference"
        lib = ctypes.CDLL(REF_SO)
        lib.rle_compress_bounds.restype = ctypes.c_uint32
        stride = (lib.rle_compress_bounds(block_size) + 15) & ~15
        fn = ctypes.cast(getattr(lib, codec.cname), ctypes.c_void_p)
        # symbols of 2 .. 16 bytes: every block from a guard-padded private copy (the reference over-reads its input; the copy is part of the time)
        enc = lib.hsrle_ref_encode_blocks_guarded if (codec.S > 1 and hasattr(lib, "hsrle_ref_encode_blocks_guarded")) else lib.hsrle_ref_encode_blocks
        enc.restype = ctypes.c_uint64
        enc.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        slots = np.zeros(nb * stride + 64, dtype=np.uint8)
        sizes = np.zeros(nb, dtype=np.uint32)
        run = lambda: enc(fn, src.ctypes.data, n, block_size, slots.ctypes.data, stride, sizes.ctypes.data)
    else:
        kind = "port"
        ora = Oracle()
        stride = (block_size + 193 + 15) & ~15
        slots = np.zeros(nb * stride + 64, dtype=np.uint8)
        sizes = np.zeros(nb, dtype=np.uint32)
        ora.lib.hso_compress_blocks.restype = ctypes.c_uint32
        ora.lib.hso_compress_blocks.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        run = lambda: ora.lib.hso_compress_blocks(codec.family, codec.S, codec.aligned, src.ctypes.data, n, block_size, slots.ctypes.data, stride, sizes.ctypes.data)
    assert run() == nb  # warm-up run, discarded
    best, total, reps = None, 0.0, 0
    t_end = time.time() + budget_s
    while reps < (3 if budget_s >= 4.0 else 1) or (time.time() < t_end and reps < 20):
        t0 = time.perf_counter()
        got = run()
        dt = time.perf_counter() - t0
        assert got == nb
        total += dt
        reps += 1
        best = dt if best is None else min(best, dt)
    same = bool((np.diff(gpu_offsets.astype(np.int64)) == sizes.astype(np.int64)).all())
    for i in list(range(0, nb, max(1, nb // 4096))) + [nb - 1]:              # every block's size, a few thousand blocks' bytes
        if not same:
            break
        same = slots[i * stride : i * stride + int(sizes[i])].tobytes() == gpu_payload_prefix[int(gpu_offsets[i]) : int(gpu_offsets[i + 1])]
    return {"value": round(n / 2**30 / (total / reps), 3), "best": round(n / 2**30 / best, 3), "unit": "GiB/s", "cores": 1, "kind": kind,
            "sample": f"encode of the first {n >> 20} MiB ({nb} blocks) of the same buffer, block by block, {reps} runs, mean", "streams_match_gpu": same}


def side_measurements(hsrle, torch, src, dev):
    """Not `value`: (1) BASELINE config 3 -- rle64_3symlut_byte on the 88 473 600-byte video-shaped frame in 4 KiB blocks, decoded plain
    and split (one lane per 1 KiB sub-block; the container is untouched); (2) BASELINE config 2 as ONE monolithic stream: the first
    1 GiB of the headline buffer written by the many-lane encoder and read back through the entry-point index."""
    import ctypes

    def timed(fn, reps):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    out = {}
    fsize = 88473600
    frame = hsrle.synth(hsrle.SYNTH_VIDEO, 8, 2, fsize, device=dev)
    cont, info = hsrle.compress("rle64_3symlut_byte", frame, block_size=4096)
    dec = torch.empty(fsize, dtype=torch.uint8, device=dev)
    st = torch.zeros(16, dtype=torch.int32, device=dev)
    ws = torch.empty(max(hsrle.split_workspace_size(info, None, 0), 16), dtype=torch.uint8, device=dev)
    cdst = torch.empty(hsrle.container_bound(fsize, 4096), dtype=torch.uint8, device=dev)
    cws = torch.empty(hsrle.workspace_size(fsize, 4096), dtype=torch.uint8, device=dev)
    enc_frame_ms = timed(lambda: hsrle.compress_async("rle64_3symlut_byte", frame, cdst, 4096, workspace=cws), 10)
    del cdst, cws
    plain_ms = timed(lambda: hsrle.decompress_async(cont, info, dec, st), 10)
    plain_ok = int(st[0].item()) == 0 and torch.equal(dec, frame)
    dec.zero_()
    split_ms = timed(lambda: hsrle.decompress_split_async(cont, info, dec, ws, st, sub_block=0), 10)
    ok = plain_ok and int(st[0].item()) == 0 and torch.equal(dec, frame)
    mode = int(hsrle.lib().hsrle_split_sub_block_size(info, 0))
    out["config3_frame"] = {"codec": "rle64_3symlut_byte", "bytes": fsize, "block_size": 4096, "ratio": round(info.totalSize / fsize, 4), "encode_us": round(enc_frame_ms * 1e3, 1), "decode_us": round(plain_ms * 1e3, 1),
                            "split_decode_us": round(split_ms * 1e3, 1), "split_GiBps": round(fsize / 2**30 / (split_ms * 1e-3), 1),
                            "split_algorithmic_TBps": round((fsize + info.totalSize) / (split_ms * 1e-3) / 1e12, 3),
                            "split_mode": "packet list (one entry per packet, then 16 output bytes per lane)" if mode == 1 else f"records every {mode} bytes", "exact": bool(ok)}
    del frame, cont, dec, ws

    n = 1 << 30
    part = src[:n]
    stream = hsrle.mono_compress_dev("rle8_packed_multi", part)
    t = torch.zeros(stream.numel() + 64, dtype=torch.uint8, device=dev)
    t[: stream.numel()] = stream
    L = hsrle.lib()
    ews = torch.empty(L.hsrle_compress_mono_workspace_size(1, n), dtype=torch.uint8, device=dev)
    edst = torch.empty(hsrle.compress_bounds(n) + 64, dtype=torch.uint8, device=dev)
    dws = torch.empty(L.hsrle_decompress_mono_workspace_size(1, n, stream.numel()), dtype=torch.uint8, device=dev)
    dout = torch.empty(n, dtype=torch.uint8, device=dev)
    import time

    def wall(fn, reps):
        fn(); torch.cuda.synchronize()
        best = None
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best * 1e3

    enc_ms = wall(lambda: hsrle.mono_compress_dev("rle8_packed_multi", part, dst=edst, workspace=ews), 3)
    dec_ms = wall(lambda: hsrle.mono_decompress_dev("rle8_packed_multi", t, dst=dout, workspace=dws), 3)
    mono_exact = bool(torch.equal(dout, part))
    # the same stream through hsrle_decompress_mono_dev_async (nothing between the passes waits for the host), called and as a replayed HIP graph
    head16 = t[:16].cpu().numpy().tobytes()
    mstatus = torch.full((1,), 77, dtype=torch.int32, device=dev)
    dout.zero_()
    async_ms = wall(lambda: hsrle.mono_decompress_dev_async("rle8_packed_multi", t, head16, dout, dws, mstatus), 5)
    async_ok = int(mstatus.item()) == hsrle.MONO_DONE and bool(torch.equal(dout, part))
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        hsrle.mono_decompress_dev_async("rle8_packed_multi", t, head16, dout, dws, mstatus)
    dout.zero_(); mstatus.fill_(77)
    graph_ms = wall(graph.replay, 5)
    graph_ok = int(mstatus.item()) == hsrle.MONO_DONE and bool(torch.equal(dout, part))
    del graph
    from hsrle_testlib import big_manifest

    import hashlib

    man = big_manifest()
    want = man["cases"]["config2_1GiB"]["mono"] if man else None
    same = None
    if want is not None:
        same = stream.numel() == want["size"] and hashlib.sha256(stream.cpu().numpy().data).hexdigest() == want["sha256"]
    out["mono_1GiB"] = {"codec": "rle8_packed_multi", "stream_bytes": int(stream.numel()), "encode_ms": round(enc_ms, 3), "encode_GiBps": round(1024 / enc_ms, 1),
                        "decode_ms": round(dec_ms, 3), "decode_GiBps": round(1024 / dec_ms, 1),
                        "decode_frac": round((n + int(stream.numel())) / (dec_ms * 1e-3) / 8e12, 4), "encode_frac": round((n + int(stream.numel())) / (enc_ms * 1e-3) / 8e12, 4),
                        "decode_async_ms": round(async_ms, 3), "decode_async_exact": async_ok, "decode_graph_replay_ms": round(graph_ms, 3), "decode_graph_exact": graph_ok,
                        "stream_is_the_references": same, "decode_exact": mono_exact,
                        "note": "one monolithic reference stream, device resident; decode_ms = hsrle_decompress_mono_dev (header read, one verdict read at the end), "
                                "decode_async_ms = hsrle_decompress_mono_dev_async + one synchronize, decode_graph_replay_ms = the same call captured in a HIP graph; "
                                "frac = (C + U) / t against 8 TB/s"}

    # (3) the HOST-pointer drop-in path, as a caller of the reference gets it without changing a line (src/main.c:835, :970 call these names): the
    # same 1 GiB through rle8_packed_multi_compress / rle8_packed_decompress with numpy buffers -- H2D + kernels + D2H, PCIe-bound by construction
    # (never `value`) -- next to what the link itself gives on this box (pinned copies of the same byte counts: H2D and D2H share ~57 GB/s here,
    # so the ceiling is the SUM of the two transfers)
    import numpy as np

    host_in = part.cpu().numpy()
    cap = hsrle.compress_bounds(n)
    host_stream = np.zeros(cap, dtype=np.uint8)          # (touched: first-touch page faults are not the library's)
    host_out = np.zeros(n, dtype=np.uint8)
    cfn, dfn = L.rle8_packed_multi_compress, L.rle8_packed_decompress
    for f in (cfn, dfn):
        f.restype = ctypes.c_uint32
        f.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]

    def host_wall(fn, reps):
        best, got = None, 0
        for _ in range(reps + 1):
            t0 = time.perf_counter(); got = fn(); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best, got

    c_s, csize = host_wall(lambda: cfn(host_in.ctypes.data, n, host_stream.ctypes.data, cap), 2)
    d_s, dsize = host_wall(lambda: dfn(host_stream.ctypes.data, csize, host_out.ctypes.data, n), 2)
    pin_a = torch.empty(n, dtype=torch.uint8).pin_memory()
    dev_a = torch.empty(n, dtype=torch.uint8, device=dev)

    def copy_rate(dst, srcb, nbytes):
        best = None
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); dst[:nbytes].copy_(srcb[:nbytes], non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best

    h2d_u, d2h_u = copy_rate(dev_a, pin_a, n), copy_rate(pin_a, dev_a, n)
    h2d_c, d2h_c = copy_rate(dev_a, pin_a, int(csize)), copy_rate(pin_a, dev_a, int(csize))
    out["dropin_host"] = {"functions": "rle8_packed_multi_compress / rle8_packed_decompress (rle.h names, host pointers, one monolithic stream)", "bytes": n, "stream_bytes": int(csize),
                          "encode_GiBps": round(n / 2**30 / c_s, 2), "decode_GiBps": round(n / 2**30 / d_s, 2), "encode_ms": round(c_s * 1e3, 2), "decode_ms": round(d_s * 1e3, 2),
                          "link_pinned_H2D_GBps": round(n / h2d_u / 1e9, 1), "link_pinned_D2H_GBps": round(n / d2h_u / 1e9, 1),
                          "decode_copy_ceiling_GiBps": round(n / 2**30 / (h2d_c + d2h_u), 2), "encode_copy_ceiling_GiBps": round(n / 2**30 / (h2d_u + d2h_c), 2),
                          "decode_frac_of_copy_ceiling": round((h2d_c + d2h_u) / d_s, 3), "encode_frac_of_copy_ceiling": round((h2d_u + d2h_c) / c_s, 3),
                          "stream_is_the_references": (None if want is None else bool(int(csize) == want["size"] and hashlib.sha256(host_stream[: int(csize)].data).hexdigest() == want["sha256"])),
                          "decode_exact": bool(int(dsize) == n and np.array_equal(host_out, host_in)),
                          "note": "ceiling = pinned H2D of the input side + pinned D2H of the output side, one after the other (the link is shared: both directions at once run at half rate each); the reference CPU rates for the same buffer are cpu_baseline / encode.cpu_baseline of this line"}
    return out


def config5_rows(hsrle, torch, dev, with_cpu):
    """BASELINE config 5 (src/main.c:803-1076 prints this table per codec on the CPU): for every codec that has a reference-minted manifest --
    8 GiB run-distributed(W, seed 5), every width x {Packed, 3LUT} + the 8 bit Single codecs -- encode and decode throughput, ratio, roofline
    fraction ((C + U) / t against 8 TB/s), all 2 097 152 block streams against the compiled reference's roll-ups, and the reference CPU codec
    (one core) on the first 64 MiB of the same buffer beside it.  Not `value`."""
    import numpy as np
    from hsrle_testlib import CODEC_BY_KEY, big_manifest, big_case, rollups

    man = big_manifest()
    rows = []
    if not man:
        return rows
    names = sorted(k for k in man["cases"] if k.startswith("config5_"))
    dst = ws = out = None
    for name in names:
        e = man["cases"][name]
        codec = CODEC_BY_KEY[e["codec"]]
        size, block = e["size"], e["block"]
        src = hsrle.synth(e["kind"], codec.S, e["seed"], size, device=dev)
        if dst is None:
            dst = torch.empty(hsrle.container_bound(size, block), dtype=torch.uint8, device=dev)
            ws = torch.empty(hsrle.workspace_size(size, block), dtype=torch.uint8, device=dev)
            out = torch.empty(size, dtype=torch.uint8, device=dev)
        status = torch.zeros(16, dtype=torch.int32, device=dev)

        def timed(fn, warm, reps):
            for _ in range(warm):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        enc_ms = timed(lambda: hsrle.compress_async(e["codec"], src, dst, block, workspace=ws), 1, 3)
        info = hsrle.container_info(dst)
        container = dst[: info.totalSize]
        dec_ms = timed(lambda: hsrle.decompress_async(container, info, out, status), 2, 5)
        _, _, want = big_case(e["codec"], e["kind"], e["seed"], size, block)
        got = rollups(hsrle.hash_blocks(container, info).cpu().numpy())
        exact = (info.blockCount == e["blocks"] and info.payloadSize == e["payload_size"] and bool((got == want).all())
                 and int(status[0].item()) == 0 and torch.equal(out, src))
        alg = size + info.totalSize
        row = {"codec": e["codec"], "ratio": round(info.totalSize / size, 4), "decode_GiBps": round(size / 2**30 / (dec_ms * 1e-3), 1), "encode_GiBps": round(size / 2**30 / (enc_ms * 1e-3), 1),
               "decode_frac": round(alg / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "encode_frac": round(alg / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "bit_exact": bool(exact), "blocks_compared": int(info.blockCount)}
        if with_cpu:
            nb = (64 << 20) // block
            p0 = info.payload_start
            prefix_end = p0 + int(container[64 + 8 * nb : 64 + 8 * nb + 8].view(torch.int64).item()) + 64
            prefix = container[: min(prefix_end, container.numel())].cpu().numpy().tobytes()
            sample = src[: nb * block].cpu().numpy()
            cd = cpu_baseline(prefix, nb, block, e["codec"], sample.tobytes(), budget_s=0.4, all_cores=False)
            ce = cpu_encode_baseline(sample, block, e["codec"], prefix[p0:], np.frombuffer(prefix, dtype=np.uint64, count=nb + 1, offset=64), budget_s=0.4)
            row.update({"cpu_decode_GiBps": cd["value"], "cpu_encode_GiBps": ce["value"], "cpu_kind": cd["kind"], "cpu_cores": 1,
                        "cpu_streams_match_gpu": bool(ce["streams_match_gpu"] and cd["matches_gpu_input"])})
        rows.append(row)
        del src
    return rows


def video_rows(hsrle, torch, dev, with_cpu, kind=None, block=4096, keys=("rle8_packed_multi", "rle16_sym_packed", "rle32_3symlut_byte", "rle48_7symlut_byte", "rle64_3symlut_byte", "rle24_byte_short", "rle64_7symlut_byte_short")):
    """Side measurement (not `value`): decode and encode of the 8 GiB VIDEO-SHAPED buffer (many short packets, unevenly spread: what the capped decoder rounds of
    round 5 are for) for a few codecs of the families the 110-codec sweep has its weakest rows in.  No reference-minted manifests exist for
    these buffers: `exact` = the decode equals the input and the status word is 0; `cpu_streams_match_gpu` = the first 64 MiB of block streams equal the
    compiled reference's.  Round 6: the same rows for other shapes -- `kind` (None: video-shaped), `block`, `keys` -- serve `extras.large_blocks` (64 KiB blocks: the
    windowed position-parallel encoders)."""
    import numpy as np
    from hsrle_testlib import CODEC_BY_KEY

    rows = []
    size = 8 << 30
    kind = hsrle.SYNTH_VIDEO if kind is None else kind
    dst = torch.empty(hsrle.container_bound(size, block), dtype=torch.uint8, device=dev)
    ws = torch.empty(hsrle.workspace_size(size, block), dtype=torch.uint8, device=dev)
    out = torch.empty(size, dtype=torch.uint8, device=dev)
    for key in keys:
        codec = CODEC_BY_KEY[key]
        src = hsrle.synth(kind, codec.S, 5, size, device=dev)
        status = torch.zeros(16, dtype=torch.int32, device=dev)

        def timed(fn, warm, reps):
            for _ in range(warm):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        enc_ms = timed(lambda: hsrle.compress_async(key, src, dst, block, workspace=ws), 1, 3)
        info = hsrle.container_info(dst)
        container = dst[: info.totalSize]
        dec_ms = timed(lambda: hsrle.decompress_async(container, info, out, status), 2, 5)
        exact = int(status[0].item()) == 0 and torch.equal(out, src)
        alg = size + info.totalSize
        row = {"codec": key, "ratio": round(info.totalSize / size, 4), "decode_GiBps": round(size / 2**30 / (dec_ms * 1e-3), 1), "encode_GiBps": round(size / 2**30 / (enc_ms * 1e-3), 1),
               "decode_frac": round(alg / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "encode_frac": round(alg / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "exact": bool(exact)}
        if with_cpu:
            nb = (64 << 20) // block
            p0 = info.payload_start
            prefix_end = p0 + int(container[64 + 8 * nb : 64 + 8 * nb + 8].view(torch.int64).item()) + 64
            prefix = container[: min(prefix_end, container.numel())].cpu().numpy().tobytes()
            sample = src[: nb * block].cpu().numpy()
            cd = cpu_baseline(prefix, nb, block, key, sample.tobytes(), budget_s=0.4, all_cores=False)
            ce = cpu_encode_baseline(sample, block, key, prefix[p0:], np.frombuffer(prefix, dtype=np.uint64, count=nb + 1, offset=64), budget_s=0.4)
            row.update({"cpu_decode_GiBps": cd["value"], "cpu_encode_GiBps": ce["value"], "cpu_kind": cd["kind"], "cpu_cores": 1,
                        "cpu_streams_match_gpu": bool(ce["streams_match_gpu"] and cd["matches_gpu_input"])})
        rows.append(row)
        del src
    return rows


def kernel_name(codec_key):
    """The decode kernel instantiation behind a codec id (hsrle_decode.hip.h: k_decode_blocks<FAM, S, AL, T, R, Q, SGL>)."""
    from hsrle_testlib import CODEC_BY_KEY, FAMILY_NAMES

    c = CODEC_BY_KEY[codec_key]
    fam = FAMILY_NAMES.get(c.family, str(c.family)).upper()
    return f"k_decode_blocks<{fam},{c.S},{'sym' if c.aligned else 'byte'}>"


def spawn_ranks(n, result_fd):
    """Start n ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, the contract of torch.distributed.run)
    from a parent that never touches the GPU; rank 0's stdout (the one JSON line) is relayed to the parent's stdout."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # a rank that dies leaves the others waiting in a collective: poll, and end exactly the processes started here when one fails
    import threading

    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    codes = [p.wait() for p in procs]
    reader.join(timeout=10)
    if out0 and out0[0] and not failed:
        os.write(result_fd, out0[0])
    return 1 if failed else max(abs(c) for c in codes)


def main():
    # stdout carries exactly ONE line: the JSON result of rank 0.  Libraries print to fd 1 too (RCCL prints its version banner
    # there when a communicator is created), so everything else is sent to stderr for the lifetime of the process.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size-gib", type=float, default=8.0, help="uncompressed bytes per GPU")
    ap.add_argument("--block", type=int, default=4096)
    ap.add_argument("--codec", default="rle8_packed_multi")
    ap.add_argument("--synth", choices=["runs", "video"], default="runs", help="synthetic generator: run-distributed (headline) or video-shaped (BASELINE config 3)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements (config 3 frame, monolithic 1 GiB stream, config 5 rows)")
    ap.add_argument("--no-config5", action="store_true", help="skip the sixteen 8 GiB codec rows of BASELINE config 5 (extras.config5)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher.  It has not touched the GPU (no torch import,
        # no HIP call) and never will: it starts N children of this same script, one rank per GPU, relays rank 0's JSON line and exits
        # with the worst child's code.  (Children are ordinary child processes: nothing that has initialised the GPU is ever exec'd.)
        sys.exit(spawn_ranks(args.gpus, result_fd))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or without a launcher)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("HSRLE_BENCH_DRYRUN") == "1":   # launcher plumbing check (tests/test_bench_launcher.py): no GPU is touched
        if rank == 0:
            os.write(result_fd, (json.dumps({"dryrun": True, "n_gpus": world, "master": os.environ.get("MASTER_ADDR")}) + "\n").encode())
        return

    import torch

    assert torch.cuda.is_available(), "bench.py needs a GPU (the library has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1 or os.environ.get("HSRLE_FORCE_DIST") == "1"   # the env switch lets a 1-GPU box exercise the RCCL code path
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "MASTER_ADDR" not in os.environ:   # HSRLE_FORCE_DIST=1 on one GPU without a launcher: a world of one
            import socket

            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s.getsockname()[1]), RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=dev)

    import hsrle

    hsrle.lib()
    from hsrle_testlib import CODEC_BY_KEY, Oracle

    codec = CODEC_BY_KEY[args.codec]
    size = int(args.size_gib * (1 << 30)) // args.block * args.block
    seed = 2 if not distributed else 100 + rank  # config 2 seed 2; sharded config seeds 100 + rank

    synth_name = "run-distributed(%d)" % (8 * codec.S) if args.synth == "runs" else "video-shaped"

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- setup (untimed): synthetic input generated on the device, compressed once ----
    src = hsrle.synth(hsrle.SYNTH_RUNS if args.synth == "runs" else hsrle.SYNTH_VIDEO, codec.S, seed, size, device=dev)
    dst = torch.empty(hsrle.container_bound(size, args.block), dtype=torch.uint8, device=dev)
    ws = torch.empty(hsrle.workspace_size(size, args.block), dtype=torch.uint8, device=dev)
    hsrle.compress_async(args.codec, src, dst, args.block, workspace=ws)
    torch.cuda.synchronize()
    info = hsrle.container_info(dst)
    container = dst[: info.totalSize]
    out = torch.empty(size, dtype=torch.uint8, device=dev)
    status = torch.zeros(16, dtype=torch.int32, device=dev)

    def step():
        hsrle.decompress_async(container, info, out, status)

    for _ in range(args.warmup):
        step()
    barrier()

    # ---- timed region: exactly K steps, bracketed by barrier + synchronize; HIP events on the launch stream ----
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps

    per_rank_ms = [round(wall / args.steps * 1e3, 4)]
    if distributed:
        allw = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(allw, torch.tensor([wall], dtype=torch.float64, device=dev))
        per_rank_ms = [round(float(w.item()) / args.steps * 1e3, 4) for w in allw]
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # ---- correctness (untimed): the decode equals the input, and EVERY block stream is the reference encoder's -- through the device
    #      hash of every block and the per-256-block roll-ups minted from the compiled reference (tests/golden/big/); a workload that has
    #      no manifest (non-default --codec / --size-gib / --block) is checked on a 16 MiB sample against the oracle ----
    ok = int(status[0].item()) == 0 and torch.equal(out, src)
    from hsrle_testlib import big_case, rollups

    p0 = info.payload_start
    case = big_case(args.codec, 0 if args.synth == "runs" else 1, seed, size, args.block)
    if case is not None:
        name, entry, want = case
        got = rollups(hsrle.hash_blocks(container, info).cpu().numpy())
        streams_ok = info.blockCount == entry["blocks"] and info.payloadSize == entry["payload_size"] and bool((got == want).all())
        parity = {"blocks_compared": int(info.blockCount), "against": f"tests/golden/big/{name} (compiled reference)"}
    else:
        sample_blocks = min(info.blockCount, (16 << 20) // args.block)
        ora = Oracle()
        host_sample = src[: sample_blocks * args.block].cpu().numpy()
        table = container[64 : 64 + 8 * (sample_blocks + 1)].view(torch.int64).cpu().numpy()
        pay = container[p0 : p0 + int(table[sample_blocks])].cpu().numpy().tobytes()
        gpu_streams = [pay[int(table[i]) : int(table[i + 1])] for i in range(sample_blocks)]
        streams_ok = gpu_streams == ora.compress_blocks(codec, host_sample, args.block)
        parity = {"blocks_compared": int(sample_blocks), "against": "oracle (no manifest for this workload)"}
    ok = ok and streams_ok
    per_rank_exact, per_rank_build = [bool(ok)], [hsrle.build_id()]
    if distributed:  # bit_exact is a statement about every rank's shard; the line also says which rank said what, and which library each rank ran
        flags = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(world)]
        dist.all_gather(flags, torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev))
        per_rank_exact = [bool(f.item()) for f in flags]
        ok = all(per_rank_exact)
        mine = torch.tensor(list(hsrle.build_id().encode()[:32].ljust(32, b" ")), dtype=torch.uint8, device=dev)
        ids = [torch.zeros(32, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.all_gather(ids, mine)
        per_rank_build = [bytes(i.cpu().tolist()).decode().strip() for i in ids]

    # ---- encode throughput of the same buffer (untimed for `value`) ----
    for _ in range(2):
        hsrle.compress_async(args.codec, src, dst, args.block, workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    enc_steps = max(3, args.steps // 4)
    e0.record()
    for _ in range(enc_steps):
        hsrle.compress_async(args.codec, src, dst, args.block, workspace=ws)
    e1.record()
    torch.cuda.synchronize()
    enc_ms = e0.elapsed_time(e1) / enc_steps

    # ---- side measurements on rank 0 (untimed for `value`): BASELINE config 3 at its own size (split decode of the 4 KiB-block container),
    #      and ONE monolithic 1 GiB reference stream written and read by the GPU (what the rle.h drop-in functions do, without PCIe) ----
    extras = {}
    if rank == 0 and args.codec == "rle8_packed_multi" and args.synth == "runs" and not args.no_extras:
        try:
            extras = side_measurements(hsrle, torch, src, dev)
        except Exception as ex:  # noqa: BLE001 -- a side measurement must never cost the headline line
            extras = {"error": repr(ex)[:200]}

    # ---- the one exchange step of the sharded path: gather the per-rank compressed segments into one stream ----
    gather_ms = None
    gather_bytes = 0.0
    if distributed:
        from hsrle import dist as hd

        use_c = os.environ.get("HSRLE_DIST_C") == "1"                      # the library's own communicator (hsrle_gather_container_rccl) instead of torch's
        if use_c:
            hd.c_comm()                                                    # (communicator creation is setup, not gather time)
        # how many ranks the COMMUNICATOR spans (not WORLD_SIZE): ncclCommCount of the library's communicator, or the size of torch's RCCL process group
        rccl_ranks = hd.c_comm_ranks()[0] if use_c else dist.get_world_size()
        rccl_via = "hsrle_gather_container_rccl (ncclCommCount)" if use_c else f"torch.distributed backend {dist.get_backend()} (process group size)"
        barrier()
        g0 = time.perf_counter()
        full = hd.gather_container_c(container, size * world, root=0) if use_c else hd.gather_container(container, size * world, root=0)
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - g0) * 1e3], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                           # the slowest rank's time from the common start to its own completion
        gather_ms = float(t.item())
        # bytes that cross xGMI: every non-root rank's container (its segment travels whole; the root's own stays where it is)
        t = torch.tensor([float(info.totalSize) if rank != 0 else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        gather_bytes = float(t.item())
        del full
        if use_c:
            hd.destroy_c_comms()

    if rank == 0:
        traffic, traffic_detail, enc_traffic = measured_traffic(args.codec, size, args.block, hsrle.build_id())
        total_units = size * world
        alg_bytes = size + info.totalSize
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "rle8_extreme Packed decode throughput, uncompressed bytes (device resident block container)",
            "value": round(total_units / 2**30 / (wall / args.steps), 2),
            "unit": "GiB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": f"{args.codec} decode, {size / 2**30:g} GiB {synth_name} synthetic per GPU (seed {seed}), {args.block} B blocks, "
                                   f"ratio {info.totalSize / size:.4f}", "codec": args.codec, "block_size": args.block, "blocks_per_gpu": info.blockCount, "sharding": f"blocks x{world}"},
            "bit_exact": bool(ok),
            "parity": parity,
            "encode": {"value": round(size / 2**30 / (enc_ms * 1e-3), 2), "unit": "GiB/s", "ms": round(enc_ms, 4), "note": "same buffer: position-parallel encoder = sizes + records launch, size scan, emission launch (no staging slots, no compaction)",
                       "roofline": {"bound": "hbm", "achieved": round(alg_bytes / (enc_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(alg_bytes / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": enc_traffic, "algorithmic_bytes": int(alg_bytes),
                                    "note": "algorithmic bytes = input + container per encode; traffic = upper bound of the PMC passes over both launches (same profiles/ file as roofline.traffic_detail.source): the input is read twice (sizes + records, then emission), the records once each way, the payload written once"}},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": "committed PMC passes of this build and this workload (profiles/, tools/traffic.sh): not a measurement of this run", "traffic_detail": traffic_detail, "kernel": kernel_name(args.codec), "waves_per_cu": hsrle.kernel_waves_per_cu(args.codec, True), "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes": int(alg_bytes),
                         "note": "algorithmic bytes = container (compressed) + uncompressed output per launch; traffic = upper bound of the PMC passes in profiles/ (128 B per counted fabric read request + WRITE_SIZE), lower bound in traffic_detail"},
        }
        if gather_ms is not None:
            # the one exchange step of the sharded path ("aggregate GiB/s with and without the gather"): what the root receives
            # over xGMI -- every other rank's compressed segment -- against the time from the common start to the slowest rank's completion;
            # the ceiling for the root's ingress is 7 links x ~153 GB/s.  value_with_gather charges ONE gather to ONE decode step.
            line["gather_ms"] = round(gather_ms, 3)
            line["gather_bytes"] = int(gather_bytes)
            line["gather_GBps"] = round(gather_bytes / (gather_ms * 1e-3) / 1e9, 2) if gather_bytes else 0.0
            line["gather_link_ceiling_GBps"] = 7 * 153
            line["value_with_gather"] = round(total_units / 2**30 / (wall / args.steps + gather_ms * 1e-3), 2)
            line["per_rank_ms_per_step"] = per_rank_ms
            line["rccl_ranks"] = int(rccl_ranks)
            line["rccl_ranks_source"] = rccl_via
            line["per_rank_bit_exact"] = per_rank_exact
            line["per_rank_library_build_id"] = per_rank_build
        line["library_build_id"] = hsrle.build_id()
        if extras:
            line["extras"] = extras
        run_config5 = extras is not None and not args.no_extras and not args.no_config5 and args.codec == "rle8_packed_multi" and args.synth == "runs" and world == 1
        if not args.no_cpu:
            # rank 0's shard, on rank 0's host cores, whatever the world size (the other ranks are through their timed region)
            nb = min(info.blockCount, (1 << 30) // args.block)
            prefix_end = p0 + int(container[64 + 8 * nb : 64 + 8 * nb + 8].view(torch.int64).item()) + 64
            prefix = container[: min(prefix_end, container.numel())].cpu().numpy().tobytes()
            sample = src[: nb * args.block].cpu().numpy()
            line["cpu_baseline"] = cpu_baseline(prefix, nb, args.block, args.codec, sample.tobytes())
            import numpy as np

            line["encode"]["cpu_baseline"] = cpu_encode_baseline(sample, args.block, args.codec, prefix[p0:], np.frombuffer(prefix, dtype=np.uint64, count=nb + 1, offset=64))
        if run_config5:
            # BASELINE config 5 in the driver-run line: frees the headline buffers first (each row holds its own 8 GiB input, container, workspace, output)
            del src, dst, ws, out, container
            torch.cuda.empty_cache()
            try:
                t5 = time.time()
                rows = config5_rows(hsrle, torch, dev, with_cpu=not args.no_cpu)
                line.setdefault("extras", {})["config5"] = {"workload": "8 GiB run-distributed(W, seed 5), 4 KiB blocks, every codec with a reference-minted manifest", "rows": rows,
                                                            "all_bit_exact": bool(rows) and all(r["bit_exact"] for r in rows), "seconds": round(time.time() - t5, 1),
                                                            "cpu_sample": "reference CPU codec, one core, first 64 MiB of the same buffer, block by block" if not args.no_cpu else None}
            except Exception as ex:  # noqa: BLE001
                line.setdefault("extras", {})["config5"] = {"error": repr(ex)[:300]}
            try:
                tv = time.time()
                torch.cuda.empty_cache()
                rows = video_rows(hsrle, torch, dev, with_cpu=not args.no_cpu)
                line["extras"]["video_shaped"] = {"workload": "8 GiB video-shaped(W, seed 5), 4 KiB blocks: the data shape of BASELINE config 3 at the size of config 2", "rows": rows,
                                                  "all_exact": bool(rows) and all(r["exact"] for r in rows), "seconds": round(time.time() - tv, 1)}
            except Exception as ex:  # noqa: BLE001
                line["extras"]["video_shaped"] = {"error": repr(ex)[:300]}
            try:
                tl = time.time()
                torch.cuda.empty_cache()
                rows = video_rows(hsrle, torch, dev, with_cpu=not args.no_cpu, kind=hsrle.SYNTH_RUNS, block=65536,
                                  keys=("rle8_packed_multi", "rle8_multi", "rle16_sym_packed", "rle32_byte", "rle64_3symlut_byte", "rle24_byte_short"))
                line["extras"]["large_blocks"] = {"workload": "8 GiB run-distributed(W, seed 5), 64 KiB blocks: every block walked in 4 KiB windows by the windowed position-parallel encoders "
                                                              "(round 6; encode path = hsrle_encode_path)", "rows": rows, "all_exact": bool(rows) and all(r["exact"] for r in rows),
                                                  "encode_path": {r["codec"]: int(hsrle.lib().hsrle_encode_path(hsrle.codec_id(r["codec"]), 8 << 30, 65536)) for r in rows},
                                                  "seconds": round(time.time() - tl, 1)}
            except Exception as ex:  # noqa: BLE001
                line["extras"]["large_blocks"] = {"error": repr(ex)[:300]}
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + "\n").encode())

    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
