"""hsrle_decompress_mono_dev_async: ONE monolithic reference stream decoded without the host in the loop (walk, proof, gated entry records,
decode enqueued in one go; include/hsrle.h).  Bar: status DONE -> the output equals what the oracle's (= the reference's) encoder was given;
a stream whose entry guesses do not hold says NEEDS_REPAIR in bounded time and the synchronous function then decodes it; malformed
streams say MALFORMED and never write behind the output; the call can be captured in a HIP graph and replayed on new stream bytes.
Reference decoders: src/rle8_extreme_cpu.h:702-764, src/rleX_extreme_cpu_decode.h:27-164, src/rleX_Xsl.h:1848-1881."""
import random
import time

import numpy as np
import pytest

from hsrle_testlib import CODECS, CODEC_BY_KEY, SYNTH_RUNS, SYNTH_VIDEO, mixed_runs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hs():
    import torch

    assert torch.cuda.is_available(), "these tests need a GPU"
    import hsrle

    hsrle.lib()
    yield hsrle
    hsrle.mono_tuning(0, 0, 0)


def _dev_stream(stream, capacity=None):
    import torch

    t = torch.zeros((capacity or len(stream)) + 64, dtype=torch.uint8, device="cuda")
    t[: len(stream)] = torch.frombuffer(bytearray(stream), dtype=torch.uint8).cuda()
    return t


def _async_decode(hs, key, stream, usize, slack=0):
    import torch

    t = _dev_stream(stream)
    out = torch.full((usize + slack,), 0xA5, dtype=torch.uint8, device="cuda")
    ws = torch.empty(max(hs.mono_decompress_workspace_size(key, usize, len(stream)), 256), dtype=torch.uint8, device="cuda")
    status = torch.full((1,), 77, dtype=torch.int32, device="cuda")
    n = hs.mono_decompress_dev_async(key, t, stream[:16], out[:usize], ws, status, stream_size=len(stream))
    torch.cuda.synchronize()
    return n, int(status.item()), out, t


@pytest.mark.parametrize("key", [c.key for c in CODECS])
def test_async_mono_decode_every_codec(hs, oracle, key):
    """Every codec through the enqueue-only entry point.  The contract has two good ends and both are checked: DONE -> the output is the
    input; NEEDS_REPAIR (an entry guess of the walk did not hold: routine for the formats whose junk walks do not die, whose look-back the
    synchronous function widens from the host) -> nothing behind the output was written and the synchronous function decodes the stream."""
    import torch

    codec = CODEC_BY_KEY[key]
    data = oracle.synth(SYNTH_RUNS, codec.S, 5, (3 << 20) + 777)
    stream = oracle.compress(codec, data.tobytes())
    n, status, out, t = _async_decode(hs, key, stream, data.size, slack=4096)
    assert n == data.size
    assert status in (hs.MONO_DONE, hs.MONO_NEEDS_REPAIR), f"{key}: status {status}"
    assert bool((out[data.size:] == 0xA5).all())
    if status == hs.MONO_DONE:
        assert torch.equal(out[: data.size].cpu(), torch.from_numpy(data)), f"{key}: async decode differs"
    else:
        got, stats = hs.mono_decompress_dev(key, t, return_stats=True)
        assert stats[1] > 0 or stats[2] > 0, f"{key}: NEEDS_REPAIR, but the synchronous decode repaired nothing"   # (rounds, or a second walk of everything with a longer look-back)
        assert torch.equal(got.cpu(), torch.from_numpy(data))


@pytest.mark.parametrize("key", ["rle8_packed_multi", "rle32_byte_packed", "rle64_byte_packed"])
def test_async_mono_decode_is_done_at_once_where_walks_die(hs, oracle, key):
    """The formats with the 7-bit-or-4-byte range field (junk walks die within hops: csrc/hsrle_capi.hip plan_mono) with the regions and the look-back the
    library uses for big streams (4 KiB / 2 KiB: no wrong guess in 71 096 regions of the 1 GiB stream): the first try is the only one."""
    import torch

    codec = CODEC_BY_KEY[key]
    data = oracle.synth(SYNTH_RUNS, codec.S, 9, 32 << 20)
    stream = oracle.compress(codec, data.tobytes())
    hs.mono_tuning(0, 4096, 2048)
    try:
        n, status, out, _ = _async_decode(hs, key, stream, data.size)
    finally:
        hs.mono_tuning(0, 0, 0)
    assert status == hs.MONO_DONE and torch.equal(out.cpu(), torch.from_numpy(data))


@pytest.mark.parametrize("key", ["rle8_packed_multi", "rle8_multi", "rle24_3symlut_sym", "rle16_sym_short", "rle64_byte"])
def test_async_mono_decode_reports_repair_in_bounded_time(hs, oracle, key):
    """Tiny regions and look-backs make guesses fail (tests/test_gpu_mono.py forces repairs the same way): the gated records pass must leave
    at once -- round 4 measured 105 ms for a records walk from junk entries -- and the synchronous function must finish the job."""
    import torch

    codec = CODEC_BY_KEY[key]
    rng = random.Random(99)
    parts = []
    while sum(map(len, parts)) < (2 << 20):
        parts.append(bytes(rng.randrange(256) for _ in range(rng.choice([1, 7, 130, 300, 700]))))
        parts.append(mixed_runs(rng, rng.choice([40, 200, 1000])))
    data = np.frombuffer(b"".join(parts)[: 2 << 20], dtype=np.uint8)
    stream = oracle.compress(codec, data.tobytes())
    hs.mono_tuning(128, 64, 16)
    try:
        _async_decode(hs, key, stream, data.size)                        # warm-up: module load, first launches
        t0 = time.perf_counter()
        n, status, out, t = _async_decode(hs, key, stream, data.size, slack=4096)
        dt = time.perf_counter() - t0
        assert status in (hs.MONO_DONE, hs.MONO_NEEDS_REPAIR)
        assert dt < 0.5, f"{key}: {dt * 1e3:.1f} ms for a first try that ends in status {status}"
        assert bool((out[data.size:] == 0xA5).all())
        if status == hs.MONO_DONE:
            assert torch.equal(out[: data.size].cpu(), torch.from_numpy(data))
        got, stats = hs.mono_decompress_dev(key, t, return_stats=True)
        assert torch.equal(got.cpu(), torch.from_numpy(data)), f"{key}: synchronous decode after status {status} differs"
        assert (stats[1] > 0 or stats[2] > 0) == (status == hs.MONO_NEEDS_REPAIR), f"{key}: status {status} but the synchronous decode took {stats[1]} repair rounds, walked {stats[2]} regions again"
    finally:
        hs.mono_tuning(0, 0, 0)


@pytest.mark.parametrize("key", ["rle8_packed_multi", "rle32_sym", "rle48_7symlut_byte"])
def test_async_mono_decode_malformed(hs, oracle, key):
    codec = CODEC_BY_KEY[key]
    data = oracle.synth(SYNTH_VIDEO, codec.S, 3, 1 << 20)
    stream = oracle.compress(codec, data.tobytes())
    lie = bytearray(stream)
    lie[0:4] = (data.size - 1).to_bytes(4, "little")                      # header claims one byte less than the packets produce
    n, status, out, _ = _async_decode(hs, key, bytes(lie), data.size - 1, slack=4096)
    assert status == hs.MONO_MALFORMED and bool((out[data.size - 1:] == 0xA5).all())
    cut = bytearray(stream[: len(stream) * 2 // 3])
    cut[4:8] = len(cut).to_bytes(4, "little")                              # truncated stream with a consistent header
    n, status, out, _ = _async_decode(hs, key, bytes(cut), data.size, slack=4096)
    assert status == hs.MONO_MALFORMED and bool((out[data.size:] == 0xA5).all())
    with pytest.raises(hs.HsrleError):                                     # sizes the header cannot have: refused on the host, nothing enqueued
        bad = bytearray(stream)
        bad[4:8] = (len(stream) + 5).to_bytes(4, "little")
        _async_decode(hs, key, bytes(bad), data.size)


def test_async_mono_decode_in_a_hip_graph(hs, oracle):
    """Capture once, replay on other stream bytes of the same geometry (same header sizes): no host work per replay."""
    import torch

    key = "rle8_packed_multi"
    codec = CODEC_BY_KEY[key]
    # two inputs whose streams have the same size: the same runs, other symbols (the packets keep their lengths)
    a = oracle.synth(SYNTH_RUNS, 1, 21, 8 << 20)
    b = (a ^ np.uint8(0x5A)).astype(np.uint8)
    sa, sb = oracle.compress(codec, a.tobytes()), oracle.compress(codec, b.tobytes())
    if len(sa) != len(sb):
        pytest.skip("streams of different sizes: not the same launch geometry")
    t = _dev_stream(sa)
    out = torch.empty(a.size, dtype=torch.uint8, device="cuda")
    ws = torch.empty(max(hs.mono_decompress_workspace_size(key, a.size, len(sa)), 256), dtype=torch.uint8, device="cuda")
    status = torch.full((1,), 77, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        hs.mono_decompress_dev_async(key, t, sa[:16], out, ws, status)      # warm-up outside the capture (module load)
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        hs.mono_decompress_dev_async(key, t, sa[:16], out, ws, status)
    for stream, data in ((sb, b), (sa, a), (sb, b)):
        t[: len(stream)] = torch.frombuffer(bytearray(stream), dtype=torch.uint8).cuda()
        out.zero_()
        status.fill_(77)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert int(status.item()) == hs.MONO_DONE
        assert torch.equal(out.cpu(), torch.from_numpy(data))


# ---- the encode side (round 6): hsrle_compress_mono_dev_async, rle8_multi / rle8_packed_multi (csrc/hsrle_encode8pw.hip.h) ----
@pytest.mark.parametrize("key", ["rle8_multi", "rle8_packed_multi"])
@pytest.mark.parametrize("kind", [SYNTH_RUNS, SYNTH_VIDEO])
def test_async_mono_encode_equals_the_reference_stream(hs, oracle, key, kind):
    """The enqueue-only encode: the stream (its size from its own header and from the device word) == the oracle's (src/rle8_extreme_cpu.h:86-344)."""
    import torch

    data = oracle.synth(kind, 1, 21, (5 << 20) + 333)
    src = torch.from_numpy(data).cuda()
    dst = torch.full((hs.compress_bounds(data.size) + 64,), 0xEE, dtype=torch.uint8, device="cuda")
    ws = torch.empty(hs.lib().hsrle_compress_mono_workspace_size(hs.codec_id(key), data.size), dtype=torch.uint8, device="cuda")
    size = torch.full((1,), 0x7FFFFFFF, dtype=torch.int32, device="cuda")
    hs.mono_compress_dev_async(key, src, dst, ws, size)
    torch.cuda.synchronize()
    want = oracle.compress(CODEC_BY_KEY[key], data.tobytes())
    assert int(size.item()) == len(want)
    got = dst[: len(want)].cpu().numpy().tobytes()
    assert int.from_bytes(got[4:8], "little") == len(want)
    assert got == want


def test_async_mono_encode_refuses_the_other_codecs(hs):
    import torch

    src = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    dst = torch.empty(hs.compress_bounds(src.numel()) + 64, dtype=torch.uint8, device="cuda")
    for key in ("rle8_3symlut", "rle16_sym_packed", "rle8_single"):
        ws = torch.empty(max(hs.lib().hsrle_compress_mono_workspace_size(hs.codec_id(key), src.numel()), 256), dtype=torch.uint8, device="cuda")
        with pytest.raises(hs.HsrleError):
            hs.mono_compress_dev_async(key, src, dst, ws)


def test_async_mono_encode_in_a_hip_graph(hs, oracle):
    """Captured once, replayed on new input bytes of the same size: every replay's stream == the oracle's."""
    import torch

    key = "rle8_packed_multi"
    n = (6 << 20) + 17
    inputs = [oracle.synth(SYNTH_RUNS, 1, 31, n), oracle.synth(SYNTH_VIDEO, 1, 32, n), oracle.synth(SYNTH_RUNS, 1, 33, n)]
    src = torch.from_numpy(inputs[0]).cuda()
    dst = torch.empty(hs.compress_bounds(n) + 64, dtype=torch.uint8, device="cuda")
    ws = torch.empty(hs.lib().hsrle_compress_mono_workspace_size(hs.codec_id(key), n), dtype=torch.uint8, device="cuda")
    size = torch.zeros(1, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        hs.mono_compress_dev_async(key, src, dst, ws, size)                  # warm-up outside the capture (module load)
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        hs.mono_compress_dev_async(key, src, dst, ws, size)
    for data in (inputs[1], inputs[2], inputs[0]):
        src.copy_(torch.from_numpy(data))
        dst.fill_(0xEE)
        ws.fill_(0xC3)
        size.zero_()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        want = oracle.compress(CODEC_BY_KEY[key], data.tobytes())
        assert int(size.item()) == len(want)
        assert dst[: len(want)].cpu().numpy().tobytes() == want
