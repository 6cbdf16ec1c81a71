"""GPU parity tests: the HIP path, called through the C ABI (libhsrle_hip.so), against the oracle on the same inputs.

Bit-exact bar (integer / byte work): every block stream must equal the oracle's (= the reference's, see
test_oracle_vs_ref.py / test_oracle_golden.py) stream for that block, and every decode must reproduce the input exactly.
"""
import os
import re
import ctypes
import random
import struct

import numpy as np

import pytest

from hsrle_testlib import CODECS, CODEC_BY_KEY, FUZZ_LENGTHS, Oracle, fuzz_sections, mixed_runs, single_symbol_mix
from hsrle_testlib import SYNTH_VIDEO as SYNTH_VIDEO_KIND

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hs():
    import torch

    assert torch.cuda.is_available(), "these tests need a GPU"
    import hsrle

    hsrle.lib()  # fails loudly if the HIP library is missing
    return hsrle


def _inputs(seed, count):
    rng = random.Random(seed)
    out = []
    for it in range(count):
        k = it % 4
        if k == 0:
            d = fuzz_sections(rng)
        elif k == 1:
            d = mixed_runs(rng, rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 31, 32, 33, 34, 40, 47, 48, 49, 63, 64, 65, 66, 70, 90, 100, 128, 200, 333, 1000, 3000]))
        elif k == 2:
            d = single_symbol_mix(rng, rng.choice([1, 2, 5, 15, 16, 17, 18, 20, 31, 32, 33, 34, 40, 47, 48, 49, 63, 64, 65, 66, 70, 90, 100, 128, 200, 333, 1000, 3000, 9000]))
        else:
            d = fuzz_sections(rng, lengths=FUZZ_LENGTHS, max_sections=4)
        if d:
            out.append(d)
    return out


def _to_dev(b):
    import torch

    return torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()


@pytest.mark.parametrize("codec", CODECS, ids=lambda c: c.key)
def test_blocks_bit_exact_and_roundtrip(hs, oracle, codec):
    """Concatenate fuzz inputs, cut into small blocks: every block stream == oracle stream; decode == input."""
    inputs = _inputs(1234 + CODECS.index(codec), 40)
    data = b"".join(inputs)
    for block_size in (128, 384, 4096):
        src = _to_dev(data)
        container, info = hs.compress(codec.key, src, block_size=block_size)
        cinfo, streams = hs.split_container(container.cpu().numpy().tobytes())
        assert cinfo.blockCount == (len(data) + block_size - 1) // block_size == len(streams)
        for i, s in enumerate(streams):
            expect = oracle.compress(codec, data[i * block_size : (i + 1) * block_size])
            assert s == expect, f"{codec.key} block {i} (size {block_size}) differs from the oracle"
        out = hs.decompress(container)
        assert out.cpu().numpy().tobytes() == data


@pytest.mark.parametrize("codec", CODECS, ids=lambda c: c.key)
def test_small_containers_of_1_to_4_kib_blocks_bit_exact(hs, oracle, codec):
    """Containers of fewer than 131 072 blocks of 1 .. 4 KiB: rle8_multi / rle8_packed_multi and the plain / Packed / LUT codecs of 4, 6, 8 byte
    symbols went through the run list encoders until round 6 (now: the position-parallel encoders; csrc/hsrle_encode8r.hip.h still serves rle8_3symlut_short / rle8_7symlut_short), the others through the split encode or the
    ring encoders.  Whatever the path: every block stream == the oracle's, ragged last block included; decode == input."""
    rng = random.Random(4242 + CODECS.index(codec))
    parts = _inputs(777 + CODECS.index(codec), 60)
    # (stretches that meet within S bytes, symbols that come back, a long tail of literals, a run up to the very end)
    for S in (1, 4, 6, 8):
        sym = bytes(rng.randrange(256) for _ in range(S))
        parts.append((sym * 40)[: 37 * S + 3] + bytes([rng.randrange(256)]) + (sym * 40)[: 9 * S + 1] + bytes(rng.randrange(256) for _ in range(300)) + sym * 30)
    data = b"".join(parts)
    for block_size, cut in ((1024, 0), (1536, 1), (3072, 700), (4096, 4095)):
        d = data[: len(data) - cut] if cut else data
        src = _to_dev(d)
        container, info = hs.compress(codec.key, src, block_size=block_size)
        cinfo, streams = hs.split_container(container.cpu().numpy().tobytes())
        expect = oracle.compress_blocks(codec, np.frombuffer(d, dtype=np.uint8), block_size)
        assert len(streams) == len(expect)
        for i, (a, b) in enumerate(zip(streams, expect)):
            assert a == b, f"{codec.key} block {i} (size {block_size}) differs from the oracle"
        assert hs.decompress(container).cpu().numpy().tobytes() == d


@pytest.mark.parametrize("key", ["rle8_multi", "rle8_packed_multi", "rle8_3symlut", "rle8_7symlut", "rle16_sym_packed", "rle16_7symlut_byte", "rle24_byte", "rle24_3symlut_sym",
                                 "rle32_sym", "rle32_7symlut_byte", "rle48_byte_packed", "rle48_3symlut_sym", "rle64_sym_packed", "rle64_3symlut_byte", "rle64_7symlut_sym",
                                 "rle8_multi_short", "rle8_7symlut_short", "rle16_3symlut_byte_short", "rle32_sym_short", "rle64_1symlut_sym_short", "rle64_7symlut_byte_short"])
def test_run_list_batches_that_overflow_the_candidate_list(hs, oracle, key):
    """The 88 MB video-shaped frame (21 600 blocks of 4 KiB, BASELINE config 3) gives every wave of the run list encoders more blocks than its
    candidate list holds: the batch is flushed in the middle of the wave's blocks (and the input tile, which shares LDS with the batch's packet
    table, filled again).  All 21 600 streams == the oracle's."""
    import torch

    codec = CODEC_BY_KEY[key]
    src = hs.synth(SYNTH_VIDEO_KIND, codec.S, 2, 88473600, device="cuda")
    container, info = hs.compress(key, src, block_size=4096)
    cinfo, streams = hs.split_container(container.cpu().numpy().tobytes())
    expect = oracle.compress_blocks(codec, src.cpu().numpy(), 4096)
    bad = [i for i, (a, b) in enumerate(zip(streams, expect)) if a != b]
    assert len(streams) == len(expect) == 21600 and not bad, f"{key}: {len(bad)} block streams differ from the oracle, first {bad[:5]}"
    assert torch.equal(hs.decompress(container), src)


@pytest.mark.parametrize("codec", CODECS, ids=lambda c: c.key)
def test_dropin_monolithic(hs, oracle, codec):
    """rle.h-named entry points (host pointers, one stream): stream == oracle; decode of oracle stream == input; error returns."""
    for d in _inputs(77 + CODECS.index(codec), 24):
        cap = hs.compress_bounds(len(d))
        size, stream = hs.call_dropin(codec.cname, d, cap)
        expect = oracle.compress(codec, d)
        assert size == len(expect) and stream == expect
        size, dec = hs.call_dropin(codec.dname, expect, len(d))  # outSize == inputSize exactly, like the reference fuzzer
        assert size == len(d) and dec == d
    # error behaviour (reference: rle8_extreme_cpu.h:88-89, :704-712)
    d = b"abcabcabc" * 10
    assert hs.call_dropin(codec.cname, d, hs.compress_bounds(len(d)) - 1)[0] == 0  # outSize < bounds
    assert hs.call_dropin(codec.cname, b"", 1000)[0] == 0  # inSize == 0
    s = oracle.compress(codec, d)
    assert hs.call_dropin(codec.dname, s, len(d) - 1)[0] == 0  # outSize < uncompressed
    assert hs.call_dropin(codec.dname, s[:-1], len(d))[0] == 0  # inSize < compressedLength


def test_decode_reference_tail_flavours(hs, oracle):
    """8 bit Packed: the decoder is ISA-invariant; golden streams of both encoder tail flavours decode."""
    import json, os, base64

    path = os.path.join(os.path.dirname(__file__), "golden", "rle8_packed_tails.json")
    if not os.path.exists(path):
        pytest.skip("golden file not minted")
    for case in json.load(open(path)):
        data = base64.b64decode(case["input"])
        for k in ("sse2", "avx2"):
            stream = base64.b64decode(case[k])
            size, dec = hs.call_dropin("rle8_packed_decompress", stream, len(data))
            assert size == len(data) and dec == data


def test_malformed_block_reports_error(hs):
    import torch

    data = bytes(range(256)) * 64
    container, info = hs.compress("rle8_packed_multi", _to_dev(data), block_size=1024)
    bad = container.clone()
    # corrupt the first block's stream header (uncompressedLength)
    p0 = info.payload_start
    bad[p0] = 0x55
    with pytest.raises(hs.HsrleError):
        hs.decompress(bad)
    # corrupt the offset table: entries beyond the payload, and a decreasing pair -- an error, not a wild read
    for pos, value in ((3, 1 << 40), (5, 0)):
        bad3 = container.clone()
        bad3[64 + 8 * pos : 64 + 8 * pos + 8] = torch.frombuffer(bytearray(struct.pack("<Q", value)), dtype=torch.uint8).cuda()
        with pytest.raises(hs.HsrleError):
            hs.decompress(bad3)
    # corrupt the container magic
    bad2 = container.clone()
    bad2[0] = 0
    with pytest.raises(hs.HsrleError):
        hs.decompress(bad2)


GARBAGE_KEYS = ["rle8_multi", "rle8_packed_multi", "rle8_3symlut", "rle8_7symlut", "rle16_sym", "rle24_byte_packed", "rle48_3symlut_sym", "rle64_7symlut_byte",
                "rle128_sym_packed", "rle8_multi_short", "rle8_1symlut_short", "rle8_7symlut_short", "rle8_single_short", "rle16_sym_short",
                "rle32_1symlut_byte_short", "rle48_3symlut_byte_short", "rle64_7symlut_sym_short"]


@pytest.mark.parametrize("key", GARBAGE_KEYS)
def test_garbage_streams_end_as_errors_not_hangs(hs, key):
    """Streams are data: block streams whose packets are random bytes (stream headers and offset table intact) must end -- as an
    error bit or as some output -- without a hang and without a byte written outside [0, uncompressedSize)."""
    import torch

    rng = random.Random(hash(key) & 0xFFFF)
    data = mixed_runs(rng, 300000)
    block = 1024
    container, info = hs.compress(key, _to_dev(data), block_size=block)
    host = bytearray(container.cpu().numpy().tobytes())
    p0 = info.payload_start
    table = struct.unpack_from(f"<{info.blockCount + 1}Q", host, 64)
    for i in range(info.blockCount):
        a, b = p0 + table[i], p0 + table[i + 1]
        mode = rng.randrange(3)
        for j in range(a + 10, b):                                      # keep the 8/9/10-byte stream header
            if mode == 0 or (mode == 1 and rng.random() < 0.1) or (mode == 2 and j > (a + b) // 2):
                host[j] = rng.randrange(256)
    bad = torch.frombuffer(host, dtype=torch.uint8).cuda()
    guard = 4096
    out = torch.full((len(data) + guard,), 0xA5, dtype=torch.uint8, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    hs.decompress_async(bad, info, out[: len(data)], status)
    torch.cuda.synchronize()
    assert bool((out[len(data):] == 0xA5).all()), "wrote beyond the output"
    assert int(status.item()) != 0                                       # some block must have noticed (most of 293 do)


def test_rle8m_decode_matches_the_oracle(hs, oracle):
    """rle8m (the reference's own GPU decode path): streams from the oracle's rle8m_compress decode on the GPU
    through the drop-in names of rle.h (rle8m_opencl_decompress, rle8m_decompress) and through the device API."""
    import torch

    rng = random.Random(77)
    lib = hs.lib()
    lib.rle8m_opencl_init.restype = ctypes.c_bool
    assert lib.rle8m_opencl_init(ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0))
    cases = [(mixed_runs(rng, 200000, alphabet=3), 64), (mixed_runs(rng, 70001, alphabet=256), 7), (single_symbol_mix(rng, 9000), 1),
             (bytes([5]) * 100000 + mixed_runs(rng, 3000), 16), (mixed_runs(rng, 333), 3), (bytes(range(256)) * 40 + b"\x00" * 5000, 33),
             (mixed_runs(rng, 1 << 20, alphabet=4), 4096),
             (mixed_runs(rng, 1 << 20, alphabet=5), 200000), (single_symbol_mix(rng, 600000), 150000)]   # >= 131072 sections: one lane per section; below: one wave
    n = 0
    for data, sections in cases:
        st = oracle.rle8m_compress(sections, data)
        if st is None:
            continue
        n += 1
        for name in ("rle8m_opencl_decompress", "rle8m_decompress"):
            size, got = hs.call_dropin(name, st, len(data))
            assert size == len(data) and got == data, f"{name}: {len(data)} bytes, {sections} sections"
        dev = _to_dev(st)
        info = hs.rle8m_info(dev)
        assert (info.compressedSize, info.uncompressedSize, info.sections) == (len(st), len(data), sections)
        out = torch.full((len(data) + 256,), 0xA5, dtype=torch.uint8, device="cuda")
        status = torch.ones(1, dtype=torch.int32, device="cuda")
        hs.rle8m_decompress_async(dev, info, out[: len(data)], status)
        torch.cuda.synchronize()
        assert int(status.item()) == 0 and out[: len(data)].cpu().numpy().tobytes() == data
        assert bool((out[len(data):] == 0xA5).all())
        # a damaged stream ends as an error, not as a wild write
        bad = bytearray(st)
        for j in range(len(st) // 2, len(st), 3):
            bad[j] = rng.randrange(256)
        size, _ = hs.call_dropin("rle8m_decompress", bytes(bad), len(data))
        assert size in (0, len(data))
    assert n >= 5
    lib.rle8m_opencl_destroy()


def test_single_mode_block_in_a_multi_container_is_a_format_error(hs):
    """Codec ids 0 / 1 name the multi-symbol encoders; their decode kernels are compiled without the Single mode (include/hsrle.h,
    k_decode_blocks SGL).  A container that claims id 0 but carries mode-1 streams is reported, not mis-decoded; under its own
    id (4) the same container decodes, and the drop-in rle8_decompress takes both modes like the reference's."""
    import torch

    rng = random.Random(3)
    data = single_symbol_mix(rng, 60000)
    container, info = hs.compress("rle8_single", _to_dev(data), block_size=1024)
    assert hs.decompress(container).cpu().numpy().tobytes() == data
    fake = container.clone()
    assert int.from_bytes(fake[12:16].cpu().numpy().tobytes(), "little") == 4   # codec field of the container header
    fake[12] = 0
    with pytest.raises(hs.HsrleError):
        hs.decompress(fake)


def test_rle8m_encode_matches_the_oracle(hs, oracle):
    """rle8m_compress on the GPU (the reference's is a CPU function): the stream equals the oracle's byte for byte -- and the call gives
    up (returns 0) on exactly the inputs on which the reference does (a section that outgrows the room left in the output)."""
    import torch

    rng = random.Random(78)
    cases = [(mixed_runs(rng, 200000, alphabet=3), 64), (mixed_runs(rng, 70001, alphabet=256), 7), (single_symbol_mix(rng, 9000), 1),
             (bytes([5]) * 100000 + mixed_runs(rng, 3000), 16), (mixed_runs(rng, 333), 3), (bytes(range(256)) * 40 + b"\x00" * 5000, 33),
             (mixed_runs(rng, 1 << 20, alphabet=4), 4096), (bytes(rng.randrange(256) for _ in range(5000)), 2), (b"\x01\x02" * 3000, 5),
             (b"\x00" * 70000, 9), (mixed_runs(rng, 100, alphabet=2), 100),
             (b"ab" * 2000 + b"a" * 4000, 2)]       # 'a' carries repeat codes (one long run) and the first section is full of single a's: it outgrows the bound
    gave_up = 0
    for data, sections in cases:
        want = oracle.rle8m_compress(sections, data)
        got = hs.rle8m_compress_dropin(sections, data)
        assert got == want, f"rle8m_compress x{sections} on {len(data)} bytes: {'GPU gave up' if got is None else 'stream differs'}"
        if want is None:
            gave_up += 1
            continue
        # device resident: encode + decode without leaving the GPU
        src = _to_dev(data)
        dst = torch.empty(hs.rle8m_bounds(sections, len(data)), dtype=torch.uint8, device="cuda")
        ws = torch.full((hs.rle8m_workspace_size(len(data), sections),), 0xC3, dtype=torch.uint8, device="cuda")   # (garbage: nothing may rely on a zeroed workspace)
        status = torch.ones(1, dtype=torch.int32, device="cuda")
        hs.rle8m_compress_async(src, sections, dst, ws, status)
        torch.cuda.synchronize()
        assert int(status.item()) == 0
        info = hs.rle8m_info(dst)
        assert dst[: info.compressedSize].cpu().numpy().tobytes() == want
        out = torch.empty(len(data), dtype=torch.uint8, device="cuda")
        hs.rle8m_decompress_async(dst, info, out, status)
        torch.cuda.synchronize()
        assert int(status.item()) == 0 and torch.equal(out, src)
    assert gave_up >= 1


def test_rle8m_fuzz_small_and_ragged_inputs(hs, oracle):
    """rle8m on the fuzz grammar at small sizes: inputs shorter than one 16-byte window, sections of a few bytes, a last section
    that takes the remainder, input lengths that are no multiple of 16 (the tails of the kernels' input rings and accumulators)."""
    rng = random.Random(4711)
    n = gave_up = 0
    for it in range(260):
        length = rng.choice([1, 2, 3, 7, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 100, 127, 129, 255, 257, 500, 1000, 1023, 2049, 3000])
        k = it % 3
        data = mixed_runs(rng, length, alphabet=rng.choice([2, 3, 8, 256])) if k == 0 else single_symbol_mix(rng, length) if k == 1 else fuzz_sections(rng, max_sections=3)[:length]
        if not data:
            continue
        sections = rng.choice([1, 1, 2, 3, 5, 8, 13, 64, len(data)])
        if len(data) // sections == 0:
            sections = 1
        want = oracle.rle8m_compress(sections, data)      # None also where the reference writes behind its output (a section stream
        got = hs.rle8m_compress_dropin(sections, data)    # of up to twice the section size): the GPU reports a failure there, in bounds
        assert got == want, f"rle8m_compress x{sections} on {len(data)} bytes ({data[:40].hex()}...)"
        if want is None:
            gave_up += 1
            continue
        size, back = hs.call_dropin("rle8m_decompress", want, len(data))
        assert size == len(data) and back == data, f"rle8m_decompress x{sections} on {len(data)} bytes"
        n += 1
    assert n >= 150


LE_NAMES = ("rle8_low_entropy_compress", "rle8_low_entropy_short_compress", "rle8_low_entropy_compress_only_max_frequency", "rle8_low_entropy_short_compress_only_max_frequency")


def test_low_entropy_unsectioned_forms_match_the_oracle(hs, oracle):
    """rle8_low_entropy[_short]_compress[_only_max_frequency] / _decompress behind the reference's names on the GPU -- the
    four encoders write the oracle's (= the reference's) streams byte for byte, give up (0) exactly where the reference's stream outgrows its
    bound, and both decoders read the oracle's streams; sizes around the encoders' 256-byte tail rule and runs around 32 / 255 included."""
    rng = random.Random(812)
    lib = hs.lib()
    lib.rle8_low_entropy_compress_bounds.restype = ctypes.c_uint32
    lib.rle8_low_entropy_decompressed_size.restype = ctypes.c_uint32
    inputs = [mixed_runs(rng, 200000, alphabet=3), single_symbol_mix(rng, 9000), bytes([5]) * 100000 + mixed_runs(rng, 3000), mixed_runs(rng, 333),
              bytes(range(256)) * 40 + b"\x00" * 5000, b"\x00" * 70000, b"\x07", b"ab" * 2000 + b"a" * 4000, bytes(rng.randrange(256) for _ in range(5000)),
              mixed_runs(rng, 1 << 20, alphabet=4),
              # many pieces (the one stream is cut every 16 KiB, hsrle_rle8m.hip.h): one run across 64 pieces, runs that straddle the cuts, a stream
              # whose bytes are ALL flagged values (symbol, code, symbol ... with codes that are flagged symbols themselves: the decoder cannot
              # find a piece's parity within its look-back and falls back to one wave), flagged runs of 32 / 255 + 1 bytes at the cuts
              b"\x00" * (1 << 20), mixed_runs(rng, 300000, alphabet=2), b"\x00\x00\x01\x01" * 100000,
              (b"\x07" * 16383 + b"\x09" + b"\x07" * 33 + b"ab" * 100) * 9, b"q" * 16384 + b"q" * 256 + bytes(rng.randrange(256) for _ in range(40000)) + b"z" * 70000,
              # round 6: pieces that begin INSIDE a run (cuts at run start + k * 255 / 32 for a flagged symbol, anywhere for one that is not flagged): runs of
              # many lengths across many pieces, runs that end inside the input's last 256 bytes, a long run of a symbol that stays unflagged
              b"\x05" * (3 * 16384 + 77), b"\x05" * (5 * 16384 + 255), b"\x05" * (2 * 16384 + 100) + b"\x06" * (4 * 16384 + 31) + b"\x05" * 200,
              b"ab" * 30000 + b"c" * 100000 + b"ab" * 50, bytes(rng.randrange(256) for _ in range(3000)) + b"\x00" * 200001 + b"\x01" * 99999 + b"\x00" * 16384 + b"x",
              bytes(rng.randrange(2, 256) for _ in range(120000)) + b"\x01" * 40000 + bytes(rng.randrange(2, 256) for _ in range(300000)) + b"\x01" * 33000]
    for it in range(120):
        length = rng.choice([1, 2, 3, 15, 16, 17, 31, 32, 33, 34, 63, 64, 65, 100, 254, 255, 256, 257, 258, 300, 511, 512, 513, 700, 1000, 3000])
        alphabet = [rng.randrange(256) for _ in range(rng.choice([1, 2, 3, 5, 17]))]
        d = bytearray()
        while len(d) < length:
            d += bytes([rng.choice(alphabet)]) * rng.choice([1, 1, 2, 3, 7, 30, 31, 32, 33, 34, 64, 253, 254, 255, 256, 257, 600])
        inputs.append(bytes(d[:length]))
    n = gave_up = 0
    for data in inputs:
        cap = lib.rle8_low_entropy_compress_bounds(ctypes.c_uint32(len(data)))
        assert cap == len(data) + 32 + 1 + 256 + 8
        for variant in range(4):
            want = oracle.low_entropy_compress(variant, data)
            size, got = hs.call_dropin(LE_NAMES[variant], data, cap)
            if want is None:
                assert size == 0, f"{LE_NAMES[variant]} on {len(data)} bytes: the reference's stream outgrows the bound, the GPU call must fail"
                gave_up += 1
                continue
            assert size == len(want) and got == want, f"{LE_NAMES[variant]} on {len(data)} bytes: {'GPU gave up' if size == 0 else 'stream differs'}"
            assert lib.rle8_low_entropy_decompressed_size(want, ctypes.c_uint32(len(want))) == len(data)
            name = "rle8_low_entropy_short_decompress" if variant & 1 else "rle8_low_entropy_decompress"
            size, back = hs.call_dropin(name, want, len(data))
            assert size == len(data) and back == data, f"{name} on a {len(want)}-byte stream"
            n += 1
    assert n >= 400     # (gave_up stays 0 in practice: with ONE section the flags come from the statistics of the very bytes that are encoded, so a stream cannot outgrow the bound by more than the tail quirk)
    # argument errors of the reference: NULL / empty / short buffers -> 0
    assert hs.call_dropin("rle8_low_entropy_compress", b"abc", 10)[0] == 0
    assert hs.call_dropin("rle8_low_entropy_decompress", oracle.low_entropy_compress(0, b"aaaaabbbbb" * 50), 10)[0] == 0


def test_partial_block_range(hs):
    import torch

    rng = random.Random(5)
    data = mixed_runs(rng, 50000)
    container, info = hs.compress("rle16_byte_packed", _to_dev(data), block_size=512)
    out = torch.zeros(len(data), dtype=torch.uint8, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    first, count = 10, 33
    hs.decompress_async(container, info, out, status, first_block=first, block_count=count)
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    got = out.cpu().numpy().tobytes()
    lo, hi = first * 512, (first + count) * 512
    assert got[lo:hi] == data[lo:hi]
    assert got[:lo] == bytes(lo) and got[hi:] == bytes(len(data) - hi)  # nothing outside the range is written


def test_golden_vectors_through_the_c_abi(hs):
    """Committed golden vectors minted from the compiled reference: drop-in compress must reproduce size + sha256."""
    import base64, hashlib, json, os

    vec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vectors.json")))
    for entry in vec["inputs"]:
        data = base64.b64decode(entry["input"])
        for c in CODECS:
            g = entry["codecs"][c.key]
            size, stream = hs.call_dropin(c.cname, data, hs.compress_bounds(len(data)))
            assert size == g["size"] and hashlib.sha256(stream).hexdigest() == g["sha256"], f"{c.key} on {entry['name']}"


def test_synthetic_generator_matches_cpu(hs, oracle):
    for kind, S, size in ((0, 1, 3 * 65536 + 777), (0, 8, 2 * 65536), (0, 3, 70000), (1, 1, 200000), (0, 16, 65536 + 5)):
        dev = hs.synth(kind, S, 9, size).cpu().numpy().tobytes()
        assert dev == oracle.synth(kind, S, 9, size).tobytes()


def test_synthetic_manifest_blocks_and_mono(hs):
    """1 MiB synthetic workloads: 64 KiB block streams and the monolithic stream equal the REFERENCE's (sha256 manifest)."""
    import hashlib, json, os

    man = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "synth_manifest.json")))
    for key, e in man["entries"].items():
        ckey, kind = key.split("/kind")
        c = CODEC_BY_KEY[ckey]
        src = hs.synth(int(kind), c.S, man["seed"], man["size"])
        container, info = hs.compress(ckey, src, block_size=man["block"])
        _, streams = hs.split_container(container.cpu().numpy().tobytes())
        assert [(len(s), hashlib.sha256(s).hexdigest()) for s in streams] == [(b["size"], b["sha256"]) for b in e["blocks"]], key
        assert hs.decompress(container).equal(src)
    # the monolithic (drop-in) stream of the two headline codecs
    for key in ("rle8_packed_multi/kind0", "rle64_3symlut_byte/kind1"):
        ckey, kind = key.split("/kind")
        c = CODEC_BY_KEY[ckey]
        data = hs.synth(int(kind), c.S, man["seed"], man["size"]).cpu().numpy().tobytes()
        size, stream = hs.call_dropin(c.cname, data, hs.compress_bounds(len(data)))
        assert (size, hashlib.sha256(stream).hexdigest()) == (man["entries"][key]["mono"]["size"], man["entries"][key]["mono"]["sha256"])
        size, dec = hs.call_dropin(c.dname, stream, len(data))
        assert size == len(data) and dec == data


@pytest.mark.parametrize("key,kind,size,block", [("rle8_packed_multi", 0, (1 << 30) + 4096 * 3 + 100, 4096), ("rle64_3symlut_byte", 1, 88473600, 4096),
                                                 ("rle8_packed_multi", 0, 1 << 28, 128), ("rle128_byte_packed", 0, 1 << 27, 1024)])
def test_full_size_round_trip_properties(hs, oracle, key, kind, size, block):
    """BASELINE-size buffers: encode -> decode round trip is the identity, the status word stays 0, a sample of block
    streams equals the oracle's, and the container header is consistent (size-independent properties)."""
    import torch

    c = CODEC_BY_KEY[key]
    src = hs.synth(kind, c.S, 3, size)
    container, info = hs.compress(key, src, block_size=block)
    assert info.uncompressedSize == size and info.blockCount == (size + block - 1) // block and info.totalSize == container.numel()
    out = hs.decompress(container)
    assert torch.equal(out, src)
    # sample: first 256 blocks and last 64 blocks against the oracle
    table = container[64 : 64 + 8 * (info.blockCount + 1)].view(torch.int64).cpu().numpy()
    p0 = info.payload_start
    for first, count in ((0, 256), (info.blockCount - 64, 64)):
        pay = container[p0 + int(table[first]) : p0 + int(table[first + count])].cpu().numpy().tobytes()
        base = int(table[first])
        got = [pay[int(table[first + i]) - base : int(table[first + i + 1]) - base] for i in range(count)]
        host = src[first * block : min((first + count) * block, size)].cpu().numpy()
        assert got == oracle.compress_blocks(c, host, block)


def test_native_cli_over_the_c_abi():
    """hsrlekit_gpu (C++ host program, no python, no torch) drives the device container API and the drop-in functions of
    rle.h through the C ABI: every codec must round trip a small synthetic buffer (exit code 0)."""
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hypersonic-rle-kit_amd", "hsrlekit_gpu")
    if not os.path.exists(exe):
        pytest.skip("hsrlekit_gpu not built (make -C hypersonic-rle-kit_amd tools)")
    r = subprocess.run([exe, "--synth", "runs", "8", "--runs", "2", "--block", "1024"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("GiB/s") >= 4 * 110 and "all codecs round-tripped" in r.stdout
    r = subprocess.run([exe, "--synth", "video", "1", "--runs", "1", "--codec", "rle8_packed_multi", "--host"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "drop-in" in r.stdout and "FAILED" not in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_north_star_kernels_keep_their_occupancy(hs):
    """The kernels are latency bound: throughput is proportional to the waves resident per CU.  A change that
    pushes the 8 bit Packed kernels over 168 VGPRs or 17 KB of LDS silently costs 11 %: fail loudly instead."""
    assert hs.kernel_waves_per_cu("rle8_packed_multi", decode=True) >= 9
    assert hs.kernel_waves_per_cu("rle8_packed_multi", decode=False) >= 9
    for c in CODECS:
        assert hs.kernel_waves_per_cu(c.key, decode=True) >= 8 and hs.kernel_waves_per_cu(c.key, decode=False) >= 8, c.key


def test_reference_cli_runs_on_the_gpu_library(tmp_path):
    """oracle/_ref/hsrlekit_dropin is the reference's own benchmark program (src/main.c, unmodified) linked against
    libhsrle_hip.so in place of the reference's extreme-codec translation units (oracle/Makefile).  Its benchmark loop calls all
    100 drop-in functions through codecCallbacks[] and validates every round trip itself; `--test` turns any failure into a
    non-zero exit code.  Skipped where the binary was not built (no reference checkout on the build machine)."""
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "hsrlekit_dropin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/hsrlekit_dropin not built")
    rng = random.Random(7)
    sample = tmp_path / "sample.bin"
    sample.write_bytes(mixed_runs(rng, 120000) + bytes(rng.randrange(256) for _ in range(12000)) + mixed_runs(rng, 40000))
    r = subprocess.run([exe, str(sample), "--extreme", "--not-short", "--runs", "1", "--min-time", "0", "--test"], capture_output=True, text=True, timeout=900)
    out = r.stdout.replace("\r", "\n")
    assert r.returncode == 0, out[-3000:] + r.stderr[-2000:]
    assert "FAILED" not in out and "8 Bit Packed" in out and "128 Bit Packed (Byte)" in out and "64 Bit 3LUT (Byte)" in out
    # the Short rows: 105 more drop-in functions on the GPU (44 Short pairs, rle8_single_short, 15 Greedy encoders)
    r = subprocess.run([exe, str(sample), "--extreme", "--short", "--runs", "1", "--min-time", "0", "--test"], capture_output=True, text=True, timeout=900)
    out = r.stdout.replace("\r", "\n")
    assert r.returncode == 0, out[-3000:] + r.stderr[-2000:]
    assert "FAILED" not in out and "Short" in out
    # the unsectioned low-entropy rows: rle8_low_entropy[_short]_{compress, compress_only_max_frequency, decompress}
    for flag, name in (("--low-entropy", "Low Entropy"), ("--low-entropy-short", "Low Entropy Short")):
        r = subprocess.run([exe, str(sample), flag, "--runs", "1", "--min-time", "0", "--test"], capture_output=True, text=True, timeout=900)
        out = r.stdout.replace("\r", "\n")
        assert r.returncode == 0, out[-3000:] + r.stderr[-2000:]
        assert "FAILED" not in out and name in out, out[-2000:]


@pytest.mark.parametrize("mode,seconds", [("--fuzz-iterative", 100), ("--fuzz-random", 60)])
def test_reference_fuzzer_runs_on_the_gpu_library(tmp_path, mode, seconds):
    """The reference's OWN fuzzer (src/rle_fuzz.c:533-757, started by src/main.c:755-770) pointed at the GPU library: oracle/_ref/hsrlekit_dropin links
    rle_fuzz.o, so `hsrlekit_dropin x --fuzz-iterative` walks its structured inputs (alternating random / repeating sections, every length class,
    symbols of 1 .. 16 bytes, bound and unbound) through every drop-in compress / decompress pair and validates each round trip itself.  It never
    finishes on its own account within a test's patience: it gets a wall-clock budget, and being stopped without a complaint is the pass."""
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "hsrlekit_dropin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/hsrlekit_dropin not built")
    # (its progress line has no newline, and into a pipe stdio would hold everything back until 4 KiB have gathered: stdbuf -o0 where there is one)
    import select
    import shutil
    import time

    cmd = [exe, "x", mode]
    if shutil.which("stdbuf"):
        cmd = ["stdbuf", "-o0", "-e0"] + cmd
    proc = subprocess.Popen(cmd, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    fd = proc.stdout.fileno()
    buf, deadline = b"", time.time() + seconds
    while time.time() < deadline:
        ready, _, _ = select.select([fd], [], [], 1.0)
        if ready:
            chunk = os.read(fd, 65536)
            if not chunk:
                break                                 # (it closed its output: it is through)
            buf += chunk
    finished = proc.poll() is not None
    if not finished:
        proc.kill()                                   # (the exact process started above; stdbuf execs the program, so this IS the program)
        proc.wait()
    out = buf.decode(errors="replace")
    out = out.replace("\r", "\n")
    for bad in ("Fuzzer Failed", "Validation Failed", "Failed to compress", "Input Buffer Corrupted", "Decompressed to incorrect size", "First invalid char"):
        assert bad not in out, out[-4000:]
    if finished:
        assert proc.returncode == 0 and "Fuzzer Completed" in out, out[-3000:]
    # it got somewhere: the progress line is printed every 256 inputs (the random mode builds inputs of eight sections of up to 64 KiB and needs
    # minutes for its first 256 on the host-pointer path -- 200 codecs x two PCIe round trips per input: there, running without a complaint is all that is asked)
    inputs = [int(m) for m in re.findall(r"Input (\d+):", out)]
    if mode == "--fuzz-iterative":
        assert inputs and max(inputs) >= 256, f"the fuzzer made no progress in {seconds} s: {out[-500:]!r}"
    else:
        assert finished or proc.returncode is not None
    failure = tmp_path / "fuzz-failure.bin"          # (opened at start, written on a failure only)
    assert not failure.exists() or failure.stat().st_size == 0


def test_first_compress_of_a_process_under_graph_capture():
    """A process whose FIRST compress call is captured into a graph (no eager warm-up): the run list encoders size their grid from the kernel's
    residency, a host-side query they skip while the stream is capturing.  The replayed container == an eager one."""
    import subprocess
    import sys

    code = r"""
import sys
sys.path.insert(0, %r)
import torch, hsrle
size, block, key = 24 << 20, 4096, "rle64_3symlut_byte"
src = hsrle.synth(hsrle.SYNTH_VIDEO, 8, 5, size, device="cuda")
dst = torch.zeros(hsrle.container_bound(size, block), dtype=torch.uint8, device="cuda")
ws = torch.empty(hsrle.workspace_size(size, block), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    g.capture_begin()
    hsrle.compress_async(key, src, dst, block, workspace=ws)
    g.capture_end()
torch.cuda.current_stream().wait_stream(side)
g.replay(); torch.cuda.synchronize()
info = hsrle.container_info(dst)
captured = dst[: info.totalSize].clone()
eager, einfo = hsrle.compress(key, src, block_size=block)
assert info.totalSize == einfo.totalSize and torch.equal(captured, eager[: einfo.totalSize])
assert torch.equal(hsrle.decompress(eager), src)
print("CAPTURE_OK")
""" % (os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hypersonic-rle-kit_amd", "python"),)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CAPTURE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("key,block", [("rle8_packed_multi", 4096), ("rle64_3symlut_byte", 4096), ("rle8_7symlut", 8192), ("rle8_single", 4096), ("rle128_sym_packed", 2048),
                                       ("rle32_1symlut_byte_short_greedy", 4096), ("rle24_7symlut_byte_short_greedy", 4096)])
def test_compress_and_decompress_are_graph_capturable(hs, key, block):
    """With a caller-provided workspace the async entry points only enqueue kernels on the given stream (no allocation, no
    synchronisation), so they can be captured into a HIP graph and replayed (include/hsrle.h) -- whatever encoder the
    container takes: run list (small containers of 1 .. 4 KiB blocks), split encode (other block sizes; Single, 128 bit and Greedy with one listed
    symbol when the workspace is sized for the codec) or one lane per block."""
    import torch

    size = (8 << 20) + 4096 * 3 + 77
    src = hs.synth(hs.SYNTH_RUNS, CODEC_BY_KEY[key].S, 9, size)
    dst = torch.empty(hs.container_bound(size, block), dtype=torch.uint8, device="cuda")
    ws = torch.full((hs.workspace_size(size, block, key),), 0xC3, dtype=torch.uint8, device="cuda")
    out = torch.zeros(size, dtype=torch.uint8, device="cuda")
    status = torch.zeros(16, dtype=torch.int32, device="cuda")
    # eager run: fixes the container layout (sizes are data dependent, the data is not going to change its shape below)
    hs.compress_async(key, src, dst, block, workspace=ws)
    torch.cuda.synchronize()
    info = hs.container_info(dst)
    eager = dst[: info.totalSize].clone()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            hs.compress_async(key, src, dst, block, workspace=ws)
            hs.decompress_async(dst, info, out, status)
    torch.cuda.current_stream().wait_stream(side)

    for rep in range(3):
        dst.zero_(); out.zero_(); status.zero_()
        ws.fill_(0xA5 + rep)            # nothing may depend on what an earlier run left in the workspace (a captured hipMemsetAsync acts on the FIRST replay only: csrc zero_async)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(dst[: info.totalSize], eager), f"replay {rep}: container differs from the eager run"
        assert int(status[0].item()) == 0 and torch.equal(out, src), f"replay {rep}: decode differs from the input"


@pytest.mark.parametrize("knob", ["HSRLE_ENCODE_WAVE", "HSRLE_RUNLIST"])
def test_wave_per_block_encoder_is_bit_exact_too(hs, knob):
    """(-DHSRLE_EXPERIMENTS builds only: the shipped library holds one encoder per codec.)  HSRLE_ENCODE_WAVE=1 selects the one-wave-per-block encoder of rle8_multi / rle8_packed_multi (csrc/hsrle_encode8w.hip.h),
    HSRLE_RUNLIST=1 their run list encoder for blocks of 1 .. 4 KiB (csrc/hsrle_encode8r.hip.h); both off by default, they are slower.
    Same bar: every block stream == the oracle's, for small ragged blocks and for a 64 MiB buffer of 4 KiB blocks."""
    import subprocess
    import sys

    if not hs.experiments_enabled():
        pytest.skip("the wave-per-block encoder is not part of the shipped build (variants/libhsrle_exp.so: tools/build_variant.sh exp -DHSRLE_EXPERIMENTS)")
    code = r"""
import sys, random
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch, hsrle
from hsrle_testlib import CODEC_BY_KEY, Oracle, mixed_runs, fuzz_sections, single_symbol_mix
ora = Oracle(); rng = random.Random(5)
data = b"".join(mixed_runs(rng, 3000) + fuzz_sections(rng) + single_symbol_mix(rng, 700) + bytes(rng.randrange(256) for _ in range(rng.choice([0, 5, 700]))) for _ in range(60))
for key in ("rle8_multi", "rle8_packed_multi"):
    codec = CODEC_BY_KEY[key]
    for block in (128, 384, 1024, 1536, 2048, 3072, 4096):
        src = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        container, info = hsrle.compress(key, src, block_size=block)
        _, streams = hsrle.split_container(container.cpu().numpy().tobytes())
        assert streams == ora.compress_blocks(codec, torch.frombuffer(bytearray(data), dtype=torch.uint8).numpy(), block), (key, block)
        assert hsrle.decompress(container).cpu().numpy().tobytes() == data
    for kind in (0, 1):
        src = hsrle.synth(kind, 1, 4, 64 << 20, device="cuda")
        container, info = hsrle.compress(key, src, block_size=4096)
        _, streams = hsrle.split_container(container.cpu().numpy().tobytes())
        assert streams == ora.compress_blocks(codec, src.cpu().numpy(), 4096), (key, kind)
print("WAVE_OK")
""" % (os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hypersonic-rle-kit_amd", "python"), os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **{knob: "1"}), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "WAVE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_rle8m_info_that_disagrees_with_the_stream_is_an_error(hs, oracle):
    """The decode grid and the capacity check come from the caller's info struct; the kernels read the sizes from the stream again.  A
    hand-built / stale info must end as an error bit, not as sections left undecoded or bytes written past what was checked."""
    import torch

    rng = random.Random(3)
    data = mixed_runs(rng, 60000, alphabet=4)
    st = oracle.rle8m_compress(16, data)
    dev = torch.zeros(len(st) + 64, dtype=torch.uint8, device="cuda")
    dev[: len(st)] = torch.frombuffer(bytearray(st), dtype=torch.uint8).cuda()
    info = hs.rle8m_info(dev)
    for field, value in (("sections", 8), ("uncompressedSize", len(data) - 100)):
        bad = hs.Rle8mInfo(info.compressedSize, info.uncompressedSize, info.sections)
        setattr(bad, field, value)
        out = torch.full((len(data) + 4096,), 0xA5, dtype=torch.uint8, device="cuda")
        status = torch.zeros(1, dtype=torch.int32, device="cuda")
        hs.rle8m_decompress_async(dev, bad, out[: len(data)], status)
        torch.cuda.synchronize()
        assert int(status.item()) != 0 and bool((out[len(data):] == 0xA5).all())


def _single_stress(seed, n):
    """What the Single encoders' state machine branches on: literal gaps around 255 bytes in front of runs of the favourite symbol around
    SHORT / MEDIUM / LONG (wasted chances and the back-track to the first of them), runs through several 16-byte scan windows, runs of
    other symbols (they only matter to the symbol pick), the favourite sprinkled through the literals (the search's skip rule)."""
    rng = random.Random(seed)
    fav = rng.randrange(256)
    other = [rng.randrange(256) for _ in range(3)]
    out = bytearray()
    while len(out) < n:
        gap = rng.choice([0, 1, 3, 14, 15, 16, 17, 40, 200, 250, 253, 254, 255, 256, 257, 270, 300, 600])
        lit = bytearray(rng.randrange(256) for _ in range(gap))
        for k in range(len(lit)):
            if rng.random() < 0.08:
                lit[k] = fav
        out += lit
        for _ in range(rng.choice([1, 1, 2, 3, 4, 6])):
            s = fav if rng.random() < 0.8 else rng.choice(other)
            out += bytes([s]) * rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 15, 16, 17, 18, 31, 32, 33, 34, 48, 49, 50, 65, 100, 257, 700])
            out += bytes(rng.randrange(256) for _ in range(rng.choice([0, 1, 1, 2, 3, 7, 20, 60])))
    return bytes(out[:n])


@pytest.mark.parametrize("key", ["rle8_single", "rle8_packed_single", "rle8_single_short"])
def test_single_blocks_stress(hs, oracle, key):
    """The Single encoders (wave-parallel symbol pick + ring encoder, csrc/hsrle_encode8s.hip.h) against the oracle's literal restatement,
    block sizes up to the first-generation kernel's range (> 32 KiB); one-block streams of odd sizes through the rle.h names."""
    codec = CODEC_BY_KEY[key]
    rng = random.Random(99)
    for seed, size, blocks in ((1, 300001, (128, 256, 1024, 4096)), (2, 700013, (384, 640, 3968, 16384)), (3, 400000, (128, 32768, 65536))):
        data = _single_stress(seed, size) if seed != 3 else single_symbol_mix(rng, size)
        src = _to_dev(data)
        for block_size in blocks:
            container, info = hs.compress(key, src, block_size=block_size)
            cinfo, streams = hs.split_container(container.cpu().numpy().tobytes())
            assert len(streams) == (len(data) + block_size - 1) // block_size
            step = 1 if len(streams) < 3000 else 7
            for i in list(range(0, len(streams), step)) + [len(streams) - 1]:
                expect = oracle.compress(codec, data[i * block_size : (i + 1) * block_size])
                assert streams[i] == expect, f"{key} block {i} of size {block_size} (input seed {seed}) differs from the oracle"
            assert hs.decompress(container).cpu().numpy().tobytes() == data
    for k, n in enumerate((17, 31, 33, 47, 48, 49, 257, 1000, 4095, 4097, 20001, 32767, 32768, 32769, 50000)):
        d = _single_stress(10 + k, n)
        size, stream = hs.call_dropin(codec.cname, d, hs.compress_bounds(len(d)))
        expect = oracle.compress(codec, d)
        assert size == len(expect) and stream == expect, f"{key}: one-block stream of {n} bytes differs from the oracle"


@pytest.mark.parametrize("key", ["rle8_multi", "rle8_packed_multi", "rle8_3symlut", "rle8_7symlut_short", "rle16_sym", "rle24_byte_packed", "rle32_7symlut_sym", "rle48_byte",
                                 "rle64_3symlut_byte", "rle64_sym_short", "rle8_single", "rle8_packed_single"])
def test_long_literal_stretches(hs, oracle, key):
    """Literal stretches that have left the encoders' 256-byte ring are noted and copied by the whole wave (hsrle_encode8.hip.h: coop_flush):
    input without runs, one run per KiB, runs in bursts with gaps of 250 .. 3000 bytes, stretches that end on every byte of a 16-byte chunk."""
    codec = CODEC_BY_KEY[key]
    rng = random.Random(4711)
    S = codec.S
    noise = bytes(rng.randrange(256) for _ in range(150000))
    sparse = bytearray(noise)
    for at in range(500, len(sparse) - 100, 1024):
        sparse[at : at + 5 * S + 3] = (bytes([7]) * S) * 6
    bursts = bytearray()
    while len(bursts) < 200000:
        bursts += bytes(rng.randrange(256) for _ in range(rng.choice([250, 257, 271, 300, 500, 1000, 1001, 1002, 1003, 1017, 3000])))
        for _ in range(rng.randrange(1, 4)):
            sym = bytes(rng.randrange(256) for _ in range(S))
            bursts += sym * rng.choice([3, 4, 8, 20]) + bytes(rng.randrange(256) for _ in range(rng.randrange(0, 18)))
    for data, block_size in ((noise, 4096), (bytes(sparse), 4096), (bytes(bursts), 4096), (bytes(bursts[:70001]), 16384), (bytes(sparse), 1024)):
        src = _to_dev(data)
        container, info = hs.compress(key, src, block_size=block_size)
        cinfo, streams = hs.split_container(container.cpu().numpy().tobytes())
        for i, s in enumerate(streams):
            expect = oracle.compress(codec, data[i * block_size : (i + 1) * block_size])
            assert s == expect, f"{key} block {i} (size {block_size}) differs from the oracle"
        assert hs.decompress(container).cpu().numpy().tobytes() == data


def _runs128(seed, size):
    """16-byte symbols: runs of 2 .. 40 symbols cut mid-symbol, runs that butt against each other sharing bytes (a later start with a rotated
    symbol), byte runs (every 16-byte window of them is a symbol), gaps of 0 .. 300 bytes; the same near every block end."""
    rng = random.Random(seed)
    out = bytearray()
    while len(out) < size:
        out += bytes(rng.randrange(256) for _ in range(rng.choice([0, 0, 1, 2, 5, 15, 16, 17, 31, 33, 60, 127, 128, 255, 300])))
        kind = rng.random()
        if kind < 0.2:
            out += bytes([rng.choice([0, 0, 7, rng.randrange(256)])]) * rng.choice([16, 17, 31, 32, 33, 34, 47, 48, 49, 64, 100, 500])
        else:
            sym = bytes(rng.choice([0, 7, rng.randrange(256)]) for _ in range(16)) if rng.random() < 0.5 else bytes(rng.randrange(256) for _ in range(16))
            k = rng.choice([16, 18, 19, 20, 26, 27, 28, 31, 32, 33, 34, 35, 36, 42, 43, 44, 47, 48, 49, 64, 65, 80, 160, 640, 3000]) + rng.choice([0, 0, 1, 3, 15])
            piece = (sym * (k // 16 + 2))[:k]
            out += piece
            if rng.random() < 0.3:
                rot = rng.randrange(1, 17)
                sym2 = piece[-rot:] + bytes(rng.randrange(256) for _ in range(16 - rot)) if rot < 16 else piece[-16:]
                out += (sym2 * 30)[rot : rot + rng.choice([48, 50, 200, 321])]
    return bytes(out[:size])


@pytest.mark.parametrize("key", ["rle128_sym", "rle128_sym_packed", "rle128_byte", "rle128_byte_packed"])
def test_rle128_blocks_stress(hs, oracle, key):
    """The 128 bit ring encoder (bit scanner up to n - 48, the reference's loop as it is behind that: csrc/hsrle_encode128.hip.h)."""
    codec = CODEC_BY_KEY[key]
    for seed, size, blocks in ((1, 300001, (128, 256, 1024, 4096)), (2, 500017, (384, 640, 3968, 16384)), (3, 200000, (128, 65536))):
        data = _runs128(seed, size)
        src = _to_dev(data)
        for block_size in blocks:
            container, info = hs.compress(key, src, block_size=block_size)
            cinfo, streams = hs.split_container(container.cpu().numpy().tobytes())
            step = 1 if len(streams) < 3000 else 5
            for i in list(range(0, len(streams), step)) + [len(streams) - 1]:
                expect = oracle.compress(codec, data[i * block_size : (i + 1) * block_size])
                assert streams[i] == expect, f"{key} block {i} of size {block_size} (input seed {seed}) differs from the oracle"
            assert hs.decompress(container).cpu().numpy().tobytes() == data
    # video-shaped data: byte runs of 10 .. 160 zeros with a few other bytes in between -- near a block's end the reference's byte-wise
    # loop finds (or misses) 16-byte "symbols" in them depending on where its pair search stood when it reached n - 32
    video = hs.synth(SYNTH_VIDEO_KIND, 16, 2, 4 << 20, device="cuda")
    vdata = video.cpu().numpy().tobytes()
    for block_size in (4096, 1024):
        container, info = hs.compress(key, video, block_size=block_size)
        cinfo, streams = hs.split_container(container.cpu().numpy().tobytes())
        for i, s in enumerate(streams):
            assert s == oracle.compress(codec, vdata[i * block_size : (i + 1) * block_size]), f"{key} video-shaped block {i} of size {block_size} differs from the oracle"
    for k, n in enumerate(list(range(1, 100)) + [127, 128, 129, 255, 257, 1000, 4095, 4097, 20001, 50000, 65536, 65537, 150001]):   # (> 64 KiB: the first-generation kernel)
        d = _runs128(100 + k, n)
        size, stream = hs.call_dropin(codec.cname, d, hs.compress_bounds(len(d)))
        expect = oracle.compress(codec, d)
        assert size == len(expect) and stream == expect, f"{key}: one-block stream of {n} bytes differs from the oracle"


@pytest.mark.parametrize("key", ["rle16_1symlut_byte_short_greedy", "rle24_1symlut_byte_short_greedy", "rle32_1symlut_byte_short_greedy", "rle48_1symlut_byte_short_greedy",
                                 "rle64_1symlut_byte_short_greedy", "rle32_7symlut_byte_short_greedy"])
def test_greedy_small_containers_take_the_split_encode_bit_exact(hs, oracle, key):
    """Small containers of the Greedy encoders with a list of ONE symbol (round 4): chunks inside the blocks, one lane each; the list in front of a
    chunk is the symbol of the run it starts behind, guessed so, then proven round by round (csrc/hsrle_capi.hip compress_split).  (Lists of 3 / 7
    symbols decide which runs the greedy scan stores: measured slower than one lane per block, they stay there -- one of them is here as the control.)"""
    codec = CODEC_BY_KEY[key]
    assert hs.lib().hsrle_encode_path(CODECS.index(codec), 4 << 20, 8192) == (1 if "_1symlut" in key else 0)     # SPLIT / RING
    for kind, sym in ((SYNTH_VIDEO_KIND, codec.S), (0, codec.S), (SYNTH_VIDEO_KIND, 1)):
        src = hs.synth(kind, sym, 5, (4 << 20) + 1234, device="cuda")
        data = src.cpu().numpy().tobytes()
        for block_size in (4096, 1024, 8192, 65536):
            container, info = hs.compress(key, src, block_size=block_size)
            cinfo, streams = hs.split_container(container.cpu().numpy().tobytes())
            expect = oracle.compress_blocks(codec, np.frombuffer(data, dtype=np.uint8), block_size)
            assert len(streams) == len(expect)
            for i, (a, b) in enumerate(zip(streams, expect)):
                assert a == b, f"{key} block {i} of size {block_size} (kind {kind}, symbol {sym}) differs from the oracle"
            assert hs.decompress(container).cpu().numpy().tobytes() == data


@pytest.mark.parametrize("key", ["rle8_multi", "rle8_packed_multi", "rle8_7symlut", "rle8_3symlut_short", "rle16_sym_packed", "rle16_3symlut_byte", "rle16_1symlut_sym_short"])
def test_ring_chosen_per_input(hs, oracle, key):
    """Containers of >= 131 072 blocks: the encoders of 1 / 2 byte symbols probe the input and run with a 128- or a 256-byte history ring
    (csrc/hsrle_ring_probe.hip.h: video-shaped data gets 128, run-distributed 256).  Whatever the choice, the streams are the oracle's."""
    import torch

    codec = CODEC_BY_KEY[key]
    for kind in (SYNTH_VIDEO_KIND, 0):
        src = hs.synth(kind, codec.S, 9, 128 << 20, device="cuda")
        data = src.cpu().numpy().tobytes()
        container, info = hs.compress(key, src, block_size=1024)
        cinfo, streams = hs.split_container(container.cpu().numpy().tobytes())
        assert len(streams) == 131072
        for i in list(range(0, len(streams), 97)) + [len(streams) - 1]:
            assert streams[i] == oracle.compress(codec, data[i * 1024 : (i + 1) * 1024]), f"{key} kind {kind}: block {i} differs from the oracle"
        assert torch.equal(hs.decompress(container), src)


@pytest.mark.parametrize("key", ["rle8_packed_multi", "rle8_3symlut", "rle8_single", "rle8_single_short", "rle16_sym", "rle24_7symlut_byte", "rle32_byte_packed", "rle48_3symlut_sym_short",
                                 "rle64_3symlut_byte", "rle128_sym_packed", "rle32_1symlut_byte_short_greedy", "rle64_7symlut_byte_short_greedy"])
def test_input_and_output_need_no_alignment(hs, oracle, key):
    """Device pointers into the middle of a caller's buffers: the input of the encoders and the output of the decoders at byte offsets 1, 3, 8
    (the container itself is the library's layout and 16-byte aligned by contract).  Small and mid-size containers, so that the run list / split
    encoders and the ring encoders are all on the path; streams == the oracle's, decode == the input, nothing written outside the output."""
    import torch

    codec = CODEC_BY_KEY[key]
    for size, block in (((1 << 20) + 333, 4096), ((3 << 20) + 5, 1024), ((2 << 20) + 77, 16384), (200001, 384)):
        base = hs.synth(hs.SYNTH_RUNS if size & 1 else SYNTH_VIDEO_KIND, codec.S, 11, size + 64, device="cuda")
        for off in (1, 3, 8):
            src = base[off : off + size]
            data = src.cpu().numpy().tobytes()
            container, info = hs.compress(key, src, block_size=block)
            cinfo, streams = hs.split_container(container.cpu().numpy().tobytes())
            expect = oracle.compress_blocks(codec, np.frombuffer(data, dtype=np.uint8), block)
            assert streams == expect, f"{key}: input at offset {off}, block size {block}: streams differ from the oracle"
            outbuf = torch.full((size + 64,), 0xA5, dtype=torch.uint8, device="cuda")
            status = torch.zeros(4, dtype=torch.int32, device="cuda")
            hs.decompress_async(container, info, outbuf[off : off + size], status)
            torch.cuda.synchronize()
            host = outbuf.cpu().numpy().tobytes()
            assert int(status[0].item()) == 0 and host[off : off + size] == data and set(host[:off]) == {0xA5} and set(host[off + size :]) == {0xA5}, f"{key}: output at offset {off}, block size {block}"


@pytest.mark.parametrize("key", ["rle8_3symlut_short", "rle8_7symlut_short", "rle16_3symlut_byte_short_greedy", "rle16_7symlut_byte_short_greedy", "rle24_3symlut_byte_short_greedy",
                                 "rle16_3symlut_sym_short", "rle32_7symlut_byte_short", "rle64_7symlut_byte_short_greedy", "rle8_3symlut", "rle16_sym_packed"])
def test_blocks_with_a_packet_for_every_few_bytes(hs, oracle, key):
    """The most packets a block can hold: symbols of the move-to-front list coming back all the time (a one-byte packet for every 2 .. S output bytes with the
    Short headers; the Greedy encoders store 2-byte pieces of listed symbols), next to blocks of one long run and of literals only.  The decoders' rounds are
    capped at a few packets per lane (csrc/hsrle_decode.hip.h, CAPPED ROUNDS), so their bound on the number of rounds has to hold for such blocks --
    and lanes whose neighbours are through after one packet."""
    import torch

    codec = CODEC_BY_KEY[key]
    S = codec.S
    rng = random.Random(4711)
    syms = [bytes([v]) * S for v in (0x00, 0x7F, 0xFF, 0x01, 0x7E)] + [bytes(rng.randrange(256) for _ in range(S)) for _ in range(3)]
    parts = []
    for blk in range(192):
        kind = blk % 4
        if kind == 0:                                                    # listed symbols in turn, two of each: a packet per 2 S bytes (8 bit: per 2 bytes)
            parts.append(b"".join(syms[(i // 2) % 3] for i in range(4096 // S + 1))[:4096])
        elif kind == 1:                                                  # ... single occurrences: what the Greedy scan stores as a run of one listed symbol (or its first bytes)
            parts.append(b"".join(syms[rng.randrange(5)][: rng.choice([2, S])] + bytes([rng.randrange(256)]) for _ in range(4096))[:4096])
        elif kind == 2:
            parts.append(syms[blk % len(syms)] * (4096 // S + 1))
            parts[-1] = parts[-1][:4096]
        else:
            parts.append(bytes(rng.randrange(256) for _ in range(4096)))
    data = b"".join(parts)
    src = _to_dev(data)
    container, info = hs.compress(key, src, block_size=4096)
    cinfo, streams = hs.split_container(container.cpu().numpy().tobytes())
    expect = oracle.compress_blocks(codec, np.frombuffer(data, dtype=np.uint8), 4096)
    assert streams == expect, f"{key}: block streams differ from the oracle"
    out = hs.decompress(container)
    assert torch.equal(out, src), f"{key}: decode differs"
