"""Monolithic reference streams on the GPU: the rle.h-named decoders and hsrle_decompress_mono_dev build an
entry-point index of the ONE stream (csrc/hsrle_index.hip.h: speculative walks per stream region, a proving pass, repairs) and run
the block kernel from the entry records.  Bar: the decode equals the input the oracle's (= the reference's) encoder was given, for
every codec, whatever the index parameters are -- tiny regions and look-backs force wrong guesses, jumps over regions and repairs."""
import hashlib
import json
import os
import random

import numpy as np
import pytest

from hsrle_testlib import CODECS, CODEC_BY_KEY, FUZZ_LENGTHS, SYNTH_RUNS, SYNTH_VIDEO, fuzz_sections, mixed_runs, single_symbol_mix

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def hs():
    import torch

    assert torch.cuda.is_available(), "these tests need a GPU"
    import hsrle

    hsrle.lib()
    yield hsrle
    hsrle.mono_tuning(0, 0, 0)


def _dev_stream(stream):
    """Stream bytes in device memory as hsrle_decompress_mono_dev wants them: 128-byte aligned, 64 readable bytes behind the end."""
    import torch

    t = torch.zeros(len(stream) + 64, dtype=torch.uint8, device="cuda")
    t[: len(stream)] = torch.frombuffer(bytearray(stream), dtype=torch.uint8).cuda()
    return t


def _mixed_input(seed, size):
    rng = random.Random(seed)
    parts, n = [], 0
    while n < size:
        k = rng.randrange(5)
        if k == 0:
            d = fuzz_sections(rng)
        elif k == 1:
            d = mixed_runs(rng, rng.choice([40, 200, 1000, 3000]))
        elif k == 2:
            d = single_symbol_mix(rng, rng.choice([64, 333, 3000]))
        elif k == 3:
            d = bytes(rng.randrange(256) for _ in range(rng.choice([1, 7, 130, 300, 700])))      # literal stretches: the chain jumps over regions
        else:
            d = fuzz_sections(rng, lengths=FUZZ_LENGTHS, max_sections=3)
        parts.append(d)
        n += len(d)
    return b"".join(parts)[:size]


TUNINGS = [(0, 0, 0), (128, 64, 40), (256, 300, 16), (128, 32, 1024), (1024, 4096, 64)]


@pytest.mark.parametrize("codec", CODECS[:94] + [CODECS[109]], ids=lambda c: c.key)
def test_every_codec_through_the_index(hs, oracle, codec):
    data = _mixed_input(4242 + CODECS.index(codec), 40000)
    stream = oracle.compress(codec, data)
    assert stream is not None
    for tune in TUNINGS:
        hs.mono_tuning(*tune)
        size, dec = hs.call_dropin(codec.dname, stream, len(data))
        assert size == len(data) and dec == data, f"{codec.key} tuning {tune}"
    hs.mono_tuning(0, 0, 0)


@pytest.mark.parametrize("key", ["rle8_7symlut", "rle8_3symlut", "rle16_7symlut_sym", "rle8_multi", "rle64_7symlut_byte_short", "rle24_sym_packed"])
def test_repair_streaks_with_tiny_regions(hs, oracle, key):
    """Formats whose wrong walks do not die + regions far smaller than the look-back they would need: most guesses are wrong, the chain
    jumps over listed regions, repair walks run on into their neighbours (one writer per region record: claimed through the mark word)."""
    codec = CODEC_BY_KEY[key]
    for seed in range(8):
        data = _mixed_input(977 + CODECS.index(codec) + seed * 1000, 40000)
        stream = oracle.compress(codec, data)
        for tune in ((128, 64, 40), (256, 300, 16), (128, 32, 1024), (128, 48, 8)):
            hs.mono_tuning(*tune)
            size, dec = hs.call_dropin(codec.dname, stream, len(data))
            assert size == len(data) and dec == data, f"{key} seed {seed} tuning {tune}"
    hs.mono_tuning(0, 0, 0)


@pytest.mark.parametrize("key", ["rle8_single", "rle8_packed_single", "rle8_single_short"])
def test_single_symbol_streams(hs, oracle, key):
    codec = CODEC_BY_KEY[key]
    rng = random.Random(9)
    for tune in TUNINGS:
        hs.mono_tuning(*tune)
        for n in (1, 17, 300, 5000, 70000):
            data = single_symbol_mix(rng, n)
            stream = oracle.compress(codec, data)
            size, dec = hs.call_dropin(codec.dname, stream, len(data))
            assert size == len(data) and dec == data, f"{key} n={n} tuning {tune}"
    hs.mono_tuning(0, 0, 0)


@pytest.mark.parametrize("key", ["rle8_packed_multi", "rle8_multi", "rle8_7symlut", "rle24_byte_packed", "rle64_3symlut_byte", "rle128_sym", "rle32_1symlut_byte_short"])
def test_shapes_that_stress_the_chain(hs, oracle, key):
    """One literal spanning thousands of regions, one run spanning all blocks, runs longer than 2^16 / 2^24, alternating both."""
    codec = CODEC_BY_KEY[key]
    rng = random.Random(5)
    noise = bytes(rng.randrange(256) for _ in range(300000))
    cases = [noise, bytes(3000000), noise[:100] + bytes(70000) + noise[:33] + b"\x07" * 17000000 + noise[:5000],
             b"".join(noise[i * 50 : i * 50 + 50] + bytes([i & 255]) * (3 + i % 40) for i in range(4000))]
    for tune in ((0, 0, 0), (128, 64, 40), (4096, 8192, 1024)):
        hs.mono_tuning(*tune)
        for data in cases:
            stream = oracle.compress(codec, data)
            size, dec = hs.call_dropin(codec.dname, stream, len(data))
            assert size == len(data) and dec == data, f"{key} len {len(data)} tuning {tune}"
    hs.mono_tuning(0, 0, 0)


def test_malformed_streams_fail_cleanly(hs, oracle):
    """Streams are data: garbage behind an intact header, truncation, wrong sizes -> 0, never a hang or a write outside the output."""
    import torch

    rng = random.Random(3)
    for key in ("rle8_packed_multi", "rle8_3symlut", "rle16_sym", "rle48_7symlut_byte", "rle8_multi_short", "rle64_3symlut_sym_short"):
        codec = CODEC_BY_KEY[key]
        data = mixed_runs(rng, 200000)
        stream = bytearray(oracle.compress(codec, data))
        for tune in ((0, 0, 0), (128, 64, 40)):
            hs.mono_tuning(*tune)
            bad = bytearray(stream)
            for j in range(len(bad) // 2, len(bad)):
                bad[j] = rng.randrange(256)
            size, _ = hs.call_dropin(codec.dname, bytes(bad), len(data))
            assert size in (0, len(data))                                   # garbage may by chance still be a stream of the right size
            lie = bytearray(stream)
            lie[0:4] = (len(data) - 1).to_bytes(4, "little")               # header claims one byte less than the packets produce
            assert hs.call_dropin(codec.dname, bytes(lie), len(data))[0] == 0
            cut = bytearray(stream[: len(stream) * 2 // 3])
            cut[4:8] = len(cut).to_bytes(4, "little")                        # truncated stream with a consistent header
            assert hs.call_dropin(codec.dname, bytes(cut), len(data))[0] == 0
            # device form: the bytes behind the output stay untouched
            t = _dev_stream(bytes(bad))
            out = torch.full((len(data) + 4096,), 0xA5, dtype=torch.uint8, device="cuda")
            try:
                hs.mono_decompress_dev(key, t, dst=out[: len(data)])
            except hs.HsrleError:
                pass
            assert bool((out[len(data):] == 0xA5).all())
    hs.mono_tuning(0, 0, 0)


@pytest.mark.parametrize("key,kind,size", [("rle8_packed_multi", SYNTH_RUNS, 64 << 20), ("rle8_packed_multi", SYNTH_VIDEO, 64 << 20), ("rle64_3symlut_byte", SYNTH_VIDEO, 88473600),
                                           ("rle8_3symlut", SYNTH_RUNS, 32 << 20), ("rle16_sym_packed", SYNTH_RUNS, 32 << 20), ("rle32_7symlut_byte_short", SYNTH_VIDEO, 32 << 20)])
def test_device_resident_mono_decode_of_synthetic_workloads(hs, oracle, key, kind, size):
    """BASELINE-shaped buffers as ONE stream each (the oracle's = the reference's encoder writes it), decoded in device memory."""
    import torch

    codec = CODEC_BY_KEY[key]
    data = oracle.synth(kind, codec.S, 2, size)
    stream = oracle.compress(codec, data.tobytes())
    t = _dev_stream(stream)
    out, stats = hs.mono_decompress_dev(key, t, return_stats=True)
    assert out.numel() == size and torch.equal(out.cpu(), torch.from_numpy(data)), f"{key}: decode differs"
    regions, rounds, rewalked, lookback = stats
    print(f"{key}: regions {regions}, repair rounds {rounds}, walked again {rewalked}, look-back {lookback}")
    assert rounds <= 64, f"{key}: {rounds} repair rounds ({rewalked} of {regions} regions guessed wrong, look-back {lookback})"


# ---- the encode side: ONE stream written by many lanes (csrc/hsrle_mono_encode.hip.h) ----

# every multi-symbol codec of 8 .. 64 bit symbols: plain, Packed, LUT and the Short family (not Single, 128 bit, Greedy)
# many-lane monolithic encode: every codec (round 3: + 8 bit Single -- global symbol pick, cuts behind long runs of the symbol -- and the 128 bit
# codecs; round 4: + the Greedy encoders -- cuts behind stretches the scan cannot enter too late, lists guessed and proven -- and rle8_single_short)
MONO_ENC_KEYS = [c.key for c in CODECS]
MONO_LIST_KEYS = [k for k in MONO_ENC_KEYS if "symlut" in k]


def _wide_run_mix(seed, size, S):
    """Runs of S-byte symbols with lengths around the thresholds, cut mid-symbol, butting against each other with shared bytes (the
    overlapping periodic stretches that make a run start later, with a rotated symbol) and separated by 0 .. 300 literal bytes."""
    rng = random.Random(seed)
    out = bytearray()
    while len(out) < size:
        out += bytes(rng.randrange(256) for _ in range(rng.choice([0, 0, 1, 2, 5, 20, 60, 126, 127, 128, 140, 254, 255, 256, 300])))
        sym = bytes(rng.choice([0, 7, 200, rng.randrange(256)]) for _ in range(S)) if rng.random() < 0.7 else bytes([rng.randrange(4)]) * S
        k = rng.choice([2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 20, 40, 100]) * S // rng.choice([1, 1, 2]) + rng.randrange(S)
        piece = (sym * (k // S + 2))[:k]
        out += piece
        if rng.random() < 0.3:                                            # a second run that shares its first bytes with the end of the first
            rot = rng.randrange(1, S + 1)
            sym2 = piece[-rot:] + bytes(rng.randrange(256) for _ in range(S - rot)) if rot < S else piece[-S:]
            out += (sym2 * 30)[rot : rot + rng.choice([S * 3, S * 12, S * 20 + 1])]
    return bytes(out[:size])


def _run_mix(seed, size, counts):
    """Runs whose lengths sit around the thresholds of the emit rules, separated by 0 .. 140 literal bytes (range field 1 / 4 bytes)."""
    rng = random.Random(seed)
    out = bytearray()
    while len(out) < size:
        out += bytes(rng.randrange(256) for _ in range(rng.choice([0, 1, 2, 5, 20, 60, 126, 127, 128, 140, 254, 255, 256, 300])))
        out += bytes([rng.choice([0, 7, 7, 200, rng.randrange(256)])]) * rng.choice(counts)
    return bytes(out[:size])


@pytest.mark.parametrize("key", MONO_ENC_KEYS)
def test_mono_encode_is_the_reference_stream(hs, oracle, key):
    """Byte-identical with the sequential encoder, whatever the piece size: tiny pieces put a cut behind nearly every long run."""
    codec = CODEC_BY_KEY[key]
    rng = random.Random(11)
    cases = [_mixed_input(1 + k, n) for k, n in enumerate((70, 130, 500, 3000, 40000, 150000))]
    cases += [_run_mix(5, 60000, [1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15, 40]), _run_mix(6, 60000, [11, 12, 13, 30, 64, 65, 300]),
              bytes(200000), bytes(rng.randrange(256) for _ in range(100000)), b"\x05" * 70 + bytes(range(256)) * 100 + b"\x06" * 40,
              _run_mix(7, 30000, [12, 13]) + b"\x09" * 20, _run_mix(8, 30000, [12, 13]) + b"\x09" * 20 + b"abc", _run_mix(9, 5000, [20]) + bytes(50)]
    if "single" in key:
        # the pick: a run that reaches the end / U - 16, runs across many 4 KiB pieces, two symbols with nearly equal gains, nothing to pick
        cases += [b"ab" * 3000 + b"\x07" * 9000 + b"xy" * 10 + b"\x07" * 17, b"\x07" * 5000 + b"q" + b"\x09" * 5000 + b"r" * 3, bytes(range(256)) * 40,
                  b"z" * 33 + b"\x01" * 20000 + b"\x02" * 20001 + b"k", _run_mix(15, 300000, [2, 3, 4, 5, 8, 9, 10, 11, 40, 5000]), b"\x00" * 100 + b"\xff" * 15,
                  _run_mix(16, 70000, [4, 8, 10]) + b"\x07" * 40, bytes([7]) * 12289, bytes([7]) * 4096 + b"x" + bytes([7]) * 4097]
    if codec.S > 1:
        cases += [_wide_run_mix(20 + k, n, codec.S) for k, n in enumerate((300, 5000, 60000, 60000, 150000))]
        cases += [_wide_run_mix(30, 40000, 1), bytes(range(7)) * 9000]
    for tune in ((0, 0, 0), (0, 64, 0), (0, 100, 0), (0, 1000, 0), (0, 5000, 0)):
        hs.mono_tuning(*tune)
        for d in cases:
            size, stream = hs.call_dropin(codec.cname, d, hs.compress_bounds(len(d)))
            expect = oracle.compress(codec, d)
            assert size == len(expect) and stream == expect, f"{key} len {len(d)} tuning {tune}"
    hs.mono_tuning(0, 0, 0)


def _few_symbols(seed, size, S, symbols, counts):
    """Runs of a handful of symbols: the move-to-front list in front of a piece is whatever far earlier pieces left in it."""
    rng = random.Random(seed)
    syms = [bytes(rng.randrange(256) for _ in range(S)) for _ in range(symbols)]
    out = bytearray()
    while len(out) < size:
        out += bytes(rng.randrange(256) for _ in range(rng.choice([0, 1, 3, 9, 40])))
        # mostly the two favourites, now and then another one: most pieces see fewer distinct symbols than the list holds
        s = syms[rng.randrange(min(2, symbols))] if rng.random() < 0.97 else syms[rng.randrange(symbols)]
        out += s * rng.choice(counts)
    return bytes(out[:size])


@pytest.mark.parametrize("key", MONO_LIST_KEYS)
def test_mono_encode_lists_from_far_back(hs, oracle, key):
    """The guessed lists come from roll-ups over 64 and 4096 pieces: inputs where a list entry survives thousands of pieces."""
    codec = CODEC_BY_KEY[key]
    S = codec.S
    cases = [_few_symbols(3, 700000, S, 9, [1, 2, 3, 4, 12 // S + 3, 30]), _few_symbols(4, 300000, S, 3, [2, 3, 4, 5, 20]),
             _few_symbols(5, 300000, S, 1, [3, 14]), bytes(10000) + _few_symbols(6, 200000, S, 12, [3, 4, 5])]
    for tune in ((0, 64, 0), (0, 0, 0)):
        hs.mono_tuning(*tune)
        for d in cases:
            size, stream = hs.call_dropin(codec.cname, d, hs.compress_bounds(len(d)))
            expect = oracle.compress(codec, d)
            assert size == len(expect) and stream == expect, f"{key} len {len(d)} tuning {tune}"
            rounds, bad0, bad1, bad2 = hs.mono_encode_stats()
            assert rounds <= 8, f"{key}: {rounds} repair rounds ({bad0}, {bad1}, {bad2} wrong guesses)"
    hs.mono_tuning(0, 0, 0)


@pytest.mark.parametrize("key,kind,size", [("rle8_3symlut", SYNTH_RUNS, 32 << 20), ("rle8_7symlut_short", SYNTH_VIDEO, 32 << 20), ("rle16_3symlut_sym", SYNTH_RUNS, 32 << 20),
                                           ("rle24_7symlut_byte", SYNTH_RUNS, 32 << 20), ("rle32_1symlut_sym_short", SYNTH_RUNS, 32 << 20), ("rle48_3symlut_byte_short", SYNTH_VIDEO, 32 << 20),
                                           ("rle64_7symlut_sym", SYNTH_RUNS, 32 << 20),
                                           ("rle8_packed_multi", SYNTH_RUNS, 64 << 20), ("rle8_packed_multi", SYNTH_VIDEO, 88473600), ("rle8_multi", SYNTH_RUNS, 32 << 20),
                                           ("rle8_multi_short", SYNTH_VIDEO, 32 << 20), ("rle16_sym_packed", SYNTH_RUNS, 32 << 20), ("rle24_byte", SYNTH_RUNS, 32 << 20),
                                           ("rle32_byte_packed", SYNTH_RUNS, 32 << 20), ("rle48_sym", SYNTH_RUNS, 32 << 20), ("rle64_byte_short", SYNTH_RUNS, 32 << 20),
                                           ("rle64_sym_packed", SYNTH_VIDEO, 32 << 20), ("rle16_3symlut_byte_short_greedy", SYNTH_RUNS, 32 << 20),
                                           ("rle32_7symlut_byte_short_greedy", SYNTH_RUNS, 32 << 20), ("rle64_1symlut_byte_short_greedy", SYNTH_RUNS, 32 << 20),
                                           ("rle8_single_short", SYNTH_RUNS, 32 << 20), ("rle8_single_short", SYNTH_VIDEO, 32 << 20)])
def test_device_resident_mono_encode(hs, oracle, key, kind, size):
    import torch

    codec = CODEC_BY_KEY[key]
    src = hs.synth(kind, codec.S, 2, size, device="cuda")
    stream, chunks = hs.mono_compress_dev(key, src, return_chunks=True)
    expect = oracle.compress(codec, src.cpu().numpy().tobytes())
    assert stream.cpu().numpy().tobytes() == expect, f"{key}: stream differs from the oracle's ({chunks} chunks)"
    assert chunks > size // 32768


def test_greedy_mono_encode_gives_up_cleanly_when_the_lists_do_not_settle(hs, oracle):
    """The greedy scan's runs depend on the list in front of a chunk (it tries the listed symbols and their prefixes), so on data with many
    short runs of many symbols a wrong guess moves one chunk per repair round: the device call then says UNSUPPORTED after its round limit
    (nothing written that a caller could mistake for a stream), and the host entry point falls back to one lane -- same stream either way."""
    import hsrle

    codec = CODEC_BY_KEY["rle32_7symlut_byte_short_greedy"]
    src = hs.synth(SYNTH_VIDEO, 4, 2, 4 << 20, device="cuda")
    data = src.cpu().numpy().tobytes()
    expect = oracle.compress(codec, data)
    try:
        stream = hs.mono_compress_dev(codec.key, src)
        assert stream.cpu().numpy().tobytes() == expect
    except hsrle.HsrleError as e:
        assert e.status == hsrle.ERR_UNSUPPORTED
    size, stream = hs.call_dropin(codec.cname, data, hs.compress_bounds(len(data)))
    assert size == len(expect) and stream == expect
