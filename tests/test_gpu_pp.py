"""Position-parallel 8 bit encoder (csrc/hsrle_encode8p.hip.h: rle8_multi, rle8_packed_multi, containers of >= 1 024 blocks of <= 4 KiB):
every block stream == the oracle's, whatever the data does to its phases (runs across lanes and block ends, chains of short runs without a
long one, more candidates than the small list holds, ragged tails, unaligned stream starts).  Reference: src/rle8_extreme_cpu.h:86-344, :936-1099."""
import numpy as np
import pytest

from hsrle_testlib import CODEC_BY_KEY, SYNTH_RUNS, SYNTH_VIDEO

pytestmark = pytest.mark.gpu

KEYS = ["rle8_multi", "rle8_packed_multi"]


@pytest.fixture(scope="module")
def hs():
    import torch

    assert torch.cuda.is_available(), "these tests need a GPU"
    import hsrle

    hsrle.lib()  # fails loudly if the HIP library is missing
    return hsrle


def _dense(rng, n, lengths, alphabet, literal_max):
    """runs of the given lengths over a small alphabet with short literal gaps: chains of candidates that wait for their left neighbour"""
    out = np.empty(n + 4096, dtype=np.uint8)
    at = 0
    while at < n:
        L = int(rng.integers(0, literal_max + 1))
        out[at : at + L] = rng.integers(0, 256, L, dtype=np.uint8)
        at += L
        R = int(rng.choice(lengths))
        out[at : at + R] = rng.integers(0, alphabet)
        at += R
    return out[:n].copy()


def _cases():
    rng = np.random.default_rng(20251003)
    n = 4 << 20
    cases = {
        "zeros": np.zeros(n, dtype=np.uint8),
        "random": rng.integers(0, 256, n, dtype=np.uint8),
        "two_symbols": rng.integers(0, 2, n, dtype=np.uint8),                      # runs every other byte: > 448 candidates per 4 KiB block
        "threes": np.repeat(rng.integers(0, 3, n // 3 + 1, dtype=np.uint8), 3)[:n],  # runs of 3, 6, 9 ... of three symbols: one candidate per 3 bytes at most
        "short_chains": _dense(rng, n, [3, 3, 4, 5, 9, 10], 4, 3),                  # no run of 11: one pass per candidate
        "mixed": _dense(rng, n, [2, 3, 4, 5, 6, 7, 10, 11, 12, 31, 32, 33, 64, 65, 127, 128, 129, 130, 300, 5000], 256, 140),
        "long_literals": _dense(rng, n, [3, 4, 11, 40], 256, 900),                   # ranges beyond 127 / 255: the long range form
        "same_symbol": _dense(rng, n, [3, 3, 3, 4, 11, 12], 1, 5),                   # every run the last symbol again
    }
    return cases


@pytest.fixture(scope="module")
def cases():
    return _cases()


def _check(hs, oracle, key, data, block):
    import torch

    codec = CODEC_BY_KEY[key]
    src = torch.from_numpy(data).cuda()
    container, info = hs.compress(key, src, block_size=block)
    cinfo, streams = hs.split_container(container.cpu().numpy().tobytes())
    expect = oracle.compress_blocks(codec, data, block)
    assert len(streams) == len(expect)
    bad = [i for i, (a, b) in enumerate(zip(streams, expect)) if a != b]
    assert not bad, f"{key} block size {block}: {len(bad)} of {len(expect)} block streams differ from the oracle, first {bad[:8]}"
    assert torch.equal(hs.decompress(container), src)


@pytest.mark.parametrize("key", KEYS)
@pytest.mark.parametrize("name", ["zeros", "random", "two_symbols", "threes", "short_chains", "mixed", "long_literals", "same_symbol"])
def test_position_parallel_encoder_bit_exact(hs, oracle, cases, key, name):
    _check(hs, oracle, key, cases[name], 4096)


@pytest.mark.parametrize("key", KEYS)
@pytest.mark.parametrize("block,cut", [(128, 0), (256, 1), (1024, 77), (1536, 1535), (2048, 2047), (3968, 13), (4096, 4095), (4096, 4064), (4096, 4033)])
def test_position_parallel_encoder_block_sizes_and_ragged_tails(hs, oracle, cases, key, block, cut):
    data = np.concatenate([cases["mixed"][: 3 << 20], cases["short_chains"][: 1 << 20], cases["two_symbols"][: 1 << 20]])
    _check(hs, oracle, key, data[: data.size - cut], block)


@pytest.mark.parametrize("key", KEYS)
@pytest.mark.parametrize("kind", [SYNTH_RUNS, SYNTH_VIDEO])
def test_position_parallel_encoder_synthetic_workloads(hs, oracle, key, kind):
    data = oracle.synth(kind, 1, 11, (16 << 20) + 999)
    _check(hs, oracle, key, data, 4096)


# ---- the same two codecs, units of any length walked in 4 KiB windows (csrc/hsrle_encode8pw.hip.h): blocks above 4 KiB, monolithic streams ----
def _edge_runs(rng, n):
    """runs placed on the 4 KiB window edges: ending 0 .. 3 bytes in front of / behind an edge, runs of two across it, runs of many windows, literal stretches of many windows"""
    out = rng.integers(0, 256, n, dtype=np.uint8)
    at = 4096
    while at + 40000 < n:
        kind = int(rng.integers(0, 6))
        d = int(rng.integers(-3, 4))
        if kind == 0:
            L = int(rng.choice([2, 3, 4, 5, 6, 10, 11, 12]))
            out[at + d - L // 2 : at + d - L // 2 + L] = rng.integers(0, 4)
        elif kind == 1:
            L = int(rng.choice([2, 3, 4, 6, 11, 130, 300]))
            out[at + d - L : at + d] = rng.integers(0, 4)                    # ends at the edge + d
        elif kind == 2:
            L = int(rng.choice([2, 3, 4, 6, 11, 130, 300]))
            out[at + d : at + d + L] = rng.integers(0, 4)                    # starts at the edge + d
        elif kind == 3:
            out[at + d - 5000 : at + d + int(rng.choice([1, 2, 3, 4096, 9000]))] = rng.integers(0, 4)   # many windows long
        elif kind == 4:
            out[at - 3 : at + 3] = rng.integers(0, 256, 6, dtype=np.uint8)     # nothing at this edge
        else:
            v = int(rng.integers(0, 4))
            out[at - 8 : at - 4] = v; out[at - 2 : at + 2] = v; out[at + 5 : at + 8] = v   # short runs of ONE symbol around the edge (Packed: the same-symbol rule)
        at += 4096 * int(rng.choice([1, 1, 2, 3, 7]))
    return out


@pytest.fixture(scope="module")
def window_cases(cases):
    rng = np.random.default_rng(60606)
    out = dict(cases)
    out["edges"] = _edge_runs(rng, 4 << 20)
    return out


WINDOW_NAMES = ["zeros", "random", "two_symbols", "threes", "short_chains", "mixed", "long_literals", "same_symbol", "edges"]


@pytest.mark.parametrize("key", KEYS)
@pytest.mark.parametrize("name", WINDOW_NAMES)
@pytest.mark.parametrize("block,cut", [(4224, 0), (8192, 777), (12416, 1), (65536, 4097), (1 << 20, 65533)])
def test_windowed_encoder_blocks_bit_exact(hs, oracle, window_cases, key, name, block, cut):
    data = window_cases[name]
    assert hs.lib().hsrle_encode_path(hs.codec_id(key), data.size - cut, block) == 3      # HSRLE_PATH_POSITION_PARALLEL
    _check(hs, oracle, key, data[: data.size - cut], block)


@pytest.mark.parametrize("key", KEYS)
def test_windowed_encoder_takes_the_headline_buffer_with_large_blocks(hs, key):
    for block in (8192, 65536):
        assert hs.lib().hsrle_encode_path(hs.codec_id(key), 8 << 30, block) == 3


@pytest.mark.parametrize("key", KEYS)
@pytest.mark.parametrize("name", WINDOW_NAMES)
@pytest.mark.parametrize("n", [4 << 20, (1 << 20) + 4097, 123457, 8193])
def test_windowed_encoder_monolithic_stream_bit_exact(hs, oracle, window_cases, key, name, n):
    """hsrle_compress_mono_dev: ONE stream, cut behind long runs, a wave per chunk then a wave per window == the reference's stream (src/rle8_extreme_cpu.h:86-344)"""
    import torch

    part = window_cases[name][:n]
    got = hs.mono_compress_dev(key, torch.from_numpy(part).cuda()).cpu().numpy().tobytes()
    assert got == oracle.compress(CODEC_BY_KEY[key], part.tobytes())


@pytest.mark.parametrize("key", KEYS)
@pytest.mark.parametrize("kind", [SYNTH_RUNS, SYNTH_VIDEO])
def test_windowed_encoder_synthetic_workloads(hs, oracle, key, kind):
    import torch

    data = oracle.synth(kind, 1, 12, (24 << 20) + 4099)
    _check(hs, oracle, key, data, 65536)
    _check(hs, oracle, key, data, 16384)
    got = hs.mono_compress_dev(key, torch.from_numpy(data).cuda()).cpu().numpy().tobytes()
    assert got == oracle.compress(CODEC_BY_KEY[key], data.tobytes())


# ---- 2 .. 8 byte symbols, plain and Packed (csrc/hsrle_encodeSp.hip.h; reference: src/rleX_extreme_cpu_encode.h:14-609) ----
WIDE_KEYS = [f"rle{w}_{v}" for w in (16, 24, 32, 48, 64) for v in ("sym", "sym_packed", "byte", "byte_packed")]


def _periodic(rng, n, periods, alphabet, literal_max, lengths):
    """stretches with the given periods (in bytes; not only the codec's own symbol width), cut mid-symbol, butting against each other and against
    stretches of another period that share bytes with them: run starts that depend on where the run before ended"""
    out = np.empty(n + 8192 + literal_max + max(lengths), dtype=np.uint8)
    at = 0
    while at < n:
        L = int(rng.integers(0, literal_max + 1))
        out[at : at + L] = rng.integers(0, 256, L, dtype=np.uint8)
        at += L
        P = int(rng.choice(periods))
        sym = rng.integers(0, alphabet, P, dtype=np.uint8)
        R = int(rng.choice(lengths))
        out[at : at + R] = np.tile(sym, R // P + 2)[:R]
        at += R
    return out[:n].copy()


@pytest.fixture(scope="module")
def wide_cases():
    rng = np.random.default_rng(7051)
    n = 3 << 20
    return {
        "periods": _periodic(rng, n, [1, 2, 3, 4, 6, 8, 12, 16], 256, 40, [4, 5, 6, 7, 8, 9, 11, 12, 13, 16, 17, 18, 19, 23, 24, 25, 40, 64, 100, 300, 2000, 9000]),
        "butting": _periodic(rng, n, [2, 3, 4, 6, 8], 3, 0, [4, 6, 7, 8, 9, 12, 13, 14, 16, 17, 20, 24, 25, 33]),       # no literals, tiny alphabet: overlapping stretches everywhere
        "far_apart": _periodic(rng, n, [2, 3, 4, 6, 8], 256, 700, [8, 12, 16, 19, 24, 36, 48]),                         # ranges beyond 127 / 255
        "two_symbols": rng.integers(0, 2, n, dtype=np.uint8),
    }


@pytest.mark.parametrize("key", WIDE_KEYS)
@pytest.mark.parametrize("name", ["periods", "butting", "far_apart", "two_symbols"])
def test_position_parallel_wide_encoder_bit_exact(hs, oracle, wide_cases, key, name):
    _check(hs, oracle, key, wide_cases[name], 4096)


@pytest.mark.parametrize("key", ["rle16_byte_packed", "rle24_sym", "rle32_sym_packed", "rle48_byte", "rle64_byte_packed"])
@pytest.mark.parametrize("block,cut", [(128, 0), (384, 5), (1024, 77), (1536, 1535), (4096, 4095), (4096, 4033), (4096, 4081)])
def test_position_parallel_wide_encoder_block_sizes_and_ragged_tails(hs, oracle, wide_cases, key, block, cut):
    data = np.concatenate([wide_cases["periods"][: 2 << 20], wide_cases["butting"][: 1 << 19]])
    _check(hs, oracle, key, data[: data.size - cut], block)


# ---- the same codecs (and the LUT / Short ones below) with blocks above 4 KiB: walked in 4 KiB windows (csrc/hsrle_encodeSpw.hip.h) ----
WINDOWED_S_KEYS = WIDE_KEYS + [f"rle{w}_3symlut_{v}" for w in (24, 32, 48, 64) for v in ("sym", "byte")] + ["rle8_multi_short", "rle8_1symlut_short"] + \
    [f"rle{w}_{v}_short" for w in (16, 24, 32, 48, 64) for v in ("sym", "1symlut_sym", "byte", "1symlut_byte")] + [f"rle{w}_3symlut_{v}_short" for w in (48, 64) for v in ("sym", "byte")]


@pytest.fixture(scope="module")
def windowed_wide_cases(wide_cases):
    rng = np.random.default_rng(8088)
    n = 3 << 20
    out = dict(wide_cases)
    out["edges"] = _edge_runs(rng, n)
    out["long_literals"] = _periodic(rng, n, [2, 3, 4, 6, 8], 256, 8000, [8, 12, 16, 19, 24, 36, 48, 7000])      # literal stretches of several windows
    out["zeros"] = np.zeros(n, dtype=np.uint8)
    return out


def test_windowed_wide_encoder_covers_the_codecs_of_the_block_kernel(hs):
    assert len(WINDOWED_S_KEYS) == 54
    for key in WINDOWED_S_KEYS:
        assert hs.lib().hsrle_encode_path(hs.codec_id(key), 64 << 20, 8192) == 3, key
        assert hs.lib().hsrle_encode_path(hs.codec_id(key), 8 << 30, 65536) == 3, key


@pytest.mark.parametrize("key", WINDOWED_S_KEYS)
@pytest.mark.parametrize("block,cut", [(4224, 0), (8192, 777), (65536, 4097)])
def test_windowed_wide_encoder_blocks_bit_exact(hs, oracle, windowed_wide_cases, key, block, cut):
    for name in ("periods", "butting", "far_apart", "two_symbols", "edges", "long_literals", "zeros"):
        data = windowed_wide_cases[name]
        _check(hs, oracle, key, data[: data.size - cut], block)


@pytest.mark.parametrize("key", ["rle16_sym_packed", "rle24_byte", "rle32_3symlut_byte", "rle48_sym_short", "rle64_3symlut_byte_short", "rle64_byte_packed", "rle8_1symlut_short"])
def test_windowed_wide_encoder_large_blocks_and_wide_fields(hs, oracle, windowed_wide_cases, key):
    """blocks of 512 KiB: counts and ranges beyond 16 bits (the LUT / Short forms then carry 32 bit fields: src/rleX_Xsl.h:190-250, src/rleX_Xsl_short.h:216-357)"""
    rng = np.random.default_rng(99)
    n = 3 << 20
    data = rng.integers(0, 256, n, dtype=np.uint8)
    S = CODEC_BY_KEY[key].S
    data[100000:300000] = np.tile(rng.integers(0, 256, S, dtype=np.uint8), 200000 // S + 1)[:200000]         # a run of 200 000 bytes
    data[(1 << 19) + 90000 : (1 << 19) + 90000 + 4 * S + 3] = 7                                                  # a short run 90 000 literal bytes into the second block
    data[(1 << 20) + 70000 : (1 << 20) + 300000] = 0
    _check(hs, oracle, key, data, 1 << 19)
    _check(hs, oracle, key, windowed_wide_cases["zeros"], 1 << 19)


@pytest.mark.parametrize("key", ["rle16_byte_packed", "rle32_sym", "rle64_3symlut_byte", "rle24_1symlut_byte_short"])
@pytest.mark.parametrize("kind", [SYNTH_RUNS, SYNTH_VIDEO])
def test_windowed_wide_encoder_synthetic_workloads(hs, oracle, key, kind):
    data = oracle.synth(kind, CODEC_BY_KEY[key].S, 13, (24 << 20) + 4099)
    _check(hs, oracle, key, data, 65536)
    _check(hs, oracle, key, data, 12416)


# ---- ... and the general LUT kernel's codecs (csrc/hsrle_encodeLpw.hip.h): the list travels from window to window ----
WINDOWED_L_KEYS = ["rle8_3symlut", "rle8_7symlut", "rle16_3symlut_sym", "rle16_3symlut_byte"] + [f"rle{w}_7symlut_{v}" for w in (16, 24, 32, 48, 64) for v in ("sym", "byte")] + \
    [f"rle{w}_3symlut_{v}_short" for w in (16, 24, 32) for v in ("sym", "byte")] + [f"rle{w}_7symlut_{v}_short" for w in (16, 24, 32, 48, 64) for v in ("sym", "byte")]


def test_windowed_lut_encoder_covers_the_codecs_of_the_block_kernel(hs):
    assert len(WINDOWED_L_KEYS) == 30
    for key in WINDOWED_L_KEYS:
        assert hs.lib().hsrle_encode_path(hs.codec_id(key), 64 << 20, 8192) == 3, key
        assert hs.lib().hsrle_encode_path(hs.codec_id(key), 8 << 30, 65536) == 3, key
        assert hs.lib().hsrle_encode_path(hs.codec_id(key), 8 << 30, 1 << 20) != 3, key       # (1 MiB per block: the reference's 20 bit penalty thresholds are in reach)


@pytest.mark.parametrize("key", WINDOWED_L_KEYS)
@pytest.mark.parametrize("block,cut", [(4224, 0), (8192, 777), (65536, 4097)])
def test_windowed_lut_encoder_blocks_bit_exact(hs, oracle, windowed_wide_cases, cases, key, block, cut):
    for name in ("periods", "butting", "far_apart", "two_symbols", "edges", "long_literals", "zeros"):
        data = windowed_wide_cases[name]
        _check(hs, oracle, key, data[: data.size - cut], block)
    for name in ("threes", "short_chains", "same_symbol"):                      # (8 / 16 bit symbols: runs whose storing depends on the list)
        data = cases[name][: 2 << 20]
        _check(hs, oracle, key, data[: data.size - cut], block)


@pytest.mark.parametrize("key", ["rle8_7symlut", "rle16_3symlut_byte", "rle32_7symlut_sym", "rle64_7symlut_byte", "rle16_7symlut_sym_short", "rle24_3symlut_byte_short"])
def test_windowed_lut_encoder_large_blocks_and_wide_fields(hs, oracle, windowed_wide_cases, key):
    rng = np.random.default_rng(98)
    n = 3 << 20
    data = rng.integers(0, 256, n, dtype=np.uint8)
    S = CODEC_BY_KEY[key].S
    data[100000:300000] = np.tile(rng.integers(0, 256, S, dtype=np.uint8), 200000 // S + 1)[:200000]
    data[(1 << 19) + 90000 : (1 << 19) + 90000 + 4 * S + 3] = 7
    data[(1 << 20) + 70000 : (1 << 20) + 300000] = 0
    _check(hs, oracle, key, data, 1 << 19)
    _check(hs, oracle, key, windowed_wide_cases["zeros"], 1 << 19)


@pytest.mark.parametrize("key", ["rle8_3symlut", "rle8_7symlut", "rle16_7symlut_byte", "rle48_7symlut_sym", "rle32_7symlut_byte_short"])
@pytest.mark.parametrize("kind", [SYNTH_RUNS, SYNTH_VIDEO])
def test_windowed_lut_encoder_synthetic_workloads(hs, oracle, key, kind):
    data = oracle.synth(kind, CODEC_BY_KEY[key].S, 14, (24 << 20) + 4099)
    _check(hs, oracle, key, data, 65536)
    _check(hs, oracle, key, data, 12416)


@pytest.mark.parametrize("key", ["rle8_packed_multi", "rle8_multi", "rle16_sym_packed", "rle32_byte", "rle64_byte_packed", "rle24_sym"])
def test_windowed_encoder_blocks_of_many_mebibytes(hs, oracle, cases, key):
    """blocks of 16 MiB (4 096 windows each): positions, counts and ranges far beyond 16 bits -- a run of 5 MiB, literal stretches of 3 MiB, a ragged last block"""
    rng = np.random.default_rng(1234)
    n = (40 << 20) + 12345
    data = np.resize(cases["mixed"], n).copy()
    S = CODEC_BY_KEY[key].S
    data[(2 << 20) : (7 << 20)] = np.tile(rng.integers(0, 256, S, dtype=np.uint8), (5 << 20) // S + 1)[: 5 << 20]
    data[(9 << 20) : (12 << 20)] = rng.integers(0, 256, 3 << 20, dtype=np.uint8)
    data[(15 << 20) + 4000 : (17 << 20) + 100] = 0                                 # a run across the first block's end
    data[(33 << 20) : (36 << 20)] = rng.integers(0, 256, 3 << 20, dtype=np.uint8)
    assert hs.lib().hsrle_encode_path(hs.codec_id(key), n, 16 << 20) == 3
    _check(hs, oracle, key, data, 16 << 20)


# ---- 3 symbol LUT codecs of 3 .. 8 byte symbols (every run stored; the symbol's list index through streak heads) ----
LUT_KEYS = [f"rle{w}_3symlut_{a}" for w in (24, 32, 48, 64) for a in ("sym", "byte")]


@pytest.fixture(scope="module")
def lut_cases(wide_cases):
    rng = np.random.default_rng(991)
    n = 3 << 20
    few = _periodic(rng, n, [3, 4, 6, 8], 2, 6, [6, 8, 9, 12, 16, 17, 24, 32, 48, 130, 400])          # two byte values: the same few symbols come back all the time (indices 0 .. 2)
    defaults = np.frombuffer(bytes([0x00, 0x7F, 0xFF, 0x01]), dtype=np.uint8)
    init = _periodic(rng, n, [1], 4, 9, [6, 8, 12, 16, 24, 48])                                    # runs of single byte values ...
    init = defaults[init % 4]                                                                      # ... out of the list's initial entries and one more
    return dict(wide_cases, few_symbols=few, initial_entries=init)


@pytest.mark.parametrize("key", LUT_KEYS)
@pytest.mark.parametrize("name", ["periods", "butting", "far_apart", "two_symbols", "few_symbols", "initial_entries"])
def test_position_parallel_lut3_encoder_bit_exact(hs, oracle, lut_cases, key, name):
    _check(hs, oracle, key, lut_cases[name], 4096)


@pytest.mark.parametrize("key", ["rle24_3symlut_sym", "rle64_3symlut_byte"])
@pytest.mark.parametrize("block,cut", [(128, 0), (1024, 77), (1536, 1535), (4096, 4095), (4096, 4081)])
def test_position_parallel_lut3_encoder_block_sizes_and_ragged_tails(hs, oracle, lut_cases, key, block, cut):
    data = np.concatenate([lut_cases["few_symbols"][: 2 << 20], lut_cases["butting"][: 1 << 19]])
    _check(hs, oracle, key, data[: data.size - cut], block)


# ---- Short family without a list / with a one-symbol list, 2 .. 8 byte symbols (one-byte packed headers; src/rleX_Xsl_short.h:152-357): the emit rule is a
#      penalty on the shortest stored run that depends on the gap to the run before (and, with the list, on the symbol stored last) ----
SHORT_KEYS = [f"rle{w}_{v}" for w in (16, 24, 32, 48, 64) for v in ("sym_short", "byte_short", "1symlut_sym_short", "1symlut_byte_short")]
SHORT_KEYS += ["rle8_multi_short", "rle8_1symlut_short"]                                     # 8 bit: maximal runs of equal bytes through the same process_symbol
SHORT_KEYS += [f"rle{w}_3symlut_{a}_short" for w in (48, 64) for a in ("sym", "byte")]      # three-symbol list: where every run is stored (>= 6 byte symbols)


@pytest.fixture(scope="module")
def short_cases(lut_cases):
    rng = np.random.default_rng(4242)
    n = 3 << 20
    # gaps around the packed range limit (15) and the 3-byte form's (2047), run lengths around S + 2 .. S + 12 and the packed count limit
    near = _periodic(rng, n, [2, 3, 4, 6, 8], 4, 18, [4, 6, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 24, 30, 31, 32, 33, 34, 40])
    far = _periodic(rng, n, [2, 3, 4, 6, 8], 256, 2300, [8, 12, 16, 20, 24, 520, 530, 1040, 1100, 4090])
    return dict(lut_cases, near_limits=near, beyond_fields=far)


@pytest.mark.parametrize("key", SHORT_KEYS)
@pytest.mark.parametrize("name", ["periods", "butting", "far_apart", "two_symbols", "few_symbols", "initial_entries", "near_limits", "beyond_fields"])
def test_position_parallel_short_encoder_bit_exact(hs, oracle, short_cases, cases, key, name):
    _check(hs, oracle, key, short_cases[name], 4096)
    if key.startswith("rle8_") and name in cases:
        _check(hs, oracle, key, cases[name], 4096)


@pytest.mark.parametrize("key", ["rle8_multi_short", "rle8_1symlut_short"])
@pytest.mark.parametrize("name", ["zeros", "random", "threes", "short_chains", "mixed", "long_literals", "same_symbol"])
def test_position_parallel_short_encoder_8bit_data(hs, oracle, cases, key, name):
    _check(hs, oracle, key, cases[name], 4096)


@pytest.mark.parametrize("key", ["rle16_sym_short", "rle24_1symlut_byte_short", "rle32_byte_short", "rle48_1symlut_sym_short", "rle64_byte_short", "rle48_3symlut_byte_short", "rle64_3symlut_sym_short", "rle8_multi_short", "rle8_1symlut_short"])
@pytest.mark.parametrize("block,cut", [(128, 0), (384, 5), (1024, 77), (1536, 1535), (4096, 4095), (4096, 4081)])
def test_position_parallel_short_encoder_block_sizes_and_ragged_tails(hs, oracle, short_cases, key, block, cut):
    data = np.concatenate([short_cases["near_limits"][: 2 << 20], short_cases["butting"][: 1 << 19]])
    _check(hs, oracle, key, data[: data.size - cut], block)


def test_position_parallel_path_is_the_one_that_ran(hs):
    """The codecs above must not pass on another encoder: hsrle_encode_path says which one a container of this shape takes."""
    for key in KEYS + WIDE_KEYS + LUT_KEYS + SHORT_KEYS:
        assert hs.encode_path(key, 8 << 20, 4096) == hs.PATH_POSITION_PARALLEL, key


# ---- 8 bit Single (csrc/hsrle_encode8sp.hip.h; reference: src/rle8_extreme_cpu.c:53-153, src/rle8_extreme_cpu.h:346-700, :1103-1321) ----
SINGLE_KEYS = ["rle8_single", "rle8_packed_single", "rle8_single_short"]   # (the Short family's Single codec: same kernel, process_symbol's rule)


def _favourite(rng, n, sym, run_lengths, gaps, alphabet=256):
    """runs of ONE favourite byte between literal stretches that never hold it twice in a row: what the runs do is what the test says"""
    out = np.empty(n + 8192, dtype=np.uint8)
    at = 0
    while at < n:
        g = int(rng.choice(gaps))
        lit = rng.integers(0, alphabet, g, dtype=np.uint8)
        lit[lit == sym] = (sym + 1) % alphabet
        out[at : at + g] = lit
        at += g
        R = int(rng.choice(run_lengths))
        out[at : at + R] = sym
        at += R
    return out[:n].copy()


def _wasted_groups(rng, n):
    """the body's wasted-chances rule (:1244-1285) on runs of zeros: groups of short runs a few bytes apart behind gaps of more than 255 bytes -- and behind
    0 .. 70 stored runs, so that a group sits at any place of a 64-candidate round (literals never hold a zero)"""
    parts = []
    total = 0

    def add(a):
        nonlocal total
        parts.append(a)
        total += a.size

    while total < n:
        for _ in range(int(rng.integers(0, 71))):
            add(rng.integers(1, 250, int(rng.integers(1, 4)), dtype=np.uint8))
            add(np.zeros(int(rng.choice([2, 3, 4, 5, 9])), dtype=np.uint8))
        add(rng.integers(1, 250, int(rng.choice([250, 254, 255, 256, 257, 300, 700])), dtype=np.uint8))
        for _ in range(int(rng.choice([1, 2, 3, 3, 3, 4, 5, 7]))):
            add(np.zeros(int(rng.choice([2, 3, 4, 5, 6, 7])), dtype=np.uint8))
            add(rng.integers(1, 250, int(rng.choice([1, 2, 10, 60, 120, 125, 130, 240])), dtype=np.uint8))
    return np.concatenate(parts)[:n].copy()


@pytest.fixture(scope="module")
def single_cases():
    rng = np.random.default_rng(60606)
    n = 4 << 20
    return {
        "zeros": np.zeros(n, dtype=np.uint8),
        "random": rng.integers(0, 256, n, dtype=np.uint8),
        "two_symbols": rng.integers(0, 2, n, dtype=np.uint8),
        "three_symbols": rng.integers(0, 3, n, dtype=np.uint8),
        "short_runs": _favourite(rng, n, 7, [1, 2, 2, 3, 3, 4, 4, 5, 6, 7, 8, 9, 10, 11], [1, 1, 2, 3, 5, 14, 15, 16, 17, 31]),      # > 64 candidates per block
        "far_apart": _favourite(rng, n, 200, [2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 40, 300], [0, 1, 100, 250, 254, 255, 256, 257, 300, 900]),
        "long_runs": _favourite(rng, n, 0, [15, 16, 17, 31, 32, 33, 250, 255, 256, 257, 258, 259, 260, 1000, 5000], [1, 3, 16, 200, 400]),
        "wasted": _wasted_groups(rng, n),
    }


@pytest.mark.parametrize("key", SINGLE_KEYS)
@pytest.mark.parametrize("name", ["zeros", "random", "two_symbols", "three_symbols", "short_runs", "far_apart", "long_runs", "wasted"])
def test_position_parallel_single_encoder_bit_exact(hs, oracle, single_cases, key, name):
    _check(hs, oracle, key, single_cases[name], 4096)


@pytest.mark.parametrize("key", SINGLE_KEYS)
@pytest.mark.parametrize("block,cut", [(128, 0), (256, 1), (384, 5), (1024, 77), (1536, 1535), (2048, 2047), (3968, 13), (4096, 4095), (4096, 4081), (4096, 4080), (4096, 4079), (4096, 4064), (4096, 4033)])
def test_position_parallel_single_encoder_block_sizes_and_ragged_tails(hs, oracle, single_cases, key, block, cut):
    data = np.concatenate([single_cases["short_runs"][: 1 << 20], single_cases["far_apart"][: 1 << 20], single_cases["wasted"][: 1 << 20], single_cases["long_runs"][: 1 << 20]])
    _check(hs, oracle, key, data[: data.size - cut], block)


@pytest.mark.parametrize("key", SINGLE_KEYS)
@pytest.mark.parametrize("kind", [SYNTH_RUNS, SYNTH_VIDEO])
def test_position_parallel_single_encoder_synthetic_workloads(hs, oracle, key, kind):
    data = oracle.synth(kind, 1, 11, (16 << 20) + 999)
    _check(hs, oracle, key, data, 4096)


@pytest.mark.parametrize("key", SINGLE_KEYS)
def test_position_parallel_single_encoder_fuzz_blocks(hs, oracle, key):
    """the fuzz grammar of the reference's own fuzzer + inputs dominated by one favourite byte, as many small blocks"""
    import random

    from hsrle_testlib import FUZZ_LENGTHS, fuzz_sections, single_symbol_mix

    rng = random.Random(99)
    parts = []
    for _ in range(400):
        parts.append(fuzz_sections(rng, 8, FUZZ_LENGTHS))
        parts.append(single_symbol_mix(rng, rng.choice([300, 3000, 9000])))
    data = np.frombuffer(b"".join(parts), dtype=np.uint8).copy()
    for block in (128, 640, 4096):
        _check(hs, oracle, key, data, block)


# ---- 128 bit symbols (csrc/hsrle_encode128p.hip.h; reference: src/rle128_extreme_cpu.h:32-497) ----
KEYS_128 = ["rle128_sym", "rle128_sym_packed", "rle128_byte", "rle128_byte_packed"]


def _periodic128(rng, n, alphabet):
    """stretches of period 1 .. 32 bytes of many lengths around 16 / 32 / 48 bytes, cut mid-symbol, behind literal stretches of 0 .. 300 bytes"""
    out = bytearray()
    while len(out) < n:
        out += bytes(rng.randrange(alphabet) for _ in range(rng.choice([0, 1, 2, 5, 15, 16, 17, 40, 300])))
        P = rng.choice([1, 1, 2, 4, 8, 16, 16, 16, 32])
        sym = bytes(rng.randrange(alphabet) for _ in range(P))
        k = rng.choice([16, 17, 20, 31, 32, 33, 35, 36, 37, 40, 47, 48, 49, 63, 64, 65, 100, 300, 1000])
        out += (sym * (k // P + 2))[:k]
    return bytes(out[:n])


def _tails128(rng, n):
    """a stored run, a few other bytes, then the run's symbol again inside the last 32 bytes of the block: the byte steps at the end of the input
    (:270-300) -- Packed stores `runs` of 3 .. 16 equal bytes there -- and the pair at n - 32"""
    c = rng.randrange(256)
    lead = rng.choice([2, 3, 4, 5, 8, 12, 15, 16])
    sym = bytes([c]) * lead + bytes(rng.randrange(256) for _ in range(16 - lead))
    g = bytes(rng.choice([c, rng.randrange(256)]) for _ in range(rng.choice([0, 1, 2, 3, 5, 8, 13, 16])))
    endpart = (sym * 3)[: rng.choice([17, 18, 20, 24, 30, 31, 32, 33, 40])]
    k = max(32, n - len(g) - len(endpart) - rng.choice([0, 3, 50]))
    pre = bytes(rng.randrange(256) for _ in range(rng.choice([0, 3, 50])))
    blk = pre + (sym * (k // 16 + 1))[:k] + g + endpart
    return (blk + bytes(rng.randrange(256) for _ in range(n)))[:n]


@pytest.fixture(scope="module")
def cases128():
    import random

    rng = random.Random(128128)
    nrng = np.random.default_rng(128)
    n = 2 << 20
    out = {}
    for name, alphabet in (("periodic", 256), ("periodic_small", 3), ("periodic_one", 1)):
        out[name] = np.frombuffer(b"".join(_periodic128(rng, 4096, alphabet) for _ in range(n // 4096)), dtype=np.uint8).copy()
    out["tails_4096"] = np.frombuffer(b"".join(_tails128(rng, 4096) for _ in range(n // 4096)), dtype=np.uint8).copy()
    out["tails_256"] = np.frombuffer(b"".join(_tails128(rng, 256) for _ in range(n // 256)), dtype=np.uint8).copy()
    out["zeros"] = np.zeros(n, dtype=np.uint8)
    out["random"] = nrng.integers(0, 256, n, dtype=np.uint8)
    out["two_symbols"] = nrng.integers(0, 2, n, dtype=np.uint8)
    return out


@pytest.mark.parametrize("key", KEYS_128)
@pytest.mark.parametrize("name", ["periodic", "periodic_small", "periodic_one", "tails_4096", "zeros", "random", "two_symbols"])
def test_position_parallel_128_encoder_bit_exact(hs, oracle, cases128, key, name):
    _check(hs, oracle, key, cases128[name], 4096)


@pytest.mark.parametrize("key", KEYS_128)
def test_position_parallel_128_encoder_small_blocks_end_of_input_rules(hs, oracle, cases128, key):
    _check(hs, oracle, key, cases128["tails_256"], 256)


@pytest.mark.parametrize("key", KEYS_128)
@pytest.mark.parametrize("block,cut", [(128, 0), (256, 1), (384, 5), (1024, 77), (1536, 1535), (2048, 2047), (3968, 13), (4096, 4095), (4096, 4081), (4096, 4080), (4096, 4079), (4096, 4065), (4096, 4064), (4096, 4063), (4096, 4049), (4096, 4033)])
def test_position_parallel_128_encoder_block_sizes_and_ragged_tails(hs, oracle, cases128, key, block, cut):
    data = np.concatenate([cases128["periodic"][: 1 << 20], cases128["periodic_small"][: 1 << 19], cases128["tails_4096"][: 1 << 19]])
    _check(hs, oracle, key, data[: data.size - cut], block)


@pytest.mark.parametrize("key", KEYS_128)
@pytest.mark.parametrize("kind", [SYNTH_RUNS, SYNTH_VIDEO])
def test_position_parallel_128_encoder_synthetic_workloads(hs, oracle, key, kind):
    data = oracle.synth(kind, 16, 11, (16 << 20) + 999)
    _check(hs, oracle, key, data, 4096)


@pytest.mark.parametrize("key", KEYS_128)
def test_position_parallel_128_encoder_fuzz_blocks(hs, oracle, key):
    import random

    from hsrle_testlib import FUZZ_LENGTHS, fuzz_sections, mixed_runs

    rng = random.Random(1281)
    parts = []
    for _ in range(300):
        parts.append(fuzz_sections(rng, 8, FUZZ_LENGTHS))
        parts.append(mixed_runs(rng, rng.choice([300, 3000, 9000])))
    data = np.frombuffer(b"".join(parts), dtype=np.uint8).copy()
    for block in (128, 640, 4096):
        _check(hs, oracle, key, data, block)


# ---- LUT codecs whose packets depend on the list beyond a closed form (csrc/hsrle_encodeLp.hip.h): every 7 symbol LUT codec, the 3 symbol LUT codecs of 1 / 2 byte
#      symbols (reference: src/rleX_Xsl.h:93-346, src/rleX_Xsl_multibyte_encoder.h:18-370) ----
LUTG_KEYS = ["rle8_3symlut", "rle8_7symlut", "rle16_3symlut_sym", "rle16_3symlut_byte"] + [f"rle{w}_7symlut_{a}" for w in (16, 24, 32, 48, 64) for a in ("sym", "byte")]


def _list_marginal(rng, n, S, alphabet, lengths, gaps):
    """runs of exactly the lengths whose storing depends on the list (3 + penalty bytes) among longer ones, symbols out of a small alphabet so that they come
    back while they are still listed -- or just after they dropped out -- behind gaps on both sides of the range field's limit (63 / 127)"""
    out = np.empty(n + 8192, dtype=np.uint8)
    syms = rng.integers(0, 256, (alphabet, S), dtype=np.uint8)
    at = 0
    while at < n:
        g = int(rng.choice(gaps))
        lit = rng.integers(0, 256, g, dtype=np.uint8)
        out[at : at + g] = lit
        at += g
        R = int(rng.choice(lengths))
        out[at : at + R] = np.tile(syms[int(rng.integers(0, alphabet))], R // S + 2)[:R]
        at += R
    return out[:n].copy()


@pytest.fixture(scope="module")
def lutg_cases(lut_cases, cases):
    rng = np.random.default_rng(7007)
    n = 2 << 20
    out = dict(lut_cases)
    for name in ("zeros", "random", "threes", "short_chains", "mixed", "long_literals", "same_symbol"):
        out["b_" + name] = cases[name][:n]
    gaps = [0, 1, 2, 5, 20, 60, 61, 62, 63, 64, 100, 124, 125, 126, 127, 128, 200]
    for S in (1, 2, 3, 4, 6, 8):
        out[f"marginal{S}_few"] = _list_marginal(rng, n, S, 5, [3, 3, 4, 5, 5, 6, 7, 8, 2 * S, 2 * S + 1, 3 * S, 40], gaps)
        out[f"marginal{S}_nine"] = _list_marginal(rng, n, S, 9, [3, 4, 5, 5, 6, 2 * S, 2 * S + 1, 2 * S + 2, 3 * S + 1, 130, 131, 140], gaps)
    return out


def _symbytes(key):
    return int(key.split("_")[0][3:]) // 8


@pytest.mark.parametrize("key", LUTG_KEYS)
@pytest.mark.parametrize("name", ["periods", "butting", "far_apart", "two_symbols", "few_symbols", "initial_entries", "b_zeros", "b_random", "b_threes", "b_short_chains", "b_mixed",
                                  "b_long_literals", "b_same_symbol", "marginal_few", "marginal_nine", "marginal1_few", "marginal2_nine"])
def test_position_parallel_lut_general_encoder_bit_exact(hs, oracle, lutg_cases, key, name):
    if name in ("marginal_few", "marginal_nine"):
        name = name.replace("marginal", f"marginal{_symbytes(key)}")
    _check(hs, oracle, key, lutg_cases[name], 4096)


@pytest.mark.parametrize("key", ["rle8_3symlut", "rle8_7symlut", "rle16_3symlut_byte", "rle16_7symlut_sym", "rle32_7symlut_byte", "rle64_7symlut_sym"])
@pytest.mark.parametrize("block,cut", [(128, 0), (384, 5), (1024, 77), (1536, 1535), (4096, 4095), (4096, 4081)])
def test_position_parallel_lut_general_encoder_block_sizes_and_ragged_tails(hs, oracle, lutg_cases, key, block, cut):
    S = _symbytes(key)
    data = np.concatenate([lutg_cases[f"marginal{S}_few"][: 1 << 20], lutg_cases["few_symbols"][: 1 << 19], lutg_cases["b_short_chains"][: 1 << 19], lutg_cases["butting"][: 1 << 19]])
    _check(hs, oracle, key, data[: data.size - cut], block)


@pytest.mark.parametrize("key", LUTG_KEYS)
@pytest.mark.parametrize("kind", [SYNTH_RUNS, SYNTH_VIDEO])
def test_position_parallel_lut_general_encoder_synthetic_workloads(hs, oracle, key, kind):
    data = oracle.synth(kind, _symbytes(key), 11, (8 << 20) + 999)
    _check(hs, oracle, key, data, 4096)


# ---- Short family with a 3 / 7 symbol list where storing a run depends on the list (csrc/hsrle_encodeLp.hip.h; reference: src/rleX_Xsl_short.h:152-372) ----
SHORTL_KEYS = [f"rle{w}_3symlut_{a}_short" for w in (16, 24, 32) for a in ("sym", "byte")] + \
              [f"rle{w}_7symlut_{a}_short" for w in (16, 24, 32, 48, 64) for a in ("sym", "byte")]


@pytest.mark.parametrize("key", SHORTL_KEYS)
@pytest.mark.parametrize("name", ["periods", "butting", "far_apart", "two_symbols", "few_symbols", "initial_entries", "near_limits", "beyond_fields", "b_threes", "b_short_chains", "b_mixed",
                                  "b_same_symbol", "marginal_few", "marginal_nine", "marginal1_few"])
def test_position_parallel_short_list_encoder_bit_exact(hs, oracle, short_cases, lutg_cases, key, name):
    if name in ("marginal_few", "marginal_nine"):
        name = name.replace("marginal", f"marginal{_symbytes(key)}")
    data = short_cases[name] if name in short_cases else lutg_cases[name]
    _check(hs, oracle, key, data, 4096)


@pytest.mark.parametrize("key", ["rle16_3symlut_sym_short", "rle16_7symlut_byte_short", "rle24_3symlut_sym_short", "rle32_7symlut_byte_short", "rle64_7symlut_sym_short"])
@pytest.mark.parametrize("block,cut", [(128, 0), (384, 5), (1024, 77), (1536, 1535), (4096, 4095), (4096, 4081)])
def test_position_parallel_short_list_encoder_block_sizes_and_ragged_tails(hs, oracle, short_cases, lutg_cases, key, block, cut):
    S = _symbytes(key)
    data = np.concatenate([lutg_cases[f"marginal{S}_few"][: 1 << 20], short_cases["near_limits"][: 1 << 19], lutg_cases["b_short_chains"][: 1 << 19], short_cases["butting"][: 1 << 19]])
    _check(hs, oracle, key, data[: data.size - cut], block)


@pytest.mark.parametrize("key", SHORTL_KEYS)
@pytest.mark.parametrize("kind", [SYNTH_RUNS, SYNTH_VIDEO])
def test_position_parallel_short_list_encoder_synthetic_workloads(hs, oracle, key, kind):
    data = oracle.synth(kind, _symbytes(key), 11, (8 << 20) + 999)
    _check(hs, oracle, key, data, 4096)
