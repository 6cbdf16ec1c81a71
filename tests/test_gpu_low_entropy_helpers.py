"""The split-phase helpers of the low-entropy codec behind the reference's names (src/rle.h:67-96; include/hsrle.h): statistics, header writer /
reader and the two stream bodies as separate calls with the tables in host structs of the reference's layout.
Bar: (1) the identity the reference itself is built from (rle8_low_entropy_cpu.c:11-50) -- [u32 size][u32 inSize] + write_compress_info(get_compress_info*)
+ compress_with_info == the oracle's (= the reference's) whole stream, byte for byte, for the regular and the only_max_frequency tables;
(2) read_decompress_info + decompress_with_info give the input back; (3) where the compiled reference travels with the snapshot
(oracle/_ref/libhsrle_ref.so), every struct and every body equals what the reference's own function returns for the same arguments, the Short
bodies and tables that did NOT come from the statistics pass included."""
import ctypes
import random
import struct

import pytest

from hsrle_testlib import mixed_runs, single_symbol_mix

pytestmark = pytest.mark.gpu


class CompressInfo(ctypes.Structure):
    _fields_ = [("rle", ctypes.c_uint8 * 256), ("symbolsByProb", ctypes.c_uint8 * 256), ("symbolCount", ctypes.c_uint8)]


class DecompressInfo(ctypes.Structure):
    _fields_ = [("rle", ctypes.c_uint8 * 256), ("symbolToCount", ctypes.c_uint8 * 256)]


@pytest.fixture(scope="module")
def hs():
    import torch

    assert torch.cuda.is_available(), "these tests need a GPU"
    import hsrle

    hsrle.lib()
    return hsrle


def _bind(lib):
    u8p, u32 = ctypes.c_char_p, ctypes.c_uint32
    for name in ("rle8_low_entropy_get_compress_info", "rle8_low_entropy_get_compress_info_only_max_frequency"):
        f = getattr(lib, name)
        f.restype = ctypes.c_bool
        f.argtypes = [u8p, u32, ctypes.POINTER(CompressInfo)]
    lib.rle8_low_entropy_write_compress_info.restype = u32
    lib.rle8_low_entropy_write_compress_info.argtypes = [ctypes.POINTER(CompressInfo), u8p, u32]
    lib.rle8_low_entropy_read_decompress_info.restype = u32
    lib.rle8_low_entropy_read_decompress_info.argtypes = [u8p, u32, ctypes.POINTER(DecompressInfo)]
    for name in ("rle8_low_entropy_compress_with_info", "rle8_low_entropy_short_compress_with_info"):
        f = getattr(lib, name)
        f.restype = u32
        f.argtypes = [u8p, u32, ctypes.POINTER(CompressInfo), u8p, u32]
    for name in ("rle8_low_entropy_decompress_with_info", "rle8_low_entropy_short_decompress_with_info"):
        f = getattr(lib, name)
        f.restype = u32
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(DecompressInfo), u8p, u32]
    return lib


def _body(lib, name, data, info, slack=600):
    out = ctypes.create_string_buffer(2 * len(data) + slack)
    size = getattr(lib, name)(data, len(data), ctypes.byref(info), out, 2 * len(data) + slack - 64)
    return out.raw[:size] if size else None


def _decode(lib, name, body, dinfo, n):
    buf = ctypes.create_string_buffer(body, len(body) + 512)          # (the reference's vector loops read behind pEnd - 256 ... pEnd)
    out = ctypes.create_string_buffer(n + 512)
    base = ctypes.addressof(buf)
    got = getattr(lib, name)(base, base + len(body), ctypes.byref(dinfo), out, n)
    return out.raw[:n] if got == n else None


def _inputs():
    rng = random.Random(3301)
    inputs = [mixed_runs(rng, 200000, alphabet=3), single_symbol_mix(rng, 9000), bytes([5]) * 100000 + mixed_runs(rng, 3000), mixed_runs(rng, 333),
              bytes(range(256)) * 40 + b"\x00" * 5000, b"\x00" * 70000, b"\x07", b"ab" * 2000 + b"a" * 4000, bytes(rng.randrange(256) for _ in range(5000)),
              mixed_runs(rng, 1 << 20, alphabet=4), b"\x00\x00\x01\x01" * 100000, (b"\x07" * 16383 + b"\x09" + b"\x07" * 33 + b"ab" * 100) * 9]
    for _ in range(40):
        length = rng.choice([1, 2, 3, 17, 33, 64, 255, 256, 257, 258, 300, 513, 700, 3000])
        alphabet = [rng.randrange(256) for _ in range(rng.choice([1, 2, 3, 5, 17]))]
        d = bytearray()
        while len(d) < length:
            d += bytes([rng.choice(alphabet)]) * rng.choice([1, 1, 2, 3, 7, 31, 32, 33, 64, 254, 255, 256, 600])
        inputs.append(bytes(d[:length]))
    return inputs


def test_the_whole_stream_is_its_helpers_in_a_row(hs, oracle):
    lib = _bind(hs.lib())
    n = 0
    for data in _inputs():
        for only_max, getter in ((0, "rle8_low_entropy_get_compress_info"), (1, "rle8_low_entropy_get_compress_info_only_max_frequency")):
            want = oracle.low_entropy_compress(only_max << 1, data)
            if want is None:
                continue
            info = CompressInfo()
            assert getattr(lib, getter)(data, len(data), ctypes.byref(info)), f"{getter} on {len(data)} bytes"
            head = ctypes.create_string_buffer(600)
            hsize = lib.rle8_low_entropy_write_compress_info(ctypes.byref(info), head, 600)
            assert hsize == 33 + (info.symbolCount or 255)
            body = _body(lib, "rle8_low_entropy_compress_with_info", data, info)
            assert body is not None, f"compress_with_info on {len(data)} bytes"
            stream = struct.pack("<II", 8 + hsize + len(body), len(data)) + head.raw[:hsize] + body
            assert stream == want, f"{len(data)} bytes, only_max {only_max}: the helpers in a row are not the reference's stream"
            dinfo = DecompressInfo()
            assert lib.rle8_low_entropy_read_decompress_info(want[8:], len(want) - 8, ctypes.byref(dinfo)) == hsize
            assert sorted(dinfo.symbolToCount) == list(range(256))
            assert [dinfo.symbolToCount[s] for s in info.symbolsByProb] == list(range(256))       # count c belongs to symbolsByProb[c], the unlisted ones ascending behind
            assert _decode(lib, "rle8_low_entropy_decompress_with_info", body, dinfo, len(data)) == data
            n += 1
    assert n >= 90
    # argument errors: NULL / empty / short buffers
    info = CompressInfo()
    assert not lib.rle8_low_entropy_get_compress_info(b"", 0, ctypes.byref(info))
    assert lib.rle8_low_entropy_write_compress_info(ctypes.byref(info), ctypes.create_string_buffer(100), 100) == 0
    assert lib.rle8_low_entropy_compress_with_info(b"abcabc", 6, ctypes.byref(info), ctypes.create_string_buffer(16), 5) == 0
    bad = DecompressInfo()                                             # all counts 0: not a permutation
    assert _decode(lib, "rle8_low_entropy_decompress_with_info", b"abc", bad, 3) is None


def test_helpers_against_the_compiled_reference(hs, reference):
    """Same arguments into oracle/_ref/libhsrle_ref.so (the reference's own code): structs and bodies must be identical -- also for the Short bodies and for
    tables nobody's statistics produced (every symbol flagged, odd code orders)."""
    lib, ref = _bind(hs.lib()), _bind(reference.lib)
    rng = random.Random(77)
    n = 0
    for data in _inputs():
        for getter in ("rle8_low_entropy_get_compress_info", "rle8_low_entropy_get_compress_info_only_max_frequency"):
            a, b = CompressInfo(), CompressInfo()
            assert getattr(lib, getter)(data, len(data), ctypes.byref(a)) and getattr(ref, getter)(data, len(data), ctypes.byref(b))
            assert bytes(a) == bytes(b), f"{getter} on {len(data)} bytes"
            ha, hb = ctypes.create_string_buffer(600), ctypes.create_string_buffer(600)
            assert lib.rle8_low_entropy_write_compress_info(ctypes.byref(a), ha, 600) == ref.rle8_low_entropy_write_compress_info(ctypes.byref(b), hb, 600)
            assert ha.raw == hb.raw
            da, db = DecompressInfo(), DecompressInfo()
            assert lib.rle8_low_entropy_read_decompress_info(ha.raw, 600, ctypes.byref(da)) == ref.rle8_low_entropy_read_decompress_info(hb.raw, 600, ctypes.byref(db))
            assert bytes(da) == bytes(db)
            tables = [a]
            if len(data) >= 300:
                # tables of the caller's own: a few more flags, the codes in another order (still a permutation)
                odd = CompressInfo.from_buffer_copy(bytes(a))
                for s in rng.sample(range(256), 5):
                    odd.rle[s] = 1
                perm = list(odd.symbolsByProb)
                listed = odd.symbolCount or 255                      # (only the listed part travels in the header: the reader puts the rest in ascending order)
                if listed >= 2:
                    i, j = rng.sample(range(listed), 2)
                    perm[i], perm[j] = perm[j], perm[i]
                for k in range(256):
                    odd.symbolsByProb[k] = perm[k]
                tables.append(odd)
            for info in tables:
                for enc, dec in (("rle8_low_entropy_compress_with_info", "rle8_low_entropy_decompress_with_info"),
                                 ("rle8_low_entropy_short_compress_with_info", "rle8_low_entropy_short_decompress_with_info")):
                    mine, theirs = _body(lib, enc, data, info), _body(ref, enc, data, info)
                    assert mine == theirs, f"{enc} on {len(data)} bytes"
                    head = ctypes.create_string_buffer(600)
                    hsize = ref.rle8_low_entropy_write_compress_info(ctypes.byref(info), head, 600)
                    dinfo = DecompressInfo()
                    assert ref.rle8_low_entropy_read_decompress_info(head.raw, hsize, ctypes.byref(dinfo)) == hsize
                    assert _decode(lib, dec, theirs, dinfo, len(data)) == data, f"{dec} on {len(data)} bytes"
                    n += 1
    assert n >= 200
