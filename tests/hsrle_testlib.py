"""Shared helpers for the test-suite: ctypes access to the oracle (oracle/libhsrle_oracle.so), to the
compiled reference when it exists (oracle/_ref/libhsrle_ref.so), the codec table, and input generators.

The codec table mirrors the reference's plugin table for the hot path
(reference: src/codec_funcs.h:270-410, src/rle.h:100-394): one (compress, decompress) name pair per codec.
The fuzz grammar mirrors the reference's structured fuzzer (reference: src/rle_fuzz.c:13-44, :159-438):
alternating sections of random bytes and repeated symbols of 1..16 bytes, section lengths straddling the
1/2/4-byte length-field boundaries.
"""
import ctypes
import os
import random
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libhsrle_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libhsrle_ref.so")

PLAIN, PACKED, LUT3, LUT7, SINGLE, PACKED_SINGLE, SHORT0, SHORT1, SHORT3, SHORT7, GREEDY1, GREEDY3, GREEDY7, SINGLE_SHORT = range(14)
FAMILY_NAMES = {PLAIN: "plain", PACKED: "packed", LUT3: "3symlut", LUT7: "7symlut", SINGLE: "single", PACKED_SINGLE: "packed_single",
                SHORT0: "short", SHORT1: "1symlut_short", SHORT3: "3symlut_short", SHORT7: "7symlut_short"}


class Codec:
    def __init__(self, key, family, S, aligned, cname, dname):
        self.key, self.family, self.S, self.aligned, self.cname, self.dname = key, family, S, aligned, cname, dname

    def __repr__(self):
        return self.key


def _codec_table():
    t = [
        Codec("rle8_multi", PLAIN, 1, 0, "rle8_multi_compress", "rle8_decompress"),
        Codec("rle8_packed_multi", PACKED, 1, 0, "rle8_packed_multi_compress", "rle8_packed_decompress"),
        Codec("rle8_3symlut", LUT3, 1, 0, "rle8_3symlut_compress", "rle8_3symlut_decompress"),
        Codec("rle8_7symlut", LUT7, 1, 0, "rle8_7symlut_compress", "rle8_7symlut_decompress"),
        Codec("rle8_single", SINGLE, 1, 0, "rle8_single_compress", "rle8_decompress"),
        Codec("rle8_packed_single", PACKED_SINGLE, 1, 0, "rle8_packed_single_compress", "rle8_packed_decompress"),
    ]
    for W, S in ((16, 2), (24, 3), (32, 4), (48, 6), (64, 8), (128, 16)):
        for al, nm in ((1, "sym"), (0, "byte")):
            t.append(Codec(f"rle{W}_{nm}", PLAIN, S, al, f"rle{W}_{nm}_compress", f"rle{W}_{nm}_decompress"))
            t.append(Codec(f"rle{W}_{nm}_packed", PACKED, S, al, f"rle{W}_{nm}_packed_compress", f"rle{W}_{nm}_packed_decompress"))
            if S != 16:
                t.append(Codec(f"rle{W}_3symlut_{nm}", LUT3, S, al, f"rle{W}_3symlut_{nm}_compress", f"rle{W}_3symlut_{nm}_decompress"))
                t.append(Codec(f"rle{W}_7symlut_{nm}", LUT7, S, al, f"rle{W}_7symlut_{nm}_compress", f"rle{W}_7symlut_{nm}_decompress"))
    # Short family (reference: src/rle.h:202-348, src/codec_funcs.h:283-388): ids 50..93 in this order
    t.append(Codec("rle8_multi_short", SHORT0, 1, 0, "rle8_multi_short_compress", "rle8_multi_short_decompress"))
    for fam, k in ((SHORT1, 1), (SHORT3, 3), (SHORT7, 7)):
        t.append(Codec(f"rle8_{k}symlut_short", fam, 1, 0, f"rle8_{k}symlut_short_compress", f"rle8_{k}symlut_short_decompress"))
    for W, S in ((16, 2), (24, 3), (32, 4), (48, 6), (64, 8)):
        for al, nm in ((1, "sym"), (0, "byte")):
            t.append(Codec(f"rle{W}_{nm}_short", SHORT0, S, al, f"rle{W}_{nm}_short_compress", f"rle{W}_{nm}_short_decompress"))
            for fam, k in ((SHORT1, 1), (SHORT3, 3), (SHORT7, 7)):
                t.append(Codec(f"rle{W}_{k}symlut_{nm}_short", fam, S, al, f"rle{W}_{k}symlut_{nm}_short_compress", f"rle{W}_{k}symlut_{nm}_short_decompress"))
    return t


CODECS = _codec_table()
CODEC_BY_KEY = {c.key: c for c in CODECS}
# Greedy encoders (reference: src/rle.h:398-416, src/codec_funcs.h:298-388): paired with the Short decoders of the same grammar
GREEDY_CODECS = [Codec(f"rle{W}_{k}symlut_byte_short_greedy", fam, S, 0, f"rle{W}_{k}symlut_byte_short_compress_greedy", f"rle{W}_{k}symlut_byte_short_decompress")
                 for W, S in ((16, 2), (24, 3), (32, 4), (48, 6), (64, 8)) for fam, k in ((GREEDY1, 1), (GREEDY3, 3), (GREEDY7, 7))]
EXTREME_CODECS = CODECS[:50]     # the north-star matrix
SHORT_CODECS = CODECS[50:]
assert len(CODECS) == 94
CODECS = CODECS + GREEDY_CODECS  # library codec ids 94..108 in this order
CODECS.append(Codec("rle8_single_short", SINGLE_SHORT, 1, 0, "rle8_single_short_compress", "rle8_single_short_decompress"))   # id 109
CODEC_BY_KEY = {c.key: c for c in CODECS}


def build_oracle():
    """(Re)build the oracle library (and the reference library when /root/reference is present)."""
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)


_u8p = ctypes.POINTER(ctypes.c_uint8)


def _as_u8p(buf):
    return ctypes.cast(buf, _u8p)


class Oracle:
    """ctypes wrapper of oracle/libhsrle_oracle.so."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        self.lib = ctypes.CDLL(ORACLE_SO)
        L = self.lib
        L.hso_compress_bounds.restype = ctypes.c_uint32
        L.hso_compress_bounds.argtypes = [ctypes.c_uint32]
        L.hso_compress.restype = ctypes.c_uint32
        L.hso_compress.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
        L.hso_decompress.restype = ctypes.c_uint32
        L.hso_decompress.argtypes = L.hso_compress.argtypes
        L.hso_call.restype = ctypes.c_uint32
        L.hso_call.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]

        L.hso_synth.restype = ctypes.c_int
        L.hso_synth.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
        L.hso_compress_blocks.restype = ctypes.c_uint32
        L.hso_compress_blocks.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]

    def bounds(self, n):
        return self.lib.hso_compress_bounds(n)

    def synth(self, kind, symbol_bytes, seed, size, offset=0):
        """Bytes [offset, offset + size) of a synthetic workload (offset multiple of 64 KiB); numpy uint8 array."""
        out = np.empty(size, dtype=np.uint8)
        ok = self.lib.hso_synth(kind, symbol_bytes, seed, offset, out.ctypes.data, size)
        assert ok
        return out

    def compress_blocks(self, codec, data, block_size):
        """Encode every block of `data` (numpy uint8) independently; returns the list of block streams."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        n = data.size
        nb = (n + block_size - 1) // block_size
        stride = (self.bounds(block_size) + 15) & ~15
        out = np.empty(nb * stride, dtype=np.uint8)
        sizes = np.empty(nb, dtype=np.uint32)
        got = self.lib.hso_compress_blocks(codec.family, codec.S, codec.aligned, data.ctypes.data, n, block_size, out.ctypes.data, stride, sizes.ctypes.data)
        assert got == nb
        return [out[i * stride : i * stride + int(sizes[i])].tobytes() for i in range(nb)]

    def rle8m_compress(self, sections, data):
        """rle8m stream of `data` with `sections` sub-sections (reference: rle8m_compress), or None where the reference fails."""
        L = self.lib
        L.hso_rle8m_compress_bounds.restype = ctypes.c_uint32
        L.hso_rle8m_compress.restype = ctypes.c_uint32
        data = bytes(data)
        cap = L.hso_rle8m_compress_bounds(ctypes.c_uint32(sections), ctypes.c_uint32(len(data)))
        # the reference checks the room left against a section's INPUT size; its stream can be twice as long, and then the reference
        # writes behind `cap` (restated faithfully).  Room for that, and such a result counts as a failure (self.rle8m_overflowed).
        out = ctypes.create_string_buffer(cap + 2 * len(data) + 64)
        size = L.hso_rle8m_compress(ctypes.c_uint32(sections), data, ctypes.c_uint32(len(data)), out, ctypes.c_uint32(cap))
        self.rle8m_overflowed = size > cap
        return out.raw[:size] if 0 < size <= cap else None

    def rle8m_decompress(self, stream, out_size):
        L = self.lib
        L.hso_rle8m_decompress.restype = ctypes.c_uint32
        out = ctypes.create_string_buffer(out_size + 64)
        got = L.hso_rle8m_decompress(bytes(stream), ctypes.c_uint32(len(stream)), out, ctypes.c_uint32(out_size))
        return out.raw[:out_size] if got == out_size else None

    def low_entropy_compress(self, variant, data):
        """Unsectioned low-entropy stream (variant: bit 0 Short form, bit 1 only_max_frequency), or None where the stream outgrows the
        bound (the reference then writes behind its caller's buffer: self.le_overflowed)."""
        L = self.lib
        L.hso_low_entropy_compress_bounds.restype = ctypes.c_uint32
        L.hso_low_entropy_compress.restype = ctypes.c_uint32
        data = bytes(data)
        cap = L.hso_low_entropy_compress_bounds(ctypes.c_uint32(len(data)))
        out = ctypes.create_string_buffer(cap + 2 * len(data) + 64)
        size = L.hso_low_entropy_compress(ctypes.c_int(variant), data, ctypes.c_uint32(len(data)), out, ctypes.c_uint32(cap))
        self.le_overflowed = size > cap
        return out.raw[:size] if 0 < size <= cap else None

    def low_entropy_decompress(self, stream, out_size):
        L = self.lib
        L.hso_low_entropy_decompress.restype = ctypes.c_uint32
        out = ctypes.create_string_buffer(out_size + 64)
        got = L.hso_low_entropy_decompress(bytes(stream), ctypes.c_uint32(len(stream)), out, ctypes.c_uint32(out_size))
        return out.raw[:out_size] if got == out_size else None

    def compress(self, codec, data):
        data = bytes(data)
        n = len(data)
        cap = self.bounds(n)
        out = ctypes.create_string_buffer(cap + 64)
        size = self.lib.hso_compress(codec.family, codec.S, codec.aligned, data, n, out, cap)
        return out.raw[:size] if size else None

    def decompress(self, codec, stream, out_size=None, slack=0):
        stream = bytes(stream)
        if out_size is None:
            out_size = int.from_bytes(stream[:4], "little")
        out = ctypes.create_string_buffer(max(out_size + slack, 1))
        size = self.lib.hso_decompress(codec.family, codec.S, codec.aligned, stream, len(stream), out, out_size)
        return out.raw[:size] if size else None

    def call(self, name, data, out_cap):
        data = bytes(data)
        out = ctypes.create_string_buffer(max(out_cap, 1))
        size = self.lib.hso_call(name.encode(), data, len(data), out, out_cap)
        return size, out.raw[: (size if size != 0xFFFFFFFF else 0)]


def guard_pad(data, pad=64):
    """SURVEY.md §8c guard-padding rule: the reference encoders for widths > 8 bit read up to 2*S-1 bytes past
    inSize; fill the pad so that pad[k] differs from buf[k-d] for d in {1,2,3,4,6,8,16} -- then "bytes beyond the end
    never match", the semantics this repository adopts."""
    buf = bytearray(data) + bytearray(pad)
    n = len(data)
    for k in range(n, n + pad):
        v = 0
        while any(k - d >= 0 and buf[k - d] == v for d in (1, 2, 3, 4, 6, 8, 16)):
            v += 1
        buf[k] = v
    return buf


class Reference:
    """ctypes wrapper of the compiled reference (oracle/_ref/libhsrle_ref.so).  Only available where the library was
    built, i.e. in a container that has /root/reference (or a snapshot that carries the prebuilt .so)."""

    def __init__(self):
        self.lib = ctypes.CDLL(REF_SO)
        self.lib.rle_compress_bounds.restype = ctypes.c_uint32
        self.lib.rle_compress_bounds.argtypes = [ctypes.c_uint32]
        self.lib.hsrle_ref_set_max_simd.argtypes = [ctypes.c_int]
        self.lib.hsrle_ref_has_avx2.restype = ctypes.c_int
        self.has_avx2 = bool(self.lib.hsrle_ref_has_avx2())

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def set_max_simd(self, level):
        self.lib.hsrle_ref_set_max_simd(level)

    def rle8m_compress(self, sections, data):
        L = self.lib
        L.rle8m_compress_bounds.restype = ctypes.c_uint32
        L.rle8m_compress.restype = ctypes.c_uint32
        data = bytes(data)
        cap = L.rle8m_compress_bounds(ctypes.c_uint32(sections), ctypes.c_uint32(len(data)))
        out = ctypes.create_string_buffer(cap + 2 * len(data) + 64)   # see Oracle.rle8m_compress
        size = L.rle8m_compress(ctypes.c_uint32(sections), data, ctypes.c_uint32(len(data)), out, ctypes.c_uint32(cap))
        self.rle8m_overflowed = size > cap
        return out.raw[:size] if 0 < size <= cap else None

    def rle8m_decompress(self, stream, out_size):
        L = self.lib
        L.rle8m_decompress.restype = ctypes.c_uint32
        out = ctypes.create_string_buffer(out_size + 256)
        got = L.rle8m_decompress(bytes(stream), ctypes.c_uint32(len(stream)), out, ctypes.c_uint32(out_size))
        return out.raw[:out_size] if got == out_size else None

    LE_NAMES = ("rle8_low_entropy_compress", "rle8_low_entropy_short_compress", "rle8_low_entropy_compress_only_max_frequency", "rle8_low_entropy_short_compress_only_max_frequency")

    def low_entropy_compress(self, variant, data):
        """variant: bit 0 Short form, bit 1 only_max_frequency (LE_NAMES[variant])"""
        L = self.lib
        L.rle8_low_entropy_compress_bounds.restype = ctypes.c_uint32
        f = getattr(L, self.LE_NAMES[variant])
        f.restype = ctypes.c_uint32
        data = bytes(data)
        cap = L.rle8_low_entropy_compress_bounds(ctypes.c_uint32(len(data)))
        out = ctypes.create_string_buffer(cap + 2 * len(data) + 64)   # see Oracle.low_entropy_compress
        size = f(data, ctypes.c_uint32(len(data)), out, ctypes.c_uint32(cap))
        self.le_overflowed = size > cap
        return out.raw[:size] if 0 < size <= cap else None

    def low_entropy_decompress(self, short, stream, out_size):
        f = self.lib.rle8_low_entropy_short_decompress if short else self.lib.rle8_low_entropy_decompress
        f.restype = ctypes.c_uint32
        out = ctypes.create_string_buffer(out_size + 512)            # (the reference's vector loops write whole vectors behind the end)
        src = bytes(stream) + bytes(512)
        got = f(src, ctypes.c_uint32(len(stream)), out, ctypes.c_uint32(out_size))
        return out.raw[:out_size] if got == out_size else None

    def _fn(self, name):
        f = getattr(self.lib, name)
        f.restype = ctypes.c_uint32
        f.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
        return f

    def compress(self, codec, data, guard=True):
        n = len(data)
        buf = guard_pad(data) if guard else bytearray(data) + bytearray(64)
        inb = (ctypes.c_uint8 * len(buf)).from_buffer(buf)
        cap = self.lib.rle_compress_bounds(n)
        out = ctypes.create_string_buffer(cap + 64)
        size = self._fn(codec.cname)(inb, n, out, cap)
        return out.raw[:size] if size else None

    def decompress(self, codec, stream, out_size=None):
        stream = bytes(stream)
        if out_size is None:
            out_size = int.from_bytes(stream[:4], "little")
        # the reference scribbles up to rle_decompress_additional_size() (128) bytes past the end, and reads past the
        # stream end with vector loads: give both buffers slack.
        src = ctypes.create_string_buffer(stream + b"\xAA" * 256, len(stream) + 256)
        out = ctypes.create_string_buffer(out_size + 256)
        size = self._fn(codec.dname)(src, len(stream), out, out_size)
        return out.raw[:size] if size else None


# ------------------------------------------------------------------------------------------------------------------
# input generators

# section-length classes of the reference fuzzer (src/rle_fuzz.c:30-44, :159-206)
FUZZ_LENGTHS = list(range(1, 281)) + [768, 816, 867, 921, 978, 1039, 1104, 1173, 2048, 4096, 8192] + list(range(65528, 65561))
SMALL_LENGTHS = [1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 16, 17, 31, 32, 33, 34, 63, 64, 65, 126, 127, 128, 129, 254, 255, 256, 257, 280]
SYMBOL_LENGTHS = [1, 2, 3, 4, 6, 8, 16]


def fuzz_sections(rng, max_sections=8, lengths=None, alphabet=256):
    """One input of the reference fuzzer's grammar: alternating random / repeating-symbol sections."""
    lengths = lengths or SMALL_LENGTHS
    out = bytearray()
    nsec = rng.randrange(1, max_sections + 1)
    rep = rng.random() < 0.5
    for _ in range(nsec):
        ln = rng.choice(lengths)
        if rep:
            S = rng.choice(SYMBOL_LENGTHS + list(range(1, 17)))
            sym = bytes(rng.randrange(alphabet) for _ in range(S))
            if rng.random() < 0.25:
                sym = bytes([rng.choice([0x00, 0x7F, 0xFF, 0x01, 0x7E, 0x80, 0xFE])]) * S
            k = ln if rng.random() < 0.5 else ln * S  # length in bytes or in symbols
            b = (sym * (k // S + 2))[:k] if rng.random() < 0.5 else sym * max(1, k // S)
            out += b
        else:
            out += bytes(rng.randrange(alphabet) for _ in range(ln))
        rep = not rep
    return bytes(out)


def mixed_runs(rng, n, alphabet=None):
    """Dense mix of literal gaps and runs of every symbol width (stresses emit decisions and the LUT)."""
    out = bytearray()
    alphabet = alphabet or rng.choice([1, 2, 3, 4, 256])
    while len(out) < n:
        out += bytes(rng.randrange(alphabet) for _ in range(rng.choice([0, 1, 2, 3, 5, 9, 17, 40, 130, 260])))
        S = rng.choice([1, 1, 2, 3, 4, 6, 8, 16])
        sym = bytes(rng.randrange(alphabet) for _ in range(S))
        if rng.random() < 0.3:
            sym = bytes([rng.choice([0, 0x7F, 0xFF, 1])]) * S
        k = rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 20, 33, 40, 70, 130, 300])
        b = sym * k
        if rng.random() < 0.5 and S > 1:
            b = b[: len(b) - rng.randrange(S)]
        out += b
    return bytes(out[:n])


def single_symbol_mix(rng, n):
    """Inputs dominated by one favourite byte value (what the Single codecs are for)."""
    out = bytearray()
    alphabet = rng.choice([2, 3, 4, 256])
    fav = rng.randrange(alphabet)
    while len(out) < n:
        out += bytes(rng.randrange(alphabet) for _ in range(rng.choice([0, 1, 2, 3, 5, 9, 17, 40, 130, 260, 300])))
        s = fav if rng.random() < 0.7 else rng.randrange(alphabet)
        out += bytes([s]) * rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 15, 16, 17, 20, 33, 40, 70, 130, 260, 300])
    return bytes(out[:n])


# ------------------------------------------------------------------------------------------------------------------
# deterministic synthetic workloads (BASELINE configs).  Integer-only; the same generators exist in C
# (oracle/hsrle_synth.c, test infrastructure) and on the device (hsrle_synth_dev in the product library).

MASK64 = (1 << 64) - 1


def splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & MASK64
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return state, z ^ (z >> 31)


SYNTH_CHUNK = 65536
SYNTH_RUNS, SYNTH_VIDEO = 0, 1


def synth_chunk_py(kind, S, seed, chunk_index, size=SYNTH_CHUNK):
    """Pure-python statement of one generation chunk (slow; used to pin the C generator on small cases)."""
    st = (seed * 0x9E3779B97F4A7C15 + chunk_index * 0xD1B54A32D192ED03 + kind) & MASK64
    out = bytearray()
    if kind == SYNTH_RUNS:
        while len(out) < size:
            st, r = splitmix64(st)
            L = 1 + r % 63
            lit = bytearray()
            while len(lit) < L:
                st, v = splitmix64(st)
                lit += v.to_bytes(8, "little")
            out += lit[:L]
            st, r = splitmix64(st)
            R = 2 + r % 62
            sym = bytearray()
            while len(sym) < S:
                st, v = splitmix64(st)
                sym += v.to_bytes(8, "little")
            out += bytes(sym[:S]) * R
    else:
        vals = bytes([0x01, 0x02, 0x03, 0xFF, 0xFE, 0x04])
        while len(out) < size:
            st, r = splitmix64(st)
            Z = 40 + r % 120 if ((r >> 32) & 3) == 0 else 10 + r % 16
            out += bytes(Z)
            st, r = splitmix64(st)
            B = 1 + r % 9
            for k in range(B):
                out.append(vals[(r >> (8 + 4 * k)) % 6])
    return bytes(out[:size])


# ------------------------------------------------------------------------------------------------------------------
# host-side statement of the block container (include/hsrle.h), used by the CPU tests of the sharding logic


def build_container(codec_index, uncompressed_size, block_size, streams):
    """Assemble a container exactly as the device does (header, u64 offset table, payload, 32 zero bytes)."""
    import struct

    nb = len(streams)
    payload = b"".join(streams)
    offs = [0]
    for s in streams:
        offs.append(offs[-1] + len(s))
    total = 64 + 8 * (nb + 1) + len(payload) + 32
    head = b"HSRLEKIT" + struct.pack("<IIQIIQQ", 1, codec_index, uncompressed_size, block_size, nb, len(payload), total) + bytes(16)
    return head + struct.pack(f"<{nb + 1}Q", *offs) + payload + bytes(32)


# ------------------------------------------------------------------------------------------------------------------
# big-config manifests (tests/golden/big/, minted from the compiled reference by tests/golden/make_big_manifest.py)

BIG_DIR = os.path.join(REPO, "tests", "golden", "big")


def big_manifest():
    import json

    path = os.path.join(BIG_DIR, "manifest.json")
    return json.load(open(path)) if os.path.exists(path) else None


def big_case(codec_key, kind, seed, size, block):
    """(name, entry, roll-ups as uint64 array) of the manifest case with these generator parameters, or None."""
    man = big_manifest()
    if not man:
        return None
    for name, e in man["cases"].items():
        if (e["codec"], e["kind"], e["seed"], e["size"], e["block"]) == (codec_key, kind, seed, size, block):
            return name, e, np.fromfile(os.path.join(BIG_DIR, name + ".rollup.u64"), dtype="<u8")
    return None


def rollups(hashes, group=256):
    """Roll-up per `group` block hashes (oracle/hsrle_hash.h): r = 0; r = (rotl64(r, 7) ^ b) * 0x9E3779B97F4A7C15 -- vectorised over the groups."""
    h = np.ascontiguousarray(hashes).view(np.uint64)
    n = h.size
    groups = (n + group - 1) // group
    padded = np.zeros(groups * group, dtype=np.uint64)
    padded[:n] = h
    padded = padded.reshape(groups, group)
    counts = np.minimum(group, n - np.arange(groups) * group)
    r = np.zeros(groups, dtype=np.uint64)
    mul = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        for k in range(group):
            nxt = (((r << np.uint64(7)) | (r >> np.uint64(57))) ^ padded[:, k]) * mul
            r = np.where(k < counts, nxt, r)
    return r
