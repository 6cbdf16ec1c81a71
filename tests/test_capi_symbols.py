"""The C-ABI library loads (no GPU needed) and exports every symbol include/hsrle.h declares; no compute calls here."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "hypersonic-rle-kit_amd", "libhsrle_hip.so")
HEADER = os.path.join(REPO, "include", "hsrle.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"#ifdef HSRLE_EXPERIMENTS.*?#endif", "", text, flags=re.S)   # experiment builds only: not part of the shipped library
    names = set(re.findall(r"\b((?:hsrle|rle)[A-Za-z0-9_]*)\s*\(", re.sub(r"/\*.*?\*/", "", text, flags=re.S)))
    # macro-declared drop-in pairs
    pairs = ["rle8_3symlut", "rle8_7symlut", "rle128_sym", "rle128_sym_packed", "rle128_byte", "rle128_byte_packed"]
    for W in (16, 24, 32, 48, 64):
        pairs += [f"rle{W}_{v}" for v in ("sym", "sym_packed", "byte", "byte_packed", "3symlut_sym", "7symlut_sym", "3symlut_byte", "7symlut_byte")]
    pairs += ["rle8_multi_short", "rle8_single_short", "rle8_1symlut_short", "rle8_3symlut_short", "rle8_7symlut_short"]
    for W in (16, 24, 32, 48, 64):
        pairs += [f"rle{W}_{v}_short" for v in ("sym", "byte", "1symlut_sym", "1symlut_byte", "3symlut_sym", "3symlut_byte", "7symlut_sym", "7symlut_byte")]
    for W in (16, 24, 32, 48, 64):
        for k in (1, 3, 7):
            names.add(f"rle{W}_{k}symlut_byte_short_compress_greedy")
    for p in pairs:
        names.add(p + "_compress")
        names.add(p + "_decompress")
    return sorted(n for n in names if not n.endswith("_t"))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import subprocess

        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(REPO, "hypersonic-rle-kit_amd")])
    return ctypes.CDLL(LIB)


def test_all_declared_symbols_are_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 2 + 100 + 88 + 15 + 15  # helpers + 50 extreme + 44 Short drop-in pairs + 15 Greedy encoders + hsrle_* API
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"not exported: {missing}"


def test_reference_names_present(lib):
    """Exactly the names of the reference's rle.h for the hot path (src/rle.h:100-394)."""
    for s in ("rle_compress_bounds", "rle_decompress_additional_size", "rle8_multi_compress", "rle8_single_compress", "rle8_decompress",
              "rle8_packed_multi_compress", "rle8_packed_single_compress", "rle8_packed_decompress", "rle64_3symlut_byte_compress",
              "rle64_3symlut_byte_decompress", "rle24_sym_packed_compress", "rle128_byte_packed_decompress",
              "rle8_multi_short_compress", "rle8_1symlut_short_decompress", "rle16_sym_short_compress", "rle48_7symlut_byte_short_decompress",
              "rle8m_opencl_init", "rle8m_opencl_destroy", "rle8m_opencl_decompress", "rle8m_decompress", "rle8m_compress", "rle8m_compress_bounds"):
        assert hasattr(lib, s)


def test_pure_host_helpers(lib):
    """Helpers that need no device."""
    lib.rle_compress_bounds.restype = ctypes.c_uint32
    assert lib.rle_compress_bounds(ctypes.c_uint32(1000)) == 1193
    assert lib.rle_compress_bounds(ctypes.c_uint32((1 << 30) + 1)) == 0
    lib.rle_decompress_additional_size.restype = ctypes.c_uint32
    assert lib.rle_decompress_additional_size() == 128
    lib.hsrle_codec_from_name.restype = ctypes.c_int
    lib.hsrle_codec_name.restype = ctypes.c_char_p
    for i in range(110):
        name = lib.hsrle_codec_name(i)
        assert lib.hsrle_codec_from_name(name) == i
    assert lib.hsrle_codec_from_name(b"rle8_packed_multi") == 1 and lib.hsrle_codec_from_name(b"rle64_3symlut_byte") == 44
    assert lib.hsrle_codec_from_name(b"nope") == -1
    lib.hsrle_container_bound.restype = ctypes.c_uint64
    lib.hsrle_container_bound.argtypes = [ctypes.c_uint64, ctypes.c_uint32]
    assert lib.hsrle_container_bound(1 << 20, 4096) > (1 << 20)
    assert lib.hsrle_container_bound(1 << 20, 4000) == 0  # block size must be a multiple of 128


def test_encode_path_selection(lib):
    """hsrle_encode_path (no device): which encoder a container takes.  Big containers: one lane per block; small ones of 1 .. 4 KiB blocks: the
    run list encoders for the multi-symbol 8 bit codecs, the 2 .. 8 byte codecs and their Short family, the split encode for the rest that
    can be cut (Single, 128 bit and Greedy only with a workspace sized for the codec); other block sizes: split encode or ring."""
    lib.hsrle_encode_path.restype = ctypes.c_int
    lib.hsrle_encode_path.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32]
    lib.hsrle_codec_from_name.restype = ctypes.c_int
    cid = lambda n: lib.hsrle_codec_from_name(n.encode())
    RING, SPLIT, RUN_LIST, PP = 0, 1, 2, 3
    frame = 88473600
    # rounds 5 / 6: the position-parallel encoders take every container of <= 4 KiB blocks of their 93 codecs (all 50 extreme codecs among them), whatever its size
    for name in ("rle8_multi", "rle8_packed_multi", "rle16_sym", "rle16_byte_packed", "rle24_sym_packed", "rle32_byte_packed", "rle48_3symlut_sym", "rle64_3symlut_byte", "rle24_3symlut_byte",
                 "rle16_sym_short", "rle16_1symlut_sym_short", "rle24_byte_short", "rle48_1symlut_byte_short", "rle64_sym_short",
                 "rle8_multi_short", "rle8_1symlut_short", "rle48_3symlut_sym_short", "rle64_3symlut_byte_short",
                 "rle8_3symlut", "rle8_7symlut", "rle16_3symlut_sym", "rle16_3symlut_byte", "rle16_7symlut_byte", "rle24_7symlut_byte", "rle64_7symlut_sym",
                 "rle16_3symlut_sym_short", "rle32_3symlut_byte_short", "rle64_7symlut_byte_short", "rle24_7symlut_sym_short",
                 "rle8_single", "rle8_packed_single", "rle8_single_short", "rle128_sym", "rle128_sym_packed", "rle128_byte", "rle128_byte_packed"):
        assert cid(name) >= 0
        for size, block in ((frame, 4096), (frame, 1024), (frame, 512), (8 << 30, 4096), (4096, 128)):
            assert lib.hsrle_encode_path(cid(name), size, block) == PP, (name, size, block)
        # round 6: blocks above 4 KiB walked in 4 KiB windows (csrc/hsrle_encode8pw.hip.h, hsrle_encodeSpw.hip.h, hsrle_encodeLpw.hip.h): the two 8 bit multi-symbol codecs, the 54
        # codecs of hsrle_encodeSp.hip.h and the 30 of hsrle_encodeLp.hip.h -- plain / Packed with blocks of any size, the LUT / Short forms below 1 MiB per block (their field
        # widths and penalties go by value).  Not windowed: the 8 bit Single codecs and the 128 bit codecs
        windowed_any = name in ("rle8_multi", "rle8_packed_multi", "rle16_sym", "rle16_byte_packed", "rle24_sym_packed", "rle32_byte_packed")
        windowed_list = name not in ("rle8_single", "rle8_packed_single", "rle8_single_short", "rle128_sym", "rle128_sym_packed", "rle128_byte", "rle128_byte_packed") and not windowed_any
        if windowed_any or windowed_list:
            for size, block in ((frame, 8192), (8 << 30, 8192), (8 << 30, 65536), (1 << 20, 1 << 19), (frame, 4224)):
                assert lib.hsrle_encode_path(cid(name), size, block) == PP, (name, size, block)
            assert (lib.hsrle_encode_path(cid(name), 8 << 30, 1 << 20) == PP) == windowed_any, name
            continue
        assert lib.hsrle_encode_path(cid(name), frame, 8192) == SPLIT, name         # a wave holds 4 KiB: larger blocks by chunks
        assert lib.hsrle_encode_path(cid(name), 8 << 30, 8192) == RING, name
    for name in ("rle8_3symlut_short", "rle8_7symlut_short"):      # (the two Short codecs with a list that are not position-parallel: every pair of equal bytes would be a candidate)
        assert cid(name) >= 0
        assert lib.hsrle_encode_path(cid(name), frame, 4096) == RUN_LIST, name
        assert lib.hsrle_encode_path(cid(name), frame, 1024) == RUN_LIST, name
        assert lib.hsrle_encode_path(cid(name), 8 << 30, 4096) == RING, name        # 2 097 152 blocks
        assert lib.hsrle_encode_path(cid(name), frame, 8192) == SPLIT, name         # run list: blocks of 1 .. 4 KiB only
        assert lib.hsrle_encode_path(cid(name), frame, 512) == RING, name
    for name in ("rle16_1symlut_byte_short_greedy", "rle16_3symlut_byte_short_greedy"):
        assert lib.hsrle_encode_path(cid(name), frame, 4096) == RING, name          # (split only with a workspace sized for the codec -- the host cannot know)
    for name in ("rle16_1symlut_byte_short_greedy", "rle64_1symlut_byte_short_greedy", "rle8_single_short"):
        assert lib.hsrle_encode_path(cid(name), frame, 8192) == SPLIT, name         # round 4: their chunk encoders take the blocks of a container too
    lib.hsrle_compress_workspace_size.restype = ctypes.c_uint64
    lib.hsrle_compress_workspace_size.argtypes = [ctypes.c_uint64, ctypes.c_uint32]
    lib.hsrle_compress_workspace_size_codec.restype = ctypes.c_uint64
    lib.hsrle_compress_workspace_size_codec.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32]
    general = lib.hsrle_compress_workspace_size(frame, 4096)
    assert lib.hsrle_compress_workspace_size_codec(cid("rle8_packed_multi"), frame, 4096) == general
    assert lib.hsrle_compress_workspace_size_codec(cid("rle8_single"), frame, 4096) == general          # round 6: position-parallel, no split regions
    assert lib.hsrle_compress_workspace_size_codec(cid("rle8_single_short"), frame, 4096) == general
    assert lib.hsrle_compress_workspace_size_codec(cid("rle16_1symlut_byte_short_greedy"), frame, 4096) > 2 * frame > general
    assert lib.hsrle_compress_workspace_size_codec(cid("rle32_1symlut_byte_short_greedy"), frame, 4096) > 2 * frame
    for name in ("rle16_3symlut_byte_short_greedy", "rle64_7symlut_byte_short_greedy"):     # lists of 3 / 7 symbols decide the greedy scan's runs: one lane per block
        assert lib.hsrle_encode_path(cid(name), frame, 8192) == RING and lib.hsrle_compress_workspace_size_codec(cid(name), frame, 4096) == general
    assert lib.hsrle_compress_workspace_size_codec(cid("rle128_sym"), 8 << 30, 4096) == lib.hsrle_compress_workspace_size(8 << 30, 4096)
    # round 6: blocks above 4 KiB of a windowed codec never take the split encode: no split regions in the codec's own workspace size (the general size keeps them)
    for name in ("rle8_packed_multi", "rle32_byte", "rle64_3symlut_byte", "rle16_7symlut_sym"):
        assert lib.hsrle_compress_workspace_size_codec(cid(name), frame, 8192) < 1.2 * frame < lib.hsrle_compress_workspace_size(frame, 8192), name
    assert lib.hsrle_compress_workspace_size_codec(cid("rle128_sym"), frame, 8192) == lib.hsrle_compress_workspace_size(frame, 8192)      # (not windowed: split encode)
    assert lib.hsrle_encode_path(cid("rle8_single_short"), frame, 4096) == PP and lib.hsrle_compress_workspace_size_codec(cid("rle8_single_short"), frame, 4096) == general   # round 6
    assert lib.hsrle_encode_path(-1, frame, 4096) == -1 and lib.hsrle_encode_path(0, frame, 1000) == -1 and lib.hsrle_encode_path(0, 0, 4096) == -1


def test_codec_table_matches_tests_table(lib):
    from hsrle_testlib import CODECS

    lib.hsrle_codec_name.restype = ctypes.c_char_p
    assert [lib.hsrle_codec_name(i).decode() for i in range(110)] == [c.key for c in CODECS]
