"""The oracle (oracle/hsrle_oracle.c) against the committed golden vectors minted from the compiled reference
(tests/golden/make_golden.py).  This is what pins the oracle on a machine that has no /root/reference."""
import base64
import hashlib
import json
import os

import pytest

from hsrle_testlib import CODECS, CODEC_BY_KEY

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def sha(b):
    return hashlib.sha256(b).hexdigest()


@pytest.fixture(scope="module")
def vectors():
    return json.load(open(os.path.join(GOLDEN, "vectors.json")))


def test_encode_matches_golden(oracle, vectors):
    checked = 0
    for entry in vectors["inputs"]:
        data = base64.b64decode(entry["input"])
        for c in CODECS:
            g = entry["codecs"][c.key]
            s = oracle.compress(c, data)
            assert s is not None and len(s) == g["size"] and sha(s) == g["sha256"], f"{c.key} on {entry['name']}"
            if "stream" in g:
                assert s == base64.b64decode(g["stream"])
            assert oracle.decompress(c, s) == data
            checked += 1
    assert checked == len(CODECS) * len(vectors["inputs"])


def test_worked_example_bytes(oracle, vectors):
    """the 79-byte example, checked against the literal bytes quoted there."""
    entry = [e for e in vectors["inputs"] if e["name"] == "worked_example"][0]
    data = base64.b64decode(entry["input"])
    s = oracle.compress(CODEC_BY_KEY["rle8_packed_multi"], data)
    assert len(s) == 86
    assert s[:9] == bytes.fromhex("4F00000056000000" + "00")
    assert s[9:12] == bytes.fromhex("0A780C") and s[12:17] == b"ABCDE"
    assert s[17:19] == bytes.fromhex("8206") and s[19:21] == b"FG"
    assert s[21:30] == bytes.fromhex("80" + "00000000" + "73000000")
    assert len(oracle.compress(CODEC_BY_KEY["rle8_multi"], data)) == 90
    assert len(oracle.compress(CODEC_BY_KEY["rle8_3symlut"], data)) == 84
    assert len(oracle.compress(CODEC_BY_KEY["rle64_3symlut_byte"], data)) == 95


def test_decoder_is_tail_flavour_invariant(oracle):
    """8 bit Packed: streams of the SSE2-body and of the AVX2-body encoder both decode; the oracle encoder is the AVX2 one."""
    packed = CODEC_BY_KEY["rle8_packed_multi"]
    cases = json.load(open(os.path.join(GOLDEN, "rle8_packed_tails.json")))
    differing = 0
    for case in cases:
        data = base64.b64decode(case["input"])
        sse2, avx2 = base64.b64decode(case["sse2"]), base64.b64decode(case["avx2"])
        assert oracle.decompress(packed, sse2) == data
        assert oracle.decompress(packed, avx2) == data
        assert oracle.compress(packed, data) == avx2
        differing += sse2 != avx2
    assert differing > 0  # the fixture really holds both flavours


def test_synthetic_manifest(oracle):
    """1 MiB synthetic buffers (the bench workloads at small size): monolithic and 64 KiB block streams match the reference."""
    man = json.load(open(os.path.join(GOLDEN, "synth_manifest.json")))
    for key, e in man["entries"].items():
        ckey, kind = key.split("/kind")
        c = CODEC_BY_KEY[ckey]
        buf = oracle.synth(int(kind), c.S, man["seed"], man["size"])
        mono = oracle.compress(c, buf.tobytes())
        assert (len(mono), sha(mono)) == (e["mono"]["size"], e["mono"]["sha256"]), key
        streams = oracle.compress_blocks(c, buf, man["block"])
        assert [(len(s), sha(s)) for s in streams] == [(b["size"], b["sha256"]) for b in e["blocks"]], key


def test_error_returns(oracle):
    """Argument / header checks of the reference (rle8_extreme_cpu.h:88-89, :704-712; rleX_extreme_cpu.h:49-50, :84-91)."""
    d = b"abcabcabc" * 10
    for c in CODECS:
        s = oracle.compress(c, d)
        size, _ = oracle.call(c.cname, d, oracle.bounds(len(d)) - 1)
        assert size == 0
        size, _ = oracle.call(c.cname, b"", 1000)
        assert size == 0
        size, _ = oracle.call(c.dname, s, len(d) - 1)
        assert size == 0
        size, _ = oracle.call(c.dname, s[:-1], len(d))
        assert size == 0
        size, out = oracle.call(c.dname, s, len(d) + 77)  # outSize larger than needed is fine (main.c:970)
        assert size == len(d) and out == d
    assert oracle.call("rle8_nonexistent_compress", d, 1000)[0] == 0xFFFFFFFF
    assert oracle.bounds(100) == 293 and oracle.bounds((1 << 30) + 1) == 0


def test_rle8m_matches_golden(oracle, vectors):
    """rle8m streams against the vectors minted from the compiled reference."""
    inputs = {e["name"]: base64.b64decode(e["input"]) for e in vectors["inputs"]}
    r8 = json.load(open(os.path.join(GOLDEN, "rle8m_vectors.json")))
    assert len(r8) > 150
    for g in r8:
        data = inputs[g["name"]]
        s = oracle.rle8m_compress(g["sections"], data)
        if g["sha256"] is None:
            assert s is None
        else:
            assert s is not None and len(s) == g["size"] and sha(s) == g["sha256"], f"rle8m x{g['sections']} on {g['name']}"
            assert oracle.rle8m_decompress(s, len(data)) == data
