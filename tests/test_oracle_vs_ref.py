"""Differential test: the oracle against the compiled reference (oracle/_ref/libhsrle_ref.so) on the reference fuzzer's
input grammar.  Runs wherever the reference library exists (this container; the GPU box carries the prebuilt .so);
skipped otherwise -- the committed golden vectors (test_oracle_golden.py) pin the oracle there."""
import random

from hsrle_testlib import CODECS, FUZZ_LENGTHS, fuzz_sections, mixed_runs, single_symbol_mix


def _inputs(seed, count):
    rng = random.Random(seed)
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
            yield d


def test_streams_identical_and_cross_decode(oracle, reference):
    n = 0
    for d in _inputs(4242, 240):
        for c in CODECS:
            r = reference.compress(c, d)
            m = oracle.compress(c, d)
            assert r == m, f"{c.key}: oracle stream differs from the reference (input {len(d)} bytes)"
            assert oracle.decompress(c, r) == d
            n += 1
    assert n > 10000


def test_reference_decodes_oracle_streams(oracle, reference):
    for d in _inputs(7, 40):
        for c in CODECS:
            assert reference.decompress(c, oracle.compress(c, d)) == d


def test_bounds_helpers(oracle, reference):
    for n in (0, 1, 100, 1 << 20, 1 << 30, (1 << 30) + 1):
        assert oracle.bounds(n) == reference.lib.rle_compress_bounds(n)


def test_rle8m_streams_identical(oracle, reference):
    """rle8m: the oracle's restatement against rle8m_compress / rle8m_decompress of the compiled reference,
    including the inputs on which the reference gives up (a section that outgrows the room left in the output)."""
    rng = random.Random(5)
    n = 0
    for d in _inputs(21, 400):
        for sections in (1, 2, 3, rng.choice([4, 5, 7, 8, 16, 33])):
            if len(d) // sections == 0:
                continue
            r, m = reference.rle8m_compress(sections, d), oracle.rle8m_compress(sections, d)
            assert r == m, f"rle8m x{sections}: oracle stream differs from the reference (input {len(d)} bytes)"
            if r is not None:
                assert oracle.rle8m_decompress(r, len(d)) == d and reference.rle8m_decompress(m, len(d)) == d
            n += 1
    assert n > 1000
