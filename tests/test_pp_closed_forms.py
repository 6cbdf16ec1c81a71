"""The phase-free closed forms behind the round-6 position-parallel encoders (8 bit Single: csrc/hsrle_encode8sp.hip.h; 128 bit: csrc/hsrle_encode128p.hip.h),
restated in plain Python (tools/single_pp_model.py, tools/rle128_pp_model.py) and checked against the oracle on the CPU: what the reference's window scanners
DECIDE does not depend on where their windows happen to stand, except in the two places the models replay.  Reference: src/rle8_extreme_cpu.h:346-700,
:1103-1321, src/rle128_extreme_cpu.h:32-497."""
import os
import random
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))

from hsrle_testlib import CODEC_BY_KEY, FUZZ_LENGTHS, fuzz_sections, mixed_runs, single_symbol_mix


def _inputs(rng, gens, cases):
    for _ in range(cases):
        data = rng.choice(gens)()
        if not data:
            continue
        data = data[:4096]
        cut = rng.choice([0, 0, 1, 3, 15, 16, 17])
        if cut and len(data) > cut:
            data = data[: len(data) - cut]
        yield data


@pytest.mark.parametrize("key", ["rle8_single", "rle8_packed_single"])
def test_single_decisions_have_a_phase_free_closed_form(oracle, key):
    import single_pp_model as M

    rng = random.Random(606)
    gens = [lambda: fuzz_sections(rng, 8, FUZZ_LENGTHS), lambda: mixed_runs(rng, rng.choice([300, 3000, 4096])), lambda: single_symbol_mix(rng, rng.choice([100, 3000, 4096])),
            lambda: M.triples(rng, rng.choice([500, 4096]), rng.choice([0, 7, 255])), lambda: bytes(rng.randrange(rng.choice([2, 3, 5])) for _ in range(rng.choice([1, 15, 16, 17, 33, 100, 4096])))]
    n = 0
    for data in _inputs(rng, gens, 400):
        want = oracle.compress(CODEC_BY_KEY[key], data)
        assert M.model(data, want[9], key == "rle8_packed_single") == want, f"{key}: the closed form differs from the oracle on {len(data)} bytes"
        n += 1
    assert n >= 300


@pytest.mark.parametrize("key", ["rle128_sym", "rle128_sym_packed", "rle128_byte", "rle128_byte_packed"])
def test_rle128_walk_has_a_closed_form_on_match_bits(oracle, key):
    import rle128_pp_model as M

    rng = random.Random(128)
    gens = [lambda: fuzz_sections(rng, 8, FUZZ_LENGTHS), lambda: mixed_runs(rng, rng.choice([300, 3000, 4096])), lambda: M.periodic(rng, rng.choice([100, 600, 4096]), rng.choice([1, 2, 3, 256])),
            lambda: M.tails(rng, rng.choice([40, 64, 65, 100, 200, 777, 4096])), lambda: M.tails2(rng, rng.choice([100, 200, 777, 2000])),
            lambda: bytes(rng.randrange(rng.choice([1, 2, 3])) for _ in range(rng.choice([1, 15, 16, 17, 31, 32, 33, 47, 48, 49, 100, 4096])))]
    n = 0
    for data in _inputs(rng, gens, 250):
        want = oracle.compress(CODEC_BY_KEY[key], data)
        assert M.model(data, "sym" in key, "packed" in key) == want, f"{key}: the closed form differs from the oracle on {len(data)} bytes"
        n += 1
    assert n >= 200
