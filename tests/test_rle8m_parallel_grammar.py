"""The position-parallel restatements behind the wave-per-section rle8m kernels (hsrle_rle8m.hip.h: k_rle8m_decode_wave,
k_rle8m_encode_wave), checked on the CPU against the oracle's sequential codec (which is pinned against the compiled reference,
test_oracle_vs_ref.py).  Pure host logic: no GPU, no library call.

decode: a stream byte is a repeat code iff the run of flagged-valued bytes right before it has odd length.
encode: a symbol without a repeat code is its own token; inside a maximal run of a flagged symbol -- cut at the section start and
        before the section's last byte -- tokens start every 255 bytes; the token that ENDS at a byte carries (distance to the run
        start) mod 255 as its count."""
import random
import struct

from hsrle_testlib import fuzz_sections, mixed_runs, single_symbol_mix


def _parse(stream):
    size, total, sections = struct.unpack_from("<III", stream, 0)
    assert size == len(stream)
    ends = list(struct.unpack_from(f"<{sections - 1}I", stream, 12)) + [size]
    at = 12 + 4 * (sections - 1)
    flagged = [bool(stream[at + s // 8] >> (s % 8) & 1) for s in range(256)]
    listed = stream[at + 32] or 255
    symbols = list(stream[at + 33 : at + 33 + listed])
    code_to_count = {sym: k for k, sym in enumerate(symbols)}
    nxt = listed
    for s in range(256):
        if s not in code_to_count:
            code_to_count[s] = nxt & 0xFF
            nxt += 1
    count_to_code = {}
    for s in range(256):
        count_to_code.setdefault(code_to_count[s], s)
    begin = at + 33 + listed
    bounds = [(begin if k == 0 else ends[k - 1], ends[k]) for k in range(sections)]
    return total, sections, flagged, code_to_count, count_to_code, bounds


def _decode_section_parallel(sec, flagged, code_to_count):
    """Every position decides on its own: role from the parity of the flagged run before it, packet length from its code."""
    out = bytearray()
    n = len(sec)
    roles = []
    for i in range(n):
        r = 0
        while r < i and flagged[sec[i - 1 - r]]:
            r += 1
        roles.append(r & 1)                      # 1 = repeat code
    for i in range(n):
        if roles[i]:
            continue
        b = sec[i]
        reps = 0
        if flagged[b]:
            assert i + 1 < n and roles[i + 1] == 1
            reps = code_to_count[sec[i + 1]]
        out += bytes([b]) * (1 + reps)
    return bytes(out)


def _encode_section_parallel(data, flagged, count_to_code):
    n = len(data)
    out = bytearray()
    run_start = 0
    for i in range(n):
        if i == 0 or data[i] != data[i - 1] or i == n - 1:
            run_start = i
        b = data[i]
        rel = i - run_start
        if not flagged[b]:
            out.append(b)
            continue
        next_breaks = i + 1 >= n - 1 or data[i + 1] != b
        if i == n - 1 or next_breaks or (rel + 1) % 255 == 0:
            out += bytes([b, count_to_code[rel % 255]])
    return bytes(out)


def test_parallel_grammar_equals_the_sequential_codec(oracle):
    rng = random.Random(99)
    checked = 0
    for it in range(220):
        k = it % 4
        if k == 0:
            data = mixed_runs(rng, rng.choice([50, 700, 3000, 9000]), alphabet=rng.choice([2, 3, 8, 256]))
        elif k == 1:
            data = single_symbol_mix(rng, rng.choice([300, 2000, 8000]))
        elif k == 2:
            data = fuzz_sections(rng, max_sections=4)
        else:
            data = bytes([rng.randrange(3)]) * rng.choice([254, 255, 256, 509, 510, 511, 766, 1500]) + mixed_runs(rng, 64, alphabet=3) + bytes([1]) * rng.choice([1, 2, 255, 300])
        if not data:
            continue
        sections = rng.choice([1, 2, 3, 5, 8])
        if len(data) // sections == 0:
            sections = 1
        stream = oracle.rle8m_compress(sections, data)
        if stream is None:
            continue
        total, nsec, flagged, code_to_count, count_to_code, bounds = _parse(stream)
        assert (total, nsec) == (len(data), sections)
        ss = len(data) // sections
        for s, (a, b) in enumerate(bounds):
            piece = data[s * ss : (s + 1) * ss] if s + 1 < sections else data[s * ss :]
            assert _decode_section_parallel(stream[a:b], flagged, code_to_count) == piece, f"decode rule: section {s} of {sections}, {len(data)} bytes"
            assert _encode_section_parallel(piece, flagged, count_to_code) == stream[a:b], f"encode rule: section {s} of {sections}, {len(data)} bytes"
            checked += 1
    assert checked > 300
