"""Mint the golden vectors of this repository from the COMPILED REFERENCE (oracle/_ref/libhsrle_ref.so, built from
/root/reference/src by oracle/Makefile).  Run in a container that has /root/reference:

    python tests/golden/make_golden.py

Outputs (data only -- inputs and expected outputs, no reference source):
  tests/golden/vectors.json           inputs (base64) + per codec {size, sha256} of the reference stream, plus the full
                                      streams of the 79-byte worked example of SURVEY.md A.6
  tests/golden/rle8_packed_tails.json rle8_packed_multi streams of both encoder tail flavours (SSE2 body vs AVX2 body,
                                      SURVEY.md A.5 q1) for small inputs; the canonical one is AVX2
  tests/golden/rle8m_vectors.json     {size, sha256} of the reference's rle8m stream of every vectors.json input for 1/2/3/7/16 sections
  tests/golden/synth_manifest.json    {size, sha256} of the reference stream of every 1 MiB synthetic buffer
                                      (run-distributed per symbol width, video-shaped), cut into 64 KiB blocks, and of
                                      the monolithic 1 MiB stream

The reference encoders for widths > 8 bit read past inSize; inputs are guard padded as SURVEY.md §8c prescribes
(hsrle_testlib.guard_pad) so "bytes beyond the end never match".
"""
import base64
import hashlib
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from hsrle_testlib import CODECS, FUZZ_LENGTHS, Oracle, Reference, fuzz_sections, mixed_runs, single_symbol_mix  # noqa: E402

WORKED = b"ABCDE" + b"x" * 12 + b"FG" + b"x" * 4 + b"HIJ" + b"y" * 3 + b"KLMNOPQRSTUVWXYZ0123456789abcdefghijklmnopqrstuvw" + b"\x00"


def sha(b):
    return hashlib.sha256(b).hexdigest()


def inputs():
    rng = random.Random(20240601)
    out = [("worked_example", WORKED)]
    for n in (1, 2, 3, 5, 8, 15, 16, 17, 31, 32, 33, 34, 47, 48, 49, 63, 64, 65, 66, 95, 96):
        out.append((f"mixed_{n}", mixed_runs(rng, n)))
    for k in range(12):
        out.append((f"fuzz_small_{k}", fuzz_sections(rng)))
    for k in range(4):
        out.append((f"fuzz_long_{k}", fuzz_sections(rng, lengths=FUZZ_LENGTHS, max_sections=3)))
    for k in range(6):
        out.append((f"single_{k}", single_symbol_mix(rng, rng.choice([100, 333, 1000, 3000, 9000]))))
    # LUT stress: runs cycling through 2,3,4,8,9 distinct symbols
    for ns in (2, 3, 4, 8, 9):
        b = bytearray()
        for k in range(60):
            b += bytes([(k % ns) * 17 + 1]) * rng.choice([3, 4, 5, 12, 20]) + bytes(rng.randrange(256) for _ in range(rng.choice([0, 1, 2, 7])))
        out.append((f"lut_cycle_{ns}", bytes(b)))
    # Packed stress: same-symbol runs of length 2,3,4 separated by 0..130 literals
    b = bytearray()
    for gap in list(range(0, 12)) + [125, 126, 127, 128, 129, 130]:
        for ln in (2, 3, 4):
            b += b"\x07" * ln + bytes((i * 7 + 3) % 251 + 1 for i in range(gap))
    out.append(("packed_stress", bytes(b)))
    return [(n, d) for n, d in out if len(d) > 0]


def main():
    ref = Reference()
    assert ref.has_avx2, "the canonical rle8_packed_multi stream needs the AVX2 encoder body"
    ora = Oracle()

    vec = {"note": "minted by tests/golden/make_golden.py from the compiled reference", "inputs": []}
    for name, data in inputs():
        entry = {"name": name, "input": base64.b64encode(data).decode(), "codecs": {}}
        for c in CODECS:
            s = ref.compress(c, data)
            assert s is not None and ref.decompress(c, s) == data
            e = {"size": len(s), "sha256": sha(s)}
            if name == "worked_example":
                e["stream"] = base64.b64encode(s).decode()
            entry["codecs"][c.key] = e
        vec["inputs"].append(entry)
    json.dump(vec, open(os.path.join(HERE, "vectors.json"), "w"), indent=0)

    # both tail flavours of rle8_packed_multi
    packed = [c for c in CODECS if c.key == "rle8_packed_multi"][0]
    rng = random.Random(99)
    tails = []
    for n in (33, 40, 48, 64, 65, 70, 90, 100, 128, 200, 333, 632):
        for _ in range(2):
            d = mixed_runs(rng, n, alphabet=rng.choice([2, 3, 256]))
            ref.set_max_simd(1)
            sse2 = ref.compress(packed, d)
            ref.set_max_simd(0)
            avx2 = ref.compress(packed, d)
            tails.append({"input": base64.b64encode(d).decode(), "sse2": base64.b64encode(sse2).decode(), "avx2": base64.b64encode(avx2).decode()})
    ref.set_max_simd(0)
    json.dump(tails, open(os.path.join(HERE, "rle8_packed_tails.json"), "w"), indent=0)
    print("tail flavours differing:", sum(1 for t in tails if t["sse2"] != t["avx2"]), "of", len(tails))

    # synthetic workloads: 1 MiB per symbol width (+ video shaped), 64 KiB blocks and monolithic
    man = {"size": 1 << 20, "block": 65536, "seed": 1, "entries": {}}
    for c in CODECS:
        for kind in (0, 1):
            if kind == 1 and c.key not in ("rle8_packed_multi", "rle64_3symlut_byte", "rle8_single", "rle8_packed_single"):
                continue
            buf = ora.synth(kind, c.S, 1, 1 << 20)
            data = buf.tobytes()
            mono = ref.compress(c, data)
            blocks = [ref.compress(c, data[i : i + 65536]) for i in range(0, len(data), 65536)]
            man["entries"][f"{c.key}/kind{kind}"] = {
                "mono": {"size": len(mono), "sha256": sha(mono)},
                "blocks": [{"size": len(b), "sha256": sha(b)} for b in blocks],
            }
    json.dump(man, open(os.path.join(HERE, "synth_manifest.json"), "w"), indent=0)

    # rle8m: reference streams of the small inputs for a few section counts (None where rle8m_compress gives up)
    r8 = []
    for name, data in inputs():
        for sections in (1, 2, 3, 7, 16):
            if len(data) // sections == 0:
                continue
            st = ref.rle8m_compress(sections, data)
            assert st is None or ref.rle8m_decompress(st, len(data)) == data
            r8.append({"name": name, "sections": sections, "size": 0 if st is None else len(st), "sha256": None if st is None else sha(st)})
    json.dump(r8, open(os.path.join(HERE, "rle8m_vectors.json"), "w"), indent=0)
    print("golden vectors written:", len(vec["inputs"]), "inputs x", len(CODECS), "codecs")


if __name__ == "__main__":
    main()
