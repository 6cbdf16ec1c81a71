"""The closed form of the 8 bit Single symbol pick that k_single_pick (csrc/hsrle_encode8s.hip.h) computes wave-parallel, restated in
python (tools/pick_model.py) and checked against the oracle's literal restatement of the reference's estimator
(src/rle8_extreme_cpu.c:53-153; byte 9 of an rle8_single stream is the picked symbol).  CPU only."""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))

from hsrle_testlib import CODEC_BY_KEY, Oracle
from pick_model import gen, pick


def test_closed_form_pick_equals_the_reference_estimator():
    ora = Oracle()
    codec = CODEC_BY_KEY["rle8_single"]
    rng = random.Random(2024)
    sizes = [1, 2, 15, 16, 17, 18, 31, 32, 33, 34, 47, 48, 49, 64, 65, 100, 257, 600, 4096]
    for t in range(4000):
        n = rng.choice(sizes) if t % 2 else rng.randrange(1, 700)
        d = gen(rng, n)
        assert pick(d) == ora.compress(codec, d)[9], f"case {t}: {n} bytes {d.hex()[:200]}"
