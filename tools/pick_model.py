"""Closed form of the 8 bit Single symbol pick checked against the oracle (symbol byte of the stream)."""
import random, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from hsrle_testlib import Oracle, CODEC_BY_KEY

M32 = 0xFFFFFFFF

def pick(d):
    n = len(d)
    prob = [0] * 256
    pc = [0] * 256
    if d[0] != 0:
        pc[0] = M32
    end = n - 16
    def reg(s, c):
        prob[s] = (prob[s] + c) & M32
        pc[s] = (pc[s] + 1) & M32
    last = (~d[0]) & 0xFF
    if end <= 0:
        reg(last, 0)
    else:
        # the first loop trip: window [0, 16) against ~d[0]
        if last in d[0:16]:
            reg(last, 0)
        # maximal runs of >= 2 equal bytes
        runs = []
        j = 0
        while j < n - 1:
            if d[j] == d[j + 1]:
                e = j + 1
                while e < n and d[e] == d[j]:
                    e += 1
                runs.append((j, e - j))
                j = e
            else:
                j += 1
        i0 = 0
        fin = None
        for (j, L) in runs:
            # search from i0 (i0 < end holds here)
            q = i0 + 15 * ((j - i0) // 15)
            if q >= end:
                break
            # found at j: last = d[j], count = 1, i = j + 1
            if j + L < end:
                reg(d[j], L - (L - 1) // 16)
                i0 = j + L
                continue
            # late run: literal
            i = j + 1
            count = 1
            done = False
            while i < end:
                rem = L - (i - j)
                if rem >= 16:
                    count += 15; i += 16
                else:
                    count += rem; i += rem
                    reg(d[j], count)
                    done = True
                    break
            if not done:
                fin = (d[j], count)
            else:
                # registered with i = j + L >= end: the search loop is skipped
                fin = (d[i], 1)
            break
        if fin is None:
            # the search runs off the end
            i = i0 + 15 * ((end - i0 + 14) // 15) if i0 < end else i0
            fin = (d[i], 1)
        reg(*fin)
    best, bs = 0, 0
    for s in range(256):
        if pc[s] > 0 and prob[s] // pc[s] > 2:
            saved = (prob[s] - pc[s] * 2) & M32
            if saved > best:
                best, bs = saved, s
    return bs

def gen(rng, n):
    mode = rng.randrange(6)
    out = bytearray()
    alpha = [rng.randrange(256) for _ in range(rng.choice([1, 2, 3, 5, 40]))]
    while len(out) < n:
        if mode == 0:
            out.append(rng.choice(alpha))
        else:
            out += bytes([rng.choice(alpha)]) * rng.choice([1, 1, 2, 3, 4, 5, 15, 16, 17, 18, 31, 32, 33, 34, 48, 49, 50, 100])
            out += bytes(rng.randrange(256) for _ in range(rng.choice([0, 0, 1, 2, 14, 15, 16, 30])))
    return bytes(out[:n])

if __name__ == "__main__":
    ora = Oracle()
    codec = CODEC_BY_KEY["rle8_single"]
    rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    bad = 0
    for t in range(N):
        n = rng.choice([1, 2, 15, 16, 17, 18, 31, 32, 33, 34, 47, 48, 49, 64, 65, 100, 257, 600, 4096]) if rng.random() < 0.5 else rng.randrange(1, 700)
        d = gen(rng, n)
        want = ora.compress(codec, d)[9]
        got = pick(d)
        if want != got:
            bad += 1
            if bad < 6:
                print("MISMATCH n", n, "want", want, "got", got, d.hex())
    print("cases", N, "mismatches", bad)
