"""Position-parallel form of the 128 bit encoders (csrc/hsrle_encode128p.hip.h), checked against the oracle on the CPU.

The reference (src/rle128_extreme_cpu.h:32-497) walks: extend the current run 16 bytes at a time, decide, hop through a pair search, and near the end
of the input step byte by byte with a symbol it re-reads at every step.  The kernel works on bits instead: E[j] = (d[j] == d[j + 16]),
E1[j] = (d[j] == d[j + 1]).  This script restates the closed forms in plain Python, builds the stream and compares it with the oracle's
(python tools/rle128_pp_model.py [cases] [seed]).  CPU only.
"""
import os
import random
import struct
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from hsrle_testlib import CODEC_BY_KEY, Oracle, fuzz_sections, mixed_runs, FUZZ_LENGTHS  # noqa: E402

STATS = {"tail_pair": 0, "tail_mini": 0, "lead": 0}


def model(d, aligned, packed):
    n = len(d)
    S = 16
    range7 = packed and not aligned
    if not packed:
        SHORT, MEDIUM, LONG, MAXR = S + 4, 0, S + 11, 255
    else:
        SHORT, MEDIUM, LONG, MAXR = 3, S + 3, (S + 11 if range7 else S + 10), (127 if range7 else 255)
    E = [1 if j + 16 < n and d[j] == d[j + 16] else 0 for j in range(n)] + [0] * 64
    E1 = [1 if j + 1 < n and d[j] == d[j + 1] else 0 for j in range(n)] + [0] * 64
    T = n - 32
    pad = bytes(d) + b"\0" * 32

    def ones_from(B, j, cap=1 << 30):
        c = 0
        while c < cap and B[j + c]:
            c += 1
        return c

    last = b"\0" * 16
    lastRLE = 0
    out = bytearray(struct.pack("<II", n, 0))
    ended = False

    def decide(sym, count, rng):
        if not packed:
            short_ok = rng <= MAXR and count >= SHORT
        else:
            short_ok = rng <= MAXR and ((count >= SHORT and sym == last) or count >= MEDIUM)
        return 1 if short_ok else (2 if count >= LONG else 0)

    def put(sym, p, e, k):
        nonlocal last, lastRLE
        count = e - p
        rng = p - lastRLE + 1
        c = (count // S - SHORT // S + 1) if aligned else (count - SHORT + 1)
        if not packed:
            out.extend(sym)
            out.extend(bytes([c]) if c <= 255 else b"\0" + struct.pack("<I", c))
        else:
            same = 0x80 if sym == last else 0
            last = sym
            out.extend(bytes([c | same]) if c <= 127 else bytes([same]) + struct.pack("<I", c))
            if not same:
                out.extend(sym)
        if k == 1:
            out.append((rng << 1) & 0xFF if range7 else rng)
        elif range7:
            out.extend(struct.pack("<I", (rng << 1) | 1))
        else:
            out.extend(b"\0" + struct.pack("<I", rng))
        out.extend(d[lastRLE:p])
        lastRLE = e

    # ---- the block starts inside a run of its first symbol (count 0, no pair needed) ----
    i = 0
    if n > 16:
        L0 = ones_from(E, 0)
        i = 16
        while i < n - 16 and i <= L0:
            i += 16
        if not aligned and i < n - 16:
            i = L0 + 16
        sym = pad[0:16]
        k = decide(sym, i, 1)
        if k:
            STATS["lead"] += 1
            put(sym, 0, i, k)
    # ---- body: the first window of 16 set bits at or behind the resume position, in front of n - 32 ----
    while True:
        # hops of the pair search (:233-268): behind the highest clear bit of the window
        found = -1
        while i < T:
            win = E[i:i + 16]
            if all(win):
                found = i
                break
            hz = max(k for k in range(16) if not win[k])
            i += hz + 1
        if found < 0:
            break
        p = found
        L = ones_from(E, p)
        i = p + 32
        while i < n - 16 and i <= p + L:
            i += 16
        if not aligned and i < n - 16:
            i = p + L + 16
        sym = pad[p:p + 16]
        k = decide(sym, i - p, p - lastRLE + 1)
        if k:
            put(sym, p, i, k)
    # ---- tail: byte steps with a symbol re-read at every step (:270-300); i >= n - 32 here ----
    j = i
    final32 = False
    while j < n:
        sym = pad[j:j + 16] if j + 16 <= n else bytes(d[j:n]) + b"\0" * (16 - (n - j))
        if j == T and all(E[j:j + 16]):
            final32 = True
            STATS["tail_pair"] += 1
            break
        if j + 1 < n - 16:
            c1 = ones_from(E1, j, 16)
            if c1 >= 16:
                count = 16
            else:
                count = 0 if aligned else c1
            i2 = j + 1 + count
            k = decide(sym, count, i2 - lastRLE - count + 1)
            if k:
                STATS["tail_mini"] += 1
                # (put() takes p, e: the packet's literals are d[lastRLE .. i2 - count))
                put(sym, i2 - count, i2, k)
            j = i2
        else:
            j += 1
    if final32:
        sym = pad[T:T + 16]
        k = decide(sym, 32, n - lastRLE - 32 + 1)
        put(sym, T, n, k)
        ended = True
    # terminators (the 128 bit encoder always writes the plain end marker's range field: SURVEY.md A.5 q11)
    if not packed:
        out.extend(b"\0" * 16 + b"\0" + struct.pack("<I", 0))
    else:
        out.extend(b"\x80" + struct.pack("<I", 0))
    if ended:
        out.extend(b"\0" + struct.pack("<I", 0))
    else:
        kk = n - lastRLE
        if range7:
            out.extend(struct.pack("<I", ((kk + 1) << 1) | 1))
        else:
            out.extend(b"\0" + struct.pack("<I", kk + 1))
        out.extend(d[lastRLE:n])
    struct.pack_into("<I", out, 4, len(out))
    return bytes(out)


def periodic(rng, n, alphabet):
    out = bytearray()
    while len(out) < n:
        out += bytes(rng.randrange(alphabet) for _ in range(rng.choice([0, 1, 2, 5, 15, 16, 17, 40, 300])))
        P = rng.choice([1, 1, 2, 4, 8, 16, 16, 16, 32])
        sym = bytes(rng.randrange(alphabet) for _ in range(P))
        k = rng.choice([16, 17, 20, 31, 32, 33, 35, 36, 37, 40, 47, 48, 49, 63, 64, 65, 100, 300, 1000])
        out += (sym * (k // P + 2))[:k]
    return bytes(out[:n])


def tails(rng, n):
    """periodic data up to the end of the input whose symbol starts with a few equal bytes: the byte steps of the tail find `runs` of the last stored symbol"""
    c = rng.randrange(256)
    lead = rng.choice([2, 3, 4, 5, 8, 15, 16])
    sym = bytes([c]) * lead + bytes(rng.randrange(256) for _ in range(16 - lead))
    if rng.random() < 0.3:
        sym = bytes(rng.choice([c, c, c, (c + 1) & 255]) for _ in range(16))
    pre = bytes(rng.randrange(256) for _ in range(rng.choice([0, 1, 7, 40, 130, 300])))
    body = (sym * (n // 16 + 2))[: max(0, n - len(pre) - rng.choice([0, 0, 1, 2, 5, 17, 33]))]
    post = bytes(rng.choice([c, rng.randrange(256)]) for _ in range(n))
    return (pre + body + post)[:n]


def tails2(rng, n):
    """a stored run, a few other bytes, then the run's symbol again inside the last 32 bytes: Packed stores `runs` of 3 .. 16 equal bytes there"""
    c = rng.randrange(256)
    lead = rng.choice([4, 5, 8, 12, 16])
    sym = bytes([c]) * lead + bytes(rng.randrange(256) for _ in range(16 - lead))
    g = bytes(rng.choice([c, rng.randrange(256)]) for _ in range(rng.choice([0, 1, 2, 3, 5, 8, 13, 16])))
    endpart = (sym * 3)[: rng.choice([17, 18, 20, 24, 30, 31, 32, 33, 40])]
    k = max(32, n - len(g) - len(endpart) - rng.choice([0, 3, 50]))
    pre = bytes(rng.randrange(256) for _ in range(rng.choice([0, 3, 50])))
    return pre + (sym * (k // 16 + 1))[:k] + g + endpart


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    ora = Oracle()
    gens = [lambda: fuzz_sections(rng, 8, FUZZ_LENGTHS), lambda: mixed_runs(rng, rng.choice([300, 3000, 4096])), lambda: periodic(rng, rng.choice([100, 600, 4096]), rng.choice([1, 2, 3, 256])), lambda: tails(rng, rng.choice([40, 64, 65, 100, 200, 777, 4096])), lambda: tails2(rng, rng.choice([100, 200, 777, 2000])), lambda: tails2(rng, rng.choice([100, 200, 777, 2000])),
            lambda: bytes(rng.randrange(rng.choice([1, 2, 3])) for _ in range(rng.choice([1, 15, 16, 17, 31, 32, 33, 47, 48, 49, 100, 4096])))]
    bad = 0
    for t in range(cases):
        data = rng.choice(gens)()
        if not data:
            continue
        data = data[:4096]
        cut = rng.choice([0, 0, 1, 3, 15, 16, 17])
        if cut and len(data) > cut:
            data = data[:len(data) - cut]
        for key in ("rle128_sym", "rle128_sym_packed", "rle128_byte", "rle128_byte_packed"):
            want = ora.compress(CODEC_BY_KEY[key], data)
            got = model(data, "sym" in key, "packed" in key)
            if got != want:
                bad += 1
                if bad <= 5:
                    print("MISMATCH", key, "len", len(data), "case", t, "model", len(got), "oracle", len(want), flush=True)
                    with open(f"/tmp/rle128_model_bad_{bad}.bin", "wb") as f:
                        f.write(data)
    print("cases", cases, "mismatches", bad, STATS)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
