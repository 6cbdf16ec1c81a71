"""Position-parallel form of the 8 bit Single encoders' decisions (csrc/hsrle_encode8sp.hip.h), checked against the oracle on the CPU.

The reference's scanner (src/rle8_extreme_cpu.h:1103-1321 body, :383-694 tail + final block) walks 16-byte windows at a data dependent phase.
The kernel does not: it uses the closed forms below.  This script restates them in plain Python, builds the whole stream from them and compares
it with the oracle's (python tools/single_pp_model.py [cases] [seed]).  CPU only.

  * every maximal run of the symbol of >= SHORT bytes is found with its true start, whatever the phase of the windows;
  * a run (p, L) is judged by the BODY's rule iff its deciding trip starts in front of n - 16:  p == 0 ? 16 (L / 16) : p + 1 + 16 ((L - 1) / 16);
  * body rule: range <= 255 -> short form; L >= LONG (Packed: L >= MEDIUM) -> long form; else a wasted chance: the third one within 255 bytes
    of the first makes all three stored (the first in the long form, the other two short);
  * tail rule: range <= 255 -> short form, L >= LONG -> long form;
  * where the body's search runs off n - 16 it skips one byte: a run that starts exactly there loses its first byte (phase dependent: the windows
    are replayed from the last place where the phase is known).
"""
import os
import random
import struct
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from hsrle_testlib import CODEC_BY_KEY, Oracle, fuzz_sections, mixed_runs, single_symbol_mix, FUZZ_LENGTHS  # noqa: E402


def model(d, sym, packed):
    n = len(d)
    SHORT, LONG = (2, 10) if packed else (4, 8)
    SURE = 6 if packed else 8          # body: stored whatever the range
    end = n - 16
    m = [1 if b == sym else 0 for b in d] + [0] * 64
    # maximal runs of the symbol
    runs = []
    j = 0
    while j < n:
        if m[j]:
            e = j
            while e < n and m[e]:
                e += 1
            runs.append((j, e))
            j = e
        else:
            j += 1
    # ---- the skipped byte ----
    quirk = -1
    if end > 0 and any(max(end, 1) <= p <= end + 15 for (p, e) in runs):
        a0 = 0
        if m[0]:
            a0 = min(runs[0][1], 64)
        for (p, e) in runs:
            if e - p >= SHORT and e < end:
                a0 = max(a0, e)
        w = a0
        hit = False
        while w < end:
            win = m[w:w + 16]
            pop = sum(win)
            if pop == 0 or (not win[15] and pop < SHORT):
                w += 16
                continue
            f = w + win.index(1)
            Lf = 0
            while Lf < SHORT and m[f + Lf]:
                Lf += 1
            if Lf >= SHORT:
                hit = True
                break
            w = f + Lf
        if not hit and m[w]:
            quirk = w
    # ---- candidates ----
    cands = []
    for (p, e) in runs:
        L = e - p
        if L < SHORT:
            continue
        ie = 16 * (L // 16) if p == 0 else p + 1 + 16 * ((L - 1) // 16)
        vec = end > 0 and ie < end
        if p == quirk:
            p += 1
        cands.append([p, e, vec, 0, 0])   # k, inL
    # ---- decisions ----
    lastRLE, w, fw = 0, 0, 0
    for j, c in enumerate(cands):
        p, e, vec = c[0], c[1], c[2]
        L = e - p
        rng = p - lastRLE + 1
        if L < SHORT:
            continue
        c[4] = lastRLE
        if vec:
            if rng <= 255:
                c[3] = 1; lastRLE = e; w = 0
            elif L >= SURE:
                c[3] = 2; lastRLE = e; w = 0
            else:
                w += 1
                if w == 1 or e - fw > 255:
                    fw = p; w = 1
                elif w > 2:
                    cands[j - 2][3] = 2
                    cands[j - 1][3] = 1; cands[j - 1][4] = cands[j - 2][1]
                    c[3] = 1; c[4] = cands[j - 1][1]
                    lastRLE = e; w = 0
        else:
            if rng <= 255:
                c[3] = 1; lastRLE = e
            elif L >= LONG:
                c[3] = 2; lastRLE = e
    # ---- the stream ----
    out = bytearray(struct.pack("<IIBB", n, 0, 1, sym))
    ended = False
    for (p, e, vec, k, inL) in cands:
        if not k:
            continue
        c = e - p - SHORT + 1
        out += bytes([c]) if c <= 255 else b"\0" + struct.pack("<I", c)
        rng = p - inL + 1
        out += bytes([rng]) if k == 1 else b"\0" + struct.pack("<I", rng)
        out += bytes(d[inL:p])
        if e >= n:
            ended = True
    if ended:
        out += b"\0" * 10
    else:
        out += b"\0" * 6 + struct.pack("<I", n - lastRLE + 1) + bytes(d[lastRLE:n])
    struct.pack_into("<I", out, 4, len(out))
    return bytes(out)


def model_short(d, sym):
    """rle8_single_short (src/rleX_Xsl_short.h with SINGLE: wrapper :380-523, body :1058-1120, process_symbol :152-372): the same window scanner, no wasted
    chances, one rule for body and tail -- count >= 11, or count >= 2 + the bytes the packet needs beyond the one-byte form"""
    n = len(d)
    MINS, MINL, MAXPR, MAXPC, MAXTR, CINV, RBP, RB, TB = 2, 11, 15, 14, 2047, 15, 4, 11, 8
    end = n - 16
    m = [1 if b == sym else 0 for b in d] + [0] * 64
    runs = []
    j = 0
    while j < n:
        if m[j]:
            e = j
            while e < n and m[e]:
                e += 1
            runs.append((j, e))
            j = e
        else:
            j += 1
    quirk = -1
    if end > 0 and any(max(end, 1) <= p <= end + 15 for (p, e) in runs):
        a0 = 0
        if m[0]:
            a0 = min(runs[0][1], 64)
        for (p, e) in runs:
            if e - p >= 2 and e < end:
                a0 = max(a0, e)
        w = a0
        hit = False
        while w < end:
            win = m[w:w + 16]
            pop = sum(win)
            if pop == 0 or (not win[15] and pop < 2):
                w += 16
                continue
            f = w + win.index(1)
            Lf = 0
            while Lf < 2 and m[f + Lf]:
                Lf += 1
            if Lf >= 2:
                hit = True
                break
            w = f + Lf
        if not hit and m[w]:
            quirk = w
    out = bytearray(struct.pack("<IIB", n, 0, sym))
    lastRLE = 0
    ended = False
    for (p, e) in runs:
        if p == quirk:
            p += 1
        count = e - p
        if count < MINS:
            continue
        gap = p - lastRLE
        rng = gap + 2
        single = gap <= MAXPR and count - 2 <= MAXPC
        pen = 0
        if not single:
            pen = 2 + (0 if rng <= MAXTR else 2) + (0 if count <= 511 else 2)
        if not (count >= MINL or count >= MINS + pen):
            continue
        if single:
            out.append(((count - 2) << RBP) | gap)
        else:
            scx = count if count <= 511 else 1
            rx = rng if rng <= MAXTR else 1
            out.append((CINV << RBP) | (((scx << (RB - 8)) >> 8) & 0xFF))
            out.append(((scx << (RB - 8)) | (rx >> 8)) & 0xFF)
            out.append(rx & 0xFF)
            if scx != count:
                out += struct.pack("<H", count)
            if rx != rng:
                out += struct.pack("<H", rng)
        out += bytes(d[lastRLE:p])
        lastRLE = e
        if e >= n:
            ended = True
    if ended:
        out += bytes([CINV << RBP, TB, 1]) + b"\0" * 4
    else:
        out += bytes([CINV << RBP, TB, 0]) + b"\0" * 2 + struct.pack("<I", n - lastRLE + 2) + bytes(d[lastRLE:n])
    struct.pack_into("<I", out, 4, len(out))
    return bytes(out)


def triples(rng, n, sym):
    """runs of the symbol of 2 .. 9 bytes in groups: gaps beyond 255 bytes in front of a group, a few bytes inside it"""
    out = bytearray()
    others = [b for b in range(256) if b != sym]
    while len(out) < n:
        for _ in range(rng.choice([1, 2, 3, 4, 70])):
            out += bytes(rng.choice(others) for _ in range(rng.choice([1, 2, 5, 30, 100, 124])))
            out += bytes([sym]) * rng.choice([1, 2, 2, 3, 3, 4, 5, 6, 7, 8, 9, 10, 17, 33])
        out += bytes(rng.choice(others) for _ in range(rng.choice([0, 200, 254, 255, 256, 257, 300, 600])))
    return bytes(out[:n])


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    ora = Oracle()
    gens = [lambda: fuzz_sections(rng, 8, FUZZ_LENGTHS), lambda: mixed_runs(rng, rng.choice([300, 3000, 4096])), lambda: single_symbol_mix(rng, rng.choice([100, 3000, 4096])),
            lambda: triples(rng, rng.choice([500, 4096]), rng.choice([0, 7, 255])), lambda: bytes(rng.randrange(rng.choice([2, 3, 5])) for _ in range(rng.choice([1, 15, 16, 17, 33, 100, 4096])))]
    bad = 0
    for t in range(cases):
        data = rng.choice(gens)()
        if not data:
            continue
        data = data[:4096]
        cut = rng.choice([0, 0, 1, 3, 15, 16, 17])
        if cut and len(data) > cut:
            data = data[:len(data) - cut]
        for key in ("rle8_single", "rle8_packed_single", "rle8_single_short"):
            want = ora.compress(CODEC_BY_KEY[key], data)
            got = model_short(data, want[8]) if key == "rle8_single_short" else model(data, want[9], key == "rle8_packed_single")
            if got != want:
                bad += 1
                if bad <= 5:
                    print("MISMATCH", key, "len", len(data), "case", t, "model", len(got), "oracle", len(want), flush=True)
                    with open(f"/tmp/single_model_bad_{bad}.bin", "wb") as f:
                        f.write(data)
    print("cases", cases, "mismatches", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
