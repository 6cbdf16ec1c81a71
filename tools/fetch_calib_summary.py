#!/usr/bin/env python3
"""Summarise gpurun_out/fetch_calib/: per replay mode, the counters per kernel launch and raw FETCH_SIZE / bytes read."""
import csv, glob, os, sys

root = sys.argv[1]
names = {0: "whole lines (8 chunks / instruction)", 1: "half lines (4 + 4)", 2: "single chunks (8 x 1)", 3: "decoder mix (random 1..3 parts)"}
for mode in range(4):
    line = open(os.path.join(root, f"plain_{mode}.txt")).read().strip()
    nbytes = int(line.split("bytes")[1].split()[0])
    vals = {}
    for f in glob.glob(os.path.join(root, f"m{mode}_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_replay" in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print(f"mode {mode}: {names[mode]}   [{line}]")
    for k, v in sorted(vals.items()):
        avg = sum(v) / len(v)
        extra = ""
        if k == "FETCH_SIZE":
            extra = f"  = {avg * 1024:.0f} bytes raw  -> raw / bytes read = {avg * 1024 / nbytes:.4f}  (bytes read / raw = {nbytes / (avg * 1024):.4f})"
        if k.startswith("TCC_EA0_RDREQ") or k.startswith("TCC_BUBBLE"):
            extra = f"  -> bytes read per request = {nbytes / avg:.1f}" if avg else ""
        print(f"    {k:28s} {avg:16.1f} per launch ({len(v)} launches){extra}")

# the request model, checked against the measured request counts:  a load instruction that asks for all 8 chunks of a line -> ONE 128-byte
# request; otherwise ONE 64-byte request per half of the line it touches.  FETCH_SIZE tallies every request at 64 bytes.
import numpy as np

def mix(a):
    a = a.astype(np.uint32)
    a ^= a >> np.uint32(16); a = a * np.uint32(0x7feb352d); a ^= a >> np.uint32(15); a = a * np.uint32(0x846ca68b); a ^= a >> np.uint32(16)
    return a

line0 = open(os.path.join(root, "plain_3.txt")).read()
rows = int(line0.split("rows")[1].split()[0])
with np.errstate(over="ignore"):
    r = np.arange(rows, dtype=np.uint32)[:, None] * np.uint32(31) + np.arange(18, dtype=np.uint32)[None, :]
    h = mix(r)
a, b = h & 7, (h >> 3) & 7
lo, hi = np.minimum(a, b), np.maximum(a, b)
def halves(x0, x1):   # requests of the part [x0, x1) of a line's 8 chunks
    n = x1 - x0
    full = (n == 8)
    touched = ((x0 < 4) & (x1 > 0) & (n > 0)).astype(np.int64) + ((x1 > 4) & (n > 0)).astype(np.int64)
    return np.where(full, 0, touched), full.astype(np.int64)
n64 = n128 = 0
for x0, x1 in ((np.zeros_like(lo), lo), (lo, hi), (hi, np.full_like(hi, 8))):
    t, f = halves(x0.astype(np.int64), x1.astype(np.int64))
    n64 += int(t.sum()); n128 += int(f.sum())
print(f"model for mode 3: {n128} whole-line + {n64} half-line requests = {n128 + n64} requests, {128 * n128 + 64 * n64} bytes fetched, FETCH_SIZE would read {64 * (n128 + n64)}")
print(f"model for mode 0: {rows * 18} requests;  mode 1: {rows * 36} requests;  mode 2: {rows * 144} requests")
