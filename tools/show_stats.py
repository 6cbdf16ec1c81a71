#!/usr/bin/env python3
"""Print the head of a rocprofv3 kernel_stats.csv: python tools/show_stats.py FILE [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:10.1f} us  total {float(r['TotalDurationNs']) / 1e6:9.3f} ms")
