#!/usr/bin/env python3
"""Launches per call from a rocprofv3 --stats kernel_stats.csv:  python3 tools/kernel_count.py <kernel_stats.csv> <calls>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
tot = 0.0
for r in sorted(rows, key=lambda r: -int(r["Calls"])):
    per = int(r["Calls"]) / n
    tot += per
    if per >= 0.9:
        print(f"{per:7.2f} per call  avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:120]}")
print(f"{tot:7.2f} launches per call in all")
