#!/usr/bin/env python3
"""Busy time and gaps of a rocprofv3 --kernel-trace CSV (development aid, round 5).

    python tools/kernel_timeline.py <..._kernel_trace.csv> [gap_split_us]

Kernels are sorted by start; a gap longer than `gap_split_us` (default 1500) starts a new "burst" (a batch of the drop-in probe ends with a
host synchronisation, so a burst is one step or one setup phase).  Per burst: wall (first start to last end), busy (union of kernel
intervals), launches, the gaps by size class.  Printed: the median burst among the LAST third (the timed steps) and its top kernels."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
split = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 1500e3
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                     r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")))
rows.sort()
bursts, cur = [], []
last_end = None
for s, e, n in rows:
    if last_end is not None and s - last_end > split:
        bursts.append(cur); cur = []
    cur.append((s, e, n))
    last_end = e if last_end is None else max(last_end, e)
if cur:
    bursts.append(cur)


def stats(b):
    wall = max(e for _, e, _ in b) - b[0][0]
    busy, gaps, end = 0, [], b[0][0]
    for s, e, _ in b:
        if s > end:
            gaps.append(s - end)
            busy += e - s
            end = e
        elif e > end:
            busy += e - end
            end = e
    cls = {"<2us": 0, "2-10us": 0, "10-50us": 0, "50-300us": 0, ">300us": 0}
    for g in gaps:
        k = "<2us" if g < 2e3 else "2-10us" if g < 1e4 else "10-50us" if g < 5e4 else "50-300us" if g < 3e5 else ">300us"
        cls[k] += g
    return wall, busy, len(b), {k: round(v / 1e3, 1) for k, v in cls.items()}


big = [b for b in bursts if len(b) > 50]
print(f"{len(rows)} launches, {len(bursts)} bursts, {len(big)} with > 50 launches")
tail = big[-max(1, len(big) // 3):]
tail.sort(key=lambda b: stats(b)[0])
b = tail[len(tail) // 2]
wall, busy, n, cls = stats(b)
print(f"median timed burst: wall {wall / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {(wall - busy) / 1e3:.1f} us, {n} launches; idle by gap size (us): {cls}")
tot = defaultdict(lambda: [0, 0])
for s, e, nme in b:
    k = nme.split("(")[0][:110]
    tot[k][0] += e - s; tot[k][1] += 1
for k, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:28]:
    print(f"{t / 1e3:9.1f} us {c:4d}x  {k}")
# the long gaps of that burst, with the kernels on either side
end, prev = b[0][0], None
for s, e, nme in b:
    if s - end > 4e4:
        print(f"gap {(s - end) / 1e3:7.1f} us  after {prev.split('(')[0][:70] if prev else None}  before {nme.split('(')[0][:70]}")
    if e > end:
        end, prev = e, nme
