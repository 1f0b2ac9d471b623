#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc passes (counter_collection.csv under each directory given) as a table: one row per kernel
(name shortened, template arguments kept), one column per counter, values per dispatch; with SQ_WAVE_CYCLES / SQ_BUSY_CYCLES present the
SQ counters are also shown as fractions.   python3 tools/pmc_kernels.py <dir> [<dir> ...] [--match substring]"""
import csv, glob, os, re, sys
from collections import defaultdict
dirs, match = [], None
a = sys.argv[1:]
while a:
    x = a.pop(0)
    if x == "--match":
        match = a.pop(0)
    else:
        dirs.append(x)
acc = defaultdict(lambda: defaultdict(list))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"]
                if match and match not in k:
                    continue
                m = re.search(r"(gemm_tn_\w+_kernel)<([^>]*)>", k) or re.search(r"(gemm_tn_\w+?_kernel)I(\w+?)EEv", k)
                name = f"{m.group(1)}<{m.group(2)}>" if m else k[:60]
                acc[name + " grid " + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(acc.items()):
    print(name)
    ref = None
    for c in ("SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES"):
        if c in cs:
            ref = (c, sum(cs[c]) / len(cs[c]))
    for c, v in sorted(cs.items()):
        mean = sum(v) / len(v)
        frac = f"   {mean / ref[1]:8.4f} of {ref[0]}" if ref and c.startswith("SQ_") and ref[1] else ""
        print(f"    {c:34s} {mean:16.1f}  ({len(v)} dispatches){frac}")
