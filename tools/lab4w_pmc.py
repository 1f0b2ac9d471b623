#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc pass of tools/gemm4w_lab (counter_collection.csv) -> JSON on stdout.
   python3 tools/lab4w_pmc.py <dir> [<dir> ...]"""
import csv, glob, json, os, sys
from collections import defaultdict
out = {}
for d in sys.argv[1:]:
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"]
                name = "4-wave asm" if "gemm4w_asm" in k else "production main loop" if "gemm_tn_x16_kernel" in k else None
                if name is None:
                    continue
                acc[name + " grid " + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[os.path.basename(d.rstrip("/"))] = {k: {c: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, sys.stdout, indent=1)
