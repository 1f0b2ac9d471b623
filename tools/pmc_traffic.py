#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<round>/pmc_traffic.json.

  python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> > profiles/r1_final/pmc_traffic.json

Each pass: rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -d <dir> -- python3 bench.py --steps 1
--warmup 1 --no-cpu --no-kernel-profile --no-table --no-collate.  Units and the gfx950 correction follow
MI355X_MICROARCH.md: the counters are in KiB and FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at
64 bytes, so HBM-side bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.  Per class the FULL 65536-token launches are
kept (top quartile by bytes; the rest are the ragged last chunk of a step and the last layer's K|V-only GEMM).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

CLASSES = [   # (class, substrings that must all occur in the kernel name)
    ("gemm_qkv", ["gemm_tn_x16_kernel", "Li3ELi0E"]),        # EPI_NORM
    ("gemm_ffn1", ["gemm_tn_x16_kernel", "Li4ELi0E"]),       # EPI_NORM_GELU
    ("gemm_out_ffn2", ["gemm_tn_x16_kernel", "Li5ELi0E"]),   # EPI_NRES: out-proj and FFN2 share the instantiation
    ("attention", ["attn_bf16_kernel"]),
    ("dln_finalize", ["dln_finalize_kernel"]),
    ("embed_raw", ["embed_raw_kernel"]),
    ("score_late_fusion", ["score_late_fusion_kernel"]),
    ("rank_ndcg", ["rank_ndcg_kernel"]),
]
# rocprofv3 prints some instantiations half-demangled; EPI_NRES appears as "<bool _Accum, int, ELi0E>"
# (round 5: the kernel has a fifth template argument — `, 3, 0, true>` / `Li3ELi0ELb1E`; the mangled substrings still match)
TEMPLATE_HINTS = {"Li3ELi0E": (", 3, 0>", ", 3, 0, true>"), "Li4ELi0E": (", 4, 0>", ", 4, 0, true>"), "Li5ELi0E": ("int, ELi0E>", ", 5, 0, true>", "int, ELi0ELb1E>", ", 5, 0>")}


def classify(name):
    # round 6: the hand-scheduled kernels gemm_tn_w8_kernel / gemm_tn_w4_kernel take the 256-row launches of the same three classes
    # (same EPI template argument; w8 carries the lab's ABL argument too: <.., EPI, 0>, w4 ends at <.., EPI>)
    name_x = name.replace("gemm_tn_w8_kernel", "gemm_tn_x16_kernel")
    if "gemm_tn_w4_kernel" in name_x:
        name_x = name_x.replace("gemm_tn_w4_kernel", "gemm_tn_x16_kernel")
        for epi in (3, 4, 5):
            name_x = name_x.replace(f"Li{epi}EEE", f"Li{epi}ELi0EEE").replace(f", {epi}>", f", {epi}, 0>")
    for cls, subs in CLASSES:
        if all(s in name_x or any(h in name_x for h in TEMPLATE_HINTS.get(s, ())) for s in subs):
            return cls
    return None


def read_pass(directory, counter):
    rows = defaultdict(list)
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") != counter:
                    continue
                cls = classify(r.get("Kernel_Name", ""))
                if cls:
                    rows[cls].append((float(r["Counter_Value"]), r["Kernel_Name"]))
    return rows


def main():
    fetch, write = read_pass(sys.argv[1], "FETCH_SIZE"), read_pass(sys.argv[2], "WRITE_SIZE")
    out = {"_note": __doc__.strip().split("\n\n")[-1].replace("\n", " ")}
    for cls, _ in CLASSES:
        if cls not in fetch or cls not in write:
            continue
        def top(vals):
            v = sorted(x for x, _ in vals)
            q = v[3 * len(v) // 4:] or v
            return sum(q) / len(q)
        f_kib, w_kib = top(fetch[cls]), top(write[cls])
        ent = {"kernel": fetch[cls][0][1][:96], "launches": len(fetch[cls]), "FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib,
               "hbm_bytes_per_launch": 2 * f_kib * 1024 + w_kib * 1024}
        if cls == "gemm_out_ffn2":
            # the two GEMMs differ in what they read (ctx 100 MB vs the FFN intermediate 403 MB): split by the median
            v = sorted(x for x, _ in fetch[cls])
            mid = v[len(v) // 2]
            lo = [x for x in v if x <= mid]
            hi = [x for x in v if x > mid]
            for key, part in (("gemm_out", lo), ("gemm_ffn2", hi)):
                if part:
                    q = part[3 * len(part) // 4:] or part
                    fk = sum(q) / len(q)
                    out[key] = {"kernel": ent["kernel"], "launches": len(part), "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": w_kib,
                                "hbm_bytes_per_launch": 2 * fk * 1024 + w_kib * 1024}
        out[cls] = ent
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
