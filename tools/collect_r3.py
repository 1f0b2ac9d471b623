#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/profile_r3.sh (gpurun_out/prof_r3) into the committed summaries under
profiles/r3_final/ (python tools/collect_r3.py [raw dir] [summary dir]):

  kernel_stats.csv   rocprofv3 --kernel-trace --stats table of `bench.py --steps 3` (single stream)
  pmc_traffic.json   HBM-side bytes per launch of the encoder kernels (tools/pmc_traffic.py: 2*FETCH_SIZE + WRITE_SIZE, KiB)
  sq_counters.csv    SQ counters per kernel class (full-size launches), with the fractions of SQ_WAVE_CYCLES
  tail_pmc.json      scorer on the 495 MB MIND-large table + pooler / dot / z-score / to_dense: durations, counter bytes,
                     algorithmic bytes (tools/tail_probe.py)
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof_r3")
DST = os.path.abspath(sys.argv[2]) if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r3_final")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pmc_traffic  # noqa: E402


def one(pattern):
    hits = sorted(glob.glob(os.path.join(SRC, pattern), recursive=True))
    return hits[0] if hits else None


def counters(directory):
    """{kernel name: {counter: [values per dispatch]}}"""
    out = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(SRC, directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def main():
    os.makedirs(DST, exist_ok=True)
    st = one("stats/**/*kernel_stats.csv")
    if st:
        shutil.copy(st, os.path.join(DST, "kernel_stats.csv"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(SRC, "fetch"), os.path.join(SRC, "write")],
                       capture_output=True, text=True)
    if r.returncode == 0 and r.stdout.strip():
        with open(os.path.join(DST, "pmc_traffic.json"), "w") as f:
            f.write(r.stdout)
    else:
        print("pmc_traffic failed:", r.stderr[-500:])
    # ---- SQ counters per class: the full-size launches = top quartile by SQ_WAVE_CYCLES
    sq = counters("sq")
    names = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
             "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT"]
    per_class = defaultdict(lambda: defaultdict(list))
    for kname, cs in sq.items():
        cls = pmc_traffic.classify(kname)
        if not cls or "SQ_WAVE_CYCLES" not in cs:
            continue
        n = len(cs["SQ_WAVE_CYCLES"])
        order = sorted(range(n), key=lambda i: cs["SQ_WAVE_CYCLES"][i])
        keep = order[3 * n // 4:] or order
        for c in names:
            if c in cs and len(cs[c]) == n:
                per_class[cls][c] += [cs[c][i] for i in keep]
        per_class[cls]["_kernel"] = [kname[:100]]
    with open(os.path.join(DST, "sq_counters.csv") if per_class else os.devnull, "w", newline="") as f:   # a tail-only session keeps the old file
        w = csv.writer(f)
        w.writerow(["class", "kernel", "launches_kept"] + names + ["wait_any_frac", "wait_inst_any_frac", "active_inst_any_frac",
                                                                   "mfma_busy_cycles_per_wave_quadcycle", "lds_conflict_per_lds_inst"])
        for cls, cs in sorted(per_class.items()):
            mean = {c: (sum(cs[c]) / len(cs[c]) if cs.get(c) else float("nan")) for c in names}
            wc = mean["SQ_WAVE_CYCLES"]
            w.writerow([cls, cs["_kernel"][0], len(cs["SQ_WAVE_CYCLES"])] + [f"{mean[c]:.0f}" for c in names] +
                       [f"{mean['SQ_WAIT_ANY'] / wc:.3f}", f"{mean['SQ_WAIT_INST_ANY'] / wc:.3f}", f"{mean['SQ_ACTIVE_INST_ANY'] / wc:.3f}",
                        f"{mean['SQ_VALU_MFMA_BUSY_CYCLES'] / wc:.3f}",
                        f"{mean['SQ_LDS_BANK_CONFLICT'] / max(mean['SQ_INSTS_LDS'], 1):.3f}"])
    # ---- tail kernels
    tp = os.path.join(SRC, "tail_probe.json")
    if os.path.exists(tp):
        with open(tp) as f:
            line = [l for l in f.read().splitlines() if l.startswith("{")][-1]
        probe = json.loads(line)
        fetch, write = counters("tail_fetch"), counters("tail_write")
        dur = {}
        ts = one("tail_stats/**/*kernel_stats.csv")
        if ts:
            with open(ts, newline="") as f:
                for r in csv.DictReader(f):
                    dur[r["Name"]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
        kmap = {"score_late_fusion": ["score_late_fusion_rows_kernel"], "score_late_fusion_f16": ["score_late_fusion_f16_rows_kernel"],
                "additive_pool": ["pool_fused_kernel", "pool_pack_w_kernel", "pool_w_max_kernel"] if "additive_pool_strict" in probe else ["pool_logits_kernel", "pool_apply_kernel"],
                "additive_pool_strict": ["pool_logits_kernel", "pool_apply_kernel"],
                "dot": ["dot_rows_kernel"], "zscore_fuse": ["zscore_fuse_kernel"], "to_dense": ["to_dense_rows_kernel"],
                "phase_c_one_launch": ["score_fuse_rank_kernel"]}
        out = {"_note": "bytes: 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 per launch (KiB counters, gfx950 64-byte correction of wide reads, "
                        "MI355X_MICROARCH.md); durations from the --stats pass of the same script; algorithmic bytes from tools/tail_probe.py"}
        for op, kernels in kmap.items():
            ent = dict(probe.get(op, {}))
            tot_bytes, tot_us = 0.0, 0.0
            if op not in probe:
                continue
            for k in kernels:
                match = lambda n: k in n   # noqa: E731
                fk = [v for n, v in fetch.items() if match(n)]
                wk = [v for n, v in write.items() if match(n)]
                dk = [v for n, v in dur.items() if match(n)]
                if not fk or not wk:
                    continue
                fv, wv = fk[0]["FETCH_SIZE"], wk[0]["WRITE_SIZE"]
                b = 2 * 1024 * sum(fv) / len(fv) + 1024 * sum(wv) / len(wv)
                ent[k] = {"FETCH_SIZE_KiB": sum(fv) / len(fv), "WRITE_SIZE_KiB": sum(wv) / len(wv), "hbm_bytes_per_launch": b,
                          "avg_us": dk[0]["avg_us"] if dk else None}
                tot_bytes += b
                tot_us += dk[0]["avg_us"] if dk else 0.0
            if tot_us:
                ent["hbm_bytes_per_call"] = tot_bytes
                ent["us_per_call"] = tot_us
                ent["hbm_GBps"] = tot_bytes / tot_us / 1e3
                ent["hbm_frac_of_8TBps"] = tot_bytes / tot_us / 1e3 / 8000.0
                if "algorithmic_bytes_per_launch" in ent:
                    ent["algorithmic_GBps"] = ent["algorithmic_bytes_per_launch"] / tot_us / 1e3
                    ent["traffic_over_algorithmic"] = tot_bytes / ent["algorithmic_bytes_per_launch"]
            out[op] = ent
        with open(os.path.join(DST, "tail_pmc.json"), "w") as f:
            json.dump(out, f, indent=1)
    # ---- A/B of chunk size x store policy (tools/ab_r3.sh times + power, tools/ab_pmc_r3.sh counters)
    ab = {}
    for path in sorted(glob.glob(os.path.join(SRC, "ab_pmc_*.json"))):
        tag = os.path.basename(path)[len("ab_pmc_"):-5]
        try:
            with open(path) as f:
                tj = json.load(f)
        except Exception:
            continue
        ab[tag] = {k: {"FETCH_SIZE_KiB": v["FETCH_SIZE_KiB"], "WRITE_SIZE_KiB": v["WRITE_SIZE_KiB"], "hbm_bytes_per_launch": v["hbm_bytes_per_launch"],
                       "launches": v["launches"]} for k, v in tj.items() if isinstance(v, dict) and k.startswith(("gemm_", "attention"))}
    r3 = os.path.join(ROOT, "gpurun_out", "r3")
    for path in (sorted(glob.glob(os.path.join(r3, "ab_ct*_nt*.json"))) if ab else []):      # times only next to counters of the SAME session
        tag = os.path.basename(path)[3:-5]
        try:
            with open(path) as f:
                j = json.loads(f.read().strip().splitlines()[-1])
        except Exception:
            continue
        ent = ab.setdefault(tag, {})
        ent["candidates_per_s_f16"] = j["value"]
        ent["encoder_mfma_frac_f16"] = j["encoder_mfma_frac"]
        ent["candidates_per_s_bf16"] = j["bf16_mode"]["value"]
        ent["encoder_mfma_frac_bf16"] = j["bf16_mode"]["encoder_mfma_frac"]
        smi = path[:-5] + ".smi"
        if os.path.exists(smi):
            import re
            txt = open(smi).read()
            pw = [float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)", txt)]
            ck = [float(x) for x in re.findall(r"sclk.*?\(([0-9]+)Mhz\)", txt)]
            if pw:
                ent["socket_power_W_samples"] = {"n": len(pw), "mean": sum(pw) / len(pw), "max": max(pw)}
            if ck:
                ent["sclk_MHz_mean"] = sum(ck) / len(ck)
    if ab:
        ab["_note"] = ("tag ct<chunk tokens>_nt<1: non-temporal stores of the streaming Q|K|V / FFN outputs (production), 0: default-policy stores>; "
                       "times + rocm-smi samples from tools/ab_r3.sh (10 timed steps, both streams), counters from tools/ab_pmc_r3.sh (1 step, one stream; "
                       "bytes per launch as in pmc_traffic.json — for chunks below 65536 tokens a launch is proportionally smaller)")
        with open(os.path.join(DST, "ab_chunk_nt.json"), "w") as f:
            json.dump(ab, f, indent=1)
    tr = one("train/**/*kernel_stats.csv")
    if tr:
        shutil.copy(tr, os.path.join(DST, "train_kernel_stats.csv"))
    print("wrote", sorted(os.listdir(DST)))


if __name__ == "__main__":
    main()
