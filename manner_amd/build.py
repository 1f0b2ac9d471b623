"""Build libmanner_hip.so (gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with
the gpurun snapshot.  ``python -m manner_amd.build`` or ``__graft_entry__.build()``.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "lib", "libmanner_hip.so")
SOURCES = ["gemm.hip", "rowops.hip", "attention.hip", "scoring.hip", "pool.hip", "entity.hip", "metrics.hip", "collate.hip", "cache.hip", "encoder.hip", "train.hip", "train_attn.hip", "wgrad.hip", "train_small.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build_library(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(PKG, "lib", "obj")
    os.makedirs(objdir, exist_ok=True)
    # the hand-scheduled GEMM K-loops are generated text (tools/gen_gemm_w.py), committed; a checkout without them regenerates
    if not all(os.path.exists(os.path.join(CSRC, f"gemm_w{nw}_asm.inc")) for nw in (8, 4)):
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_gemm_w.py")], check=True, capture_output=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "train_common.h"), os.path.join(CSRC, "gemm_w4_asm.inc"), os.path.join(CSRC, "gemm_w8_asm.inc"), os.path.join(ROOT, "include", "manner_hip.h")]
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, s + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < _newest([src] + headers):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return job, r

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        for (src, obj), r in ex.map(cc, jobs):
            if verbose and (r.stdout or r.stderr):
                sys.stderr.write(r.stdout + r.stderr)
            if r.returncode:
                raise RuntimeError(f"hipcc failed on {src}")
    objs = [os.path.join(objdir, s + ".o") for s in SOURCES]
    if jobs or not os.path.exists(LIB):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs,
                           capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link of libmanner_hip.so failed")
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
