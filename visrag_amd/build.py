"""Build libvisrag_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m visrag_amd.build            # incremental
    python -m visrag_amd.build --force
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libvisrag_hip.so")
SOURCES = ["gemm.hip", "gemm192.hip", "gemm32.hip", "gemm_ablate.hip", "gemm256p.hip", "gemm256t.hip", "gemm256w4.hip", "norm.hip", "attention.hip", "misc.hip", "search.hip", "search256.hip", "search_small.hip", "search_bigk.hip", "resize.hip", "pack.hip", "engine.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newest_header() -> float:
    t = 0.0
    for d in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(d):
            if f.endswith(".h"):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdr = _newest_header()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc, *FLAGS, "-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr[-4000:]}")
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return o

    if jobs:
        if verbose:
            print(f"[visrag_amd.build] compiling {len(jobs)} file(s) for gfx950", file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if jobs or not os.path.exists(LIB) or force:
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
