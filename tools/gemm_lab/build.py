"""Build the GEMM experiment kernels of round 1 (measured-negative variants, main-loop ablations)
into tools/gemm_lab/libvisrag_gemm_lab.so — NOT part of the product library.

    python tools/gemm_lab/build.py

Entry point: vr_lab_gemm(...) with the arguments of vr_op_gemm; variants 1, 2, 4, 5, 6, 8, 10, 11,
12 and the timing-only ablations 20..29 (results of those are NOT valid products).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "visrag_amd", "csrc")
SOURCES = ["lab_gemm.hip", "gemm32.hip", "gemm_ablate.hip", "gemm256p.hip", "gemm256t.hip", "gemm256w4.hip"]


def build() -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    lib = os.path.join(HERE, "libvisrag_gemm_lab.so")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", HERE, "-I", CSRC,
           *[os.path.join(HERE, s) for s in SOURCES], os.path.join(CSRC, "gemm192.hip"), "-o", lib]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-6000:])
    return lib


if __name__ == "__main__":
    print(build())
