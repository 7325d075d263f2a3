"""Build libvisrag_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m visrag_amd.build                      # incremental
    python -m visrag_amd.build --force
    python -m visrag_amd.build --tag p1 -DVR_ATTN_PIPE=1   # A/B build: libvisrag_hip_p1.so (tools/ only)

Tagged builds exist for kernel A/B measurements (tools/ab_*.py load them by path); the product
and the tests always use the untagged library.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from typing import Iterable, List, Optional

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["gemm.hip", "gemm192.hip", "gemm256w.hip", "gemm128w.hip", "gemm_skinny.hip", "gen_persist.hip", "norm.hip", "attention.hip", "attention_w.hip", "attention_small.hip", "patch_embed.hip", "misc.hip", "search.hip", "search256.hip", "search256w.hip",
           "search_small.hip", "search_bigk.hip", "search_exact.hip", "search_band.hip", "hp_text.hip", "resize.hip", "synth.hip", "pack.hip", "engine.hip", "gen_kernels.hip", "gen.hip", "gen_vision.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# attention.hip: relaxed NaN handling only (infinities are honoured: masked scores are -inf).  Without it
# every fmaxf of an MFMA result is preceded by a canonicalising v_max_f32 x, x (32 extra VALU per tile);
# same for the clamp of the GELU epilogue (gemm*.hip: one v_max per output value).
FILE_FLAGS = {"attention.hip": ["-fno-honor-nans"], "attention_w.hip": ["-fno-honor-nans"], "attention_small.hip": ["-fno-honor-nans"], "gemm.hip": ["-fno-honor-nans"], "gemm192.hip": ["-fno-honor-nans"],
              "gemm256w.hip": ["-fno-honor-nans"], "gemm128w.hip": ["-fno-honor-nans"], "gen_persist.hip": ["-fno-honor-nans"]}
# gemm256w.hip hand-allocates the accumulation registers inside asm statements; hipcc only sees them as clobbers,
# so if it ever runs out of VGPRs there it parks the overflow in registers that hold results.  The build checks
# the generated code: outside the kernel's own asm there must be no accumulation-register traffic at all.
AGPR_CHECKED = {"gemm256w.hip", "gemm128w.hip", "search256w.hip", "attention_w.hip"}
# ... and whose score MFMAs are asm statements on arch VGPRs: the listing is also walked for operand hazards (mfma_operand_hazards)
MFMA_HAZARD_CHECKED = {"attention_w.hip"}


def lib_path(tag: str = "") -> str:
    return os.path.join(HERE, f"libvisrag_hip{'_' + tag if tag else ''}.so")


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


_AGPR_OPERAND = re.compile(r"(?<![\w.$])a(\d+|\[\d+:\d+\])(?![\w])")


def agpr_violations(asm_text: str, owned: Optional[range] = None) -> List[str]:
    """Instructions of a `hipcc -S` listing that touch accumulation registers outside `;;#ASMSTART ... ;;#ASMEND`:
    `v_accvgpr_*` moves AND any other instruction with an a-register operand (on gfx950 the compiler may address them
    directly: `ds_read_b128 a[0:3], ...`, `global_load_dwordx4 a[..]`, `scratch_store_dword ..., a5`, an MFMA with
    `a[..]` as source or destination).  Also reports VGPR spills of the file's kernels (`.vgpr_spill_count`)."""
    bad, in_asm = [], False
    for ln in asm_text.splitlines():
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        elif not in_asm and t and not t.startswith((".", ";", "//")) and not t.endswith(":"):
            code = t.split(";", 1)[0]
            parts = code.split(None, 1)
            if parts[0].startswith("v_accvgpr") or (len(parts) > 1 and _AGPR_OPERAND.search(parts[1])):
                # `owned`: the hand-allocated range of the file — hipcc parking a value of its own in an accumulation register
                # OUTSIDE it (a spill slot under a two-waves-per-SIMD budget) is its business
                if owned is None or any(k in owned for kind, k in _regs(parts[1] if len(parts) > 1 else "") if kind == "a"):
                    bad.append(code.strip())
        elif not in_asm and t.startswith(".vgpr_spill_count:"):
            if int(t.split(":", 1)[1]) != 0:
                bad.append(t + "   (a spilled VGPR next to hand-allocated accumulators)")
    return bad


_REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def _regs(operand_text: str):
    out = set()
    for m in _REG.finditer(operand_text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), k) for k in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def mfma_operand_hazards(asm_text: str, raw_states: int = 2, result_states: int = 12) -> List[str]:
    """Files whose MFMAs are asm statements (hipcc neither pads nor orders around them): walk the listing and report
      * a VALU / accvgpr write of a register that an MFMA within the next `raw_states` wait states reads (A, B or C), and
      * any non-MFMA instruction that touches an MFMA's VGPR / AGPR result within `result_states` wait states of its issue
        (an MFMA taking the whole result as its C is the accumulate chain: fine).
    Wait states are counted conservatively: one per instruction, n + 1 for `s_nop n`, 4 for an MFMA (its issue passes)."""
    bad: List[str] = []
    recent_writes: List[tuple] = []          # (states ago, regs, text) of VALU writes
    pending: List[list] = []                 # [states left, result regs, text] of MFMAs
    for ln in asm_text.splitlines():
        t = ln.strip()
        if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
            if t.endswith(":") and not t.startswith(";"):
                recent_writes, pending = [], []          # (a label: another path may join — hazards across it are not modelled)
            continue
        code = t.split(";", 1)[0].strip()
        if not code:
            continue
        parts = code.split(None, 1)
        op, args = parts[0], (parts[1] if len(parts) > 1 else "")
        states = 1
        if op == "s_nop":
            states = int(args.strip() or "0", 0) + 1
        ops = [a.strip() for a in args.split(",")]
        if op.startswith("v_mfma"):
            dst, srcs = _regs(ops[0]), [_regs(o) for o in ops[1:4]]
            for ago, regs, text in recent_writes:
                if ago < raw_states and any(regs & s_ for s_ in srcs):
                    bad.append(f"`{text}` {ago} wait state(s) in front of `{code}`")
            for pnd in pending:
                whole_c = len(srcs) == 3 and srcs[2] == pnd[1]
                if pnd[0] > 0 and not whole_c and (pnd[1] & (dst | srcs[0] | srcs[1] | (srcs[2] if len(srcs) == 3 else set()))):
                    bad.append(f"`{code}` overlaps the result of `{pnd[2]}` {result_states - pnd[0]} wait state(s) behind it")
            states = 4
            new_pending = [result_states, dst, code]
        else:
            new_pending = None
            touched = _regs(args)
            for pnd in pending:
                if pnd[0] > 0 and (pnd[1] & touched):
                    bad.append(f"`{code}` touches the result of `{pnd[2]}` {result_states - pnd[0]} wait state(s) behind it")
        if op.startswith("v_") and not op.startswith("v_mfma") and not op.startswith("v_cmp") and ops:
            recent_writes.append((0, _regs(ops[0]), code))
        recent_writes = [(a + states, r, x) for a, r, x in recent_writes if a + states < raw_states + 2]
        for pnd in pending:
            pnd[0] -= states
        pending = [p_ for p_ in pending if p_[0] > 0]
        if new_pending:
            pending.append(new_pending)
    return bad


def _check_no_compiler_agprs(hipcc: str, src: str, flags: List[str]) -> None:
    r = subprocess.run([hipcc, *[f for f in flags if f != "-fPIC"], "--cuda-device-only", "-S", src, "-o", "-"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc -S failed for {src}:\n{r.stderr[-2000:]}")
    if os.path.basename(src) in MFMA_HAZARD_CHECKED:
        hz = mfma_operand_hazards(r.stdout)
        if hz:
            raise RuntimeError(f"{os.path.basename(src)}: {len(hz)} MFMA operand hazard(s) in the generated code, first: {hz[0]}")
    owned = None
    if os.path.basename(src) == "attention_w.hip":          # O^T: a[0:79] with one wave per SIMD (the default), a[0:39] with two
        owned = range(0, 40) if "-DVR_ATTN_W_WAVES=8" in flags else range(0, 80)
    bad = agpr_violations(r.stdout, owned)
    if bad:
        raise RuntimeError(f"{os.path.basename(src)}: hipcc generated `{bad[0]}` (+{len(bad) - 1} more) outside the hand-written "
                           "asm — the accumulation registers are not the compiler's to use in this file (lower the VGPR "
                           "pressure of the code around the K-loop)")


def build(force: bool = False, verbose: bool = True, tag: str = "", defines: Optional[Iterable[str]] = None) -> str:
    defines = list(defines or [])
    obj_dir = os.path.join(HERE, "build" + ("_" + tag if tag else ""))
    lib = lib_path(tag)
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    hdr = _newest_header()
    stamp = os.path.join(obj_dir, "defines.txt")
    if tag:     # a changed define set invalidates a tagged build
        old = open(stamp).read() if os.path.exists(stamp) else None
        if old != " ".join(defines):
            force = True
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr):
            jobs.append((s, o, src))

    def compile_one(job):
        s, o, name = job
        cmd = [hipcc, *FLAGS, *FILE_FLAGS.get(name, []), *defines, "-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr[-4000:]}")
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        if name in AGPR_CHECKED:
            _check_no_compiler_agprs(hipcc, s, [*FLAGS, *FILE_FLAGS.get(name, []), *defines])
        return o

    if jobs:
        if verbose:
            print(f"[visrag_amd.build] compiling {len(jobs)} file(s) for gfx950" + (f" (tag {tag})" if tag else ""),
                  file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs: List[str] = [os.path.join(obj_dir, s.replace(".hip", ".o")) for s in SOURCES]
    if jobs or not os.path.exists(lib) or force:
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    if tag:
        with open(stamp, "w") as f:
            f.write(" ".join(defines))
    return lib


if __name__ == "__main__":
    argv = sys.argv[1:]
    tag = argv[argv.index("--tag") + 1] if "--tag" in argv else ""
    print(build(force="--force" in argv, tag=tag, defines=[a for a in argv if a.startswith("-D")]))
