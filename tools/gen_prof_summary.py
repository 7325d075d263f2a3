"""Per-token summary of a rocprofv3 --kernel-trace run of tools/evisrag_bench.py (rocpd sqlite):
    python tools/gen_prof_summary.py <results.db> [out.txt]
Takes ten consecutive full decode steps (a step = everything between two sample_final_kernel launches) and prints wall time, kernel-busy time and the per-kernel shares per token, plus the
weight-streaming GEMM by grid size (its four per-layer shapes and the lm_head)."""
import sqlite3
import sys
from collections import defaultdict


def main():
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    c = sqlite3.connect(sys.argv[1])
    rows = c.execute("select name, start, end, grid_x from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if "sample_final" in r[0]]
    n = 10
    # a decode step = the launches between two sample_final kernels.  The run holds captured steps (the same launches every
    # token), host-driven steps and prefills; take the LAST window of n consecutive intervals that all have the most common
    # launch count among full decode steps (> 100 launches: whole-model steps, not the bench's short layers-only probes)
    counts = [idx[i + 1] - idx[i] for i in range(len(idx) - 1)]
    full = [c_ for c_ in counts if c_ > 100]
    mode = max(set(full), key=full.count)
    start = next(i for i in range(len(counts) - n, -1, -1) if all(c_ == mode for c_ in counts[i:i + n]))
    a, b = idx[start], idx[start + n]
    seg = rows[a:b + 1]
    wall = (seg[-1][2] - seg[0][2]) / 1e3
    busy = sum(r[2] - r[1] for r in seg[1:]) / 1e3
    print(f"# decode steps: {n} consecutive tokens (steps {start}..{start + n - 1} of {len(counts)} in the run), per token: wall {wall / n:.1f} us, "
          f"kernels busy {busy / n:.1f} us, {(len(seg) - 1) / n:.0f} launches", file=out)
    d = defaultdict(lambda: [0, 0.0])
    for r in seg[1:]:
        k = r[0] if len(r[0]) < 110 else r[0][:107] + "..."
        d[k][0] += 1
        d[k][1] += (r[2] - r[1]) / 1e3
    print(f"{'us/token':>10} {'calls/token':>12} {'avg_us':>8}  kernel", file=out)
    for k, v in sorted(d.items(), key=lambda x: -x[1][1]):
        print(f"{v[1] / n:10.1f} {v[0] / n:12.1f} {v[1] / v[0]:8.2f}  {k}", file=out)
    print("\n# vr::gemm_skinny_kernel by grid (workgroups = grid_x / 256)", file=out)
    g = defaultdict(lambda: [0, 0.0])
    for r in seg[1:]:
        if "gemm_skinny" in r[0]:
            g[r[3] // 256][0] += 1
            g[r[3] // 256][1] += (r[2] - r[1]) / 1e3
    for wg, v in sorted(g.items()):
        print(f"{wg:6d} workgroups: {v[0] / n:6.1f} calls/token, avg {v[1] / v[0]:7.2f} us", file=out)


main()
