"""Turn a rocprofv3 (rocpd sqlite) result into a text summary: per-kernel calls / total / avg,
and PMC counter sums per kernel if counters were collected.  Usage: prof_summary.py <db> [out]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    c = sqlite3.connect(db)
    rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    print("# rocprofv3 --kernel-trace summary (durations in us)", file=out)
    print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'%':>6}  kernel", file=out)
    for name, calls, total, avg, pct in rows[:40]:
        short = name if len(name) < 150 else name[:147] + "..."
        print(f"{calls:7d} {total:12.1f} {avg:10.2f} {pct:6.2f}  {short}", file=out)
    try:
        pm = c.execute(
            "select k.kernel_name, p.name, sum(e.value), count(*) from rocpd_pmc_event e "
            "join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id "
            "join rocpd_info_kernel_symbol k on d.kernel_id = k.id group by 1,2 order by 1,2").fetchall()
    except Exception as ex:  # schema differs between versions
        pm = []
        print(f"# (no PMC table: {ex})", file=out)
    if pm:
        print("\n# PMC counters: kernel | counter | sum over dispatches | dispatches", file=out)
        for k, n, v, cnt in pm:
            if "vr::" in k:
                print(f"{k[:90]:90s} {n:28s} {v:18.0f} {cnt:6d}", file=out)


if __name__ == "__main__":
    main()
