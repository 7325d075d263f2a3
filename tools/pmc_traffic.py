"""Fold the FETCH_SIZE / WRITE_SIZE rocprofv3 passes (rocpd sqlite) into per-launch HBM bytes.
Units: the counters report KB; gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE
counts 64-B requests as 32 B for wide coalesced reads -> x2 (checked on norm_kernel, a pure
streaming kernel whose traffic is known exactly).  Usage: pmc_traffic.py <fetch_dir> <write_dir> <out_dir>"""
import glob
import json
import os
import sqlite3
import sys


def per_kernel(d):
    out = {}
    for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        c = sqlite3.connect(db)
        try:
            rows = c.execute("select kernel_name, sum(value), count(distinct dispatch_id) from counters_collection "
                             "where counter_name in ('FETCH_SIZE', 'WRITE_SIZE') group by 1").fetchall()
        except Exception as ex:
            print("skip", db, str(ex)[:200])
            continue
        for name, total, n in rows:
            a = out.setdefault(name, [0.0, 0])
            a[0] += total; a[1] += n
    return {k: v[0] / max(v[1], 1) for k, v in out.items()}


def main():
    f, w, outd = per_kernel(sys.argv[1]), per_kernel(sys.argv[2]), sys.argv[3]
    os.makedirs(outd, exist_ok=True)
    rows = sorted(((k, f.get(k, 0.0), w.get(k, 0.0)) for k in set(f) | set(w) if "vr::" in k), key=lambda r: -(2 * r[1] + r[2]))
    traffic = {k: (2.0 * fk + wk) * 1024.0 for k, fk, wk in rows}
    with open(os.path.join(outd, "table.txt"), "w") as t:
        t.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/encode_only.py + tools/search_bench.py 1000, per launch\n")
        t.write("# units: KB as reported; gfx950 correction: FETCH_SIZE x2 for wide coalesced reads (MI355X_MICROARCH.md, HBM)\n")
        t.write(f"{'kernel':90s} {'FETCH_KB':>12s} {'WRITE_KB':>12s} {'HBM bytes (2*F+W)':>20s}\n")
        for k, fk, wk in rows:
            t.write(f"{k[:90]:90s} {fk:12.0f} {wk:12.0f} {traffic[k]:20.4e}\n")
    json.dump(traffic, open(os.path.join(outd, "traffic.json"), "w"), indent=1)
    print(open(os.path.join(outd, "table.txt")).read())


if __name__ == "__main__":
    main()
