#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/c3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pl
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_WAVE_CYCLES --kernel-trace -d /tmp/pl -o p -- $R/tools/probe_lds > /tmp/pl.log 2>&1
python - > $O/probe_lds.txt <<'PY'
import sqlite3, glob
c = sqlite3.connect(glob.glob('/tmp/pl/**/*.db', recursive=True)[0])
rows = c.execute("select kernel_name, counter_name, sum(value) from counters_collection group by 1,2 order by 1,2").fetchall()
d = {}
for k, cn, v in rows: d.setdefault(k, {})[cn] = v
for k, v in d.items():
    print(k[:40], {a: int(b) for a, b in v.items()})
PY
cat $O/probe_lds.txt
