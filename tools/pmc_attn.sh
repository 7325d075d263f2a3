#!/bin/bash
# Where do the ViT attention kernel's cycles go?  SQ counters (own pass, kernel trace only) on the attention microbench.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pm_a
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d /tmp/pm_a -o a -- python $R/tools/gemm_only.py attn > /tmp/pm_a.log 2>&1
tail -2 /tmp/pm_a.log | cut -c1-200
python - <<PY
import sqlite3, glob
db = glob.glob('/tmp/pm_a/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
for k, cn, v, n in c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection where kernel_name like '%attention%' group by 1,2"):
    print(k[:50], cn, round(v / n))
PY
