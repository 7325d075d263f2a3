#!/bin/bash
# MFMA utilisation per kernel (rocprofv3 --pmc, own pass, kernel trace only) over one encode step
# and one 1k x 100k search.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc; mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -i -o "MfmaUtil\|SQ_VALU_MFMA_BUSY_CYCLES\|SQ_BUSY_CU_CYCLES\|SQ_INSTS_VALU_MFMA_MOPS_BF16\|GRBM_GUI_ACTIVE" | sort | uniq -c > $O/avail.txt
cat $O/avail.txt
rm -rf /tmp/pm_m
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace -d /tmp/pm_m -o m -- python $R/tools/encode_only.py 1 > /tmp/pm_m.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace -d /tmp/pm_m -o s -- python $R/tools/search_bench.py 1000 > /tmp/pm_s.log 2>&1
tail -2 /tmp/pm_m.log /tmp/pm_s.log
python - <<PY
import sqlite3, glob
agg = {}
for db in sorted(glob.glob('/tmp/pm_m/**/*.db', recursive=True)):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id), sum(duration)/count(*) from counters_collection group by 1,2").fetchall()
    for k, cn, v, n, d in rows:
        if 'vr::' not in k: continue
        agg.setdefault(k, {})[cn] = (v / n, n)
out = open('$O/mfma_util.txt', 'w')
out.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 on tools/encode_only.py 1 and tools/search_bench.py 1000 (per launch averages)\n")
out.write("# SQ_VALU_MFMA_BUSY_CYCLES is summed over the 4 SIMDs of a CU, SQ_BUSY_CU_CYCLES counts once per CU: MFMA pipe utilisation = ratio / 4\n")
out.write(f"{'kernel':84s} {'launches':>8s} {'MFMA_BUSY':>14s} {'BUSY_CU':>14s} {'ratio':>7s} {'MOPS_BF16':>14s}\n")
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', (0, 0))[0] * kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', (0, 1))[1]):
    mb = d.get('SQ_VALU_MFMA_BUSY_CYCLES', (0, 0)); bc = d.get('SQ_BUSY_CU_CYCLES', (0, 0)); mo = d.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', (0, 0))
    out.write(f"{k[:84]:84s} {mb[1]:8d} {mb[0]:14.0f} {bc[0]:14.0f} {(mb[0] / bc[0] if bc[0] else 0):7.3f} {mo[0]:14.0f}\n")
out.close()
print(open('$O/mfma_util.txt').read()[:3000])
PY
