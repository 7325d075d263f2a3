#!/bin/bash
# SQ counter passes (own runs, kernel trace only — never combined with other trace domains) over a single-op
# microbench.  usage: pmc_kernel.sh <outdir> <kernel-name-pattern> <cmd...>
# Writes <outdir>/pmc_<pattern>.txt: per-kernel counter averages per dispatch.
OUT=$1; PAT=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
SETS=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES"
 "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F32 SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES"
 "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_CVT SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_CYCLES"
)
: > $R/$OUT/pmc_$PAT.txt
i=0
for S in "${SETS[@]}"; do
  rm -rf /tmp/pmk_$i
  rocprofv3 --pmc $S --kernel-trace -d /tmp/pmk_$i -o p -- "$@" > /tmp/pmk_$i.log 2>&1
  python - "$PAT" /tmp/pmk_$i >> $R/$OUT/pmc_$PAT.txt 2>&1 <<'PY'
import sqlite3, glob, sys
pat, d = sys.argv[1], sys.argv[2]
dbs = glob.glob(d + '/**/*.db', recursive=True)
if not dbs:
    print("# no db in", d); sys.exit(0)
c = sqlite3.connect(dbs[0])
try:
    rows = c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection "
                     "where kernel_name like ? group by 1,2", ('%' + pat + '%',)).fetchall()
except Exception as e:
    print("# query failed:", e); rows = []
for k, cn, v, n in rows:
    print(f"{k[:70]:70s} {cn:28s} {v / n:16.0f} per dispatch ({n} dispatches)")
PY
  tail -1 /tmp/pmk_$i.log | cut -c1-160 >> $R/$OUT/pmc_$PAT.txt
  i=$((i+1))
done
cat $R/$OUT/pmc_$PAT.txt
