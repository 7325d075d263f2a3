#!/bin/bash
# HBM traffic per kernel launch from PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (no trace domains besides the kernel trace), over
# the encode workload (1 step) and one 1k x 100k search.  Writes gpurun_out/pmc/{table.txt,traffic.json};
# copy them to profiles/rNN_pmc_traffic.txt / rNN_traffic.json.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o enc -- python $R/tools/encode_only.py 1 > /tmp/pmc_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o srch -- python $R/tools/search_bench.py 1000 > /tmp/pmc_${c}_s.log 2>&1
done
python $R/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $R/gpurun_out/pmc
