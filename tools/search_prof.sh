#!/bin/bash
# per-kernel breakdown of the search path: rocprofv3 kernel trace over tools/search_bench.py
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sp && rocprofv3 --kernel-trace --stats -d /tmp/sp -o sp -- python $GRAFT_REPO_ROOT/tools/search_bench.py "$@" > /tmp/sp.log 2>&1
tail -4 /tmp/sp.log
db=$(find /tmp/sp -name '*.db' | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $db | grep -v "at::\|elementwise\|Cijk" | head -20
