#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/rpt; rocprofv3 --kernel-trace --stats -d /tmp/rpt -o s -- python $R/tools/search_bench.py 1000 > /tmp/rpt.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/rpt -name '*.db' | head -1) /tmp/rpt.txt; grep "prepass\|thr_own\|sweep256w\|merge256\|f32_to_bf16_pad" /tmp/rpt.txt | cut -c1-110
cd $R && timeout 600 python -m pytest tests/test_gpu_search.py -x -q -k "prepass or random_shapes" 2>&1 | tail -2
