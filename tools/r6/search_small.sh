#!/bin/bash
# Round 6: the handful-of-queries search without fallback launches (score rows written by the streaming sweep, flagged queries
# redone in place by their merge workgroup): tests, the hunts, stage times.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6/search_small; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_search.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 900 python tools/hunt_band.py 40 2>&1 | tail -4 | tee $O/hunt_band.log
timeout 900 python tools/hunt_search.py 2>&1 | tail -4 | tee $O/hunt_search.log
timeout 600 python tools/search_stages.py 2>&1 | tail -12 | tee $O/stages.log
timeout 600 python tools/search_bench.py 2>&1 | tail -5 | tee $O/search_bench.log
timeout 600 python tools/search_templated.py 1 50 2>&1 | tail -2 | tee $O/templated_nq1.log
