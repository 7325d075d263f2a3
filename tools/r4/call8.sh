#!/bin/bash
# round 4, call 8: grouped band pass — search tests, hunts, templated timing
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_search.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
timeout 600 python tools/hunt_band.py 60 2>&1 | tail -6 | tee $O/hunt_band.txt
timeout 300 python tools/search_templated.py 1000 10 2>&1 | tail -2 | tee $O/templated.txt
timeout 300 python tools/search_templated.py 100 10 2>&1 | tail -1 | tee -a $O/templated.txt
timeout 300 python tools/search_templated.py 1 10 2>&1 | tail -1 | tee -a $O/templated.txt
