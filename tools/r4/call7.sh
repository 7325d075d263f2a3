#!/bin/bash
# round 4, call 7: bug hunts on the new fallbacks and GEMM routes
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c7; mkdir -p $O
timeout 600 python tools/hunt_band.py 40 2>&1 | tail -12 | tee $O/hunt_band.txt
timeout 600 python tools/hunt_text_stream.py 80 2>&1 | tail -12 | tee $O/hunt_text_stream.txt
timeout 600 python tools/hunt_search.py 150 2>&1 | tail -12 | tee $O/hunt_search.txt
timeout 300 python tools/search_templated.py 1000 10 2>&1 | tail -2 | tee $O/templated.txt
