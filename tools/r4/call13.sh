#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c13; mkdir -p $O
timeout 900 python tools/hunt_encode.py 300 2>&1 | tail -6 | tee $O/hunt_encode.txt
timeout 600 python tools/hunt_shards.py 2>&1 | tail -4 | tee $O/hunt_shards.txt
timeout 600 python tools/hunt_band.py 100 2>&1 | tail -4 | tee $O/hunt_band.txt
