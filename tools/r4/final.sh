#!/bin/bash
# round 4: round-end evidence from ONE commit on ONE box — the full -m gpu suite (config 3 at 100 000 pages) + tools/round_profiles.sh
cd $GRAFT_REPO_ROOT; O=gpurun_out/round; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/r04_gpu_tests_full.log 2>&1; echo "gpu tests rc=$?"
tail -3 $O/r04_gpu_tests_full.log | tee $O/r04_gpu_tests.log
python __graft_entry__.py smoke 2>&1 | tail -2
bash tools/round_profiles.sh 04 2>&1 | tail -40
