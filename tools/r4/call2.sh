#!/bin/bash
# round 4, call 2: full -m gpu suite (config 3 at 20k pages), RESID epilogue diagnostics in-model, PIL pipeline host share
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c2; mkdir -p $O
VISRAG_TEST_CORPUS_PAGES=20000 timeout 1200 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $O/summary.txt
tail -4 $O/gpu_tests.log
bash tools/ab_libs.sh $O/ab 2 visrag_amd/libvisrag_hip.so visrag_amd/libvisrag_hip_rd1.so visrag_amd/libvisrag_hip_rd2.so visrag_amd/libvisrag_hip_tch.so 2>&1 | tee $O/ab_summary.txt
timeout 300 python tools/pil_pipeline_bench.py 2048 32 2>/dev/null | tee $O/pil.txt
