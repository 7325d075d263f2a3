#!/bin/bash
# Round-6 evidence, ONE box, one process sequence: the -m gpu suite (small corpus: the driver runs the 100k default), the
# driver's bench command line, the rocprofv3 passes of tools/round_profiles.sh, the torchrun path with two ranks on this GPU.
# The LAST GPU call of the round: no kernel commit follows it.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/round; mkdir -p $O
( rocm-smi --showuniqueid --showclocks --showpower 2>/dev/null | grep -v '^=' | head -20; hostname ) > $O/r06_box.txt
VISRAG_TEST_CORPUS_PAGES=3200 timeout 1500 python -m pytest tests -q -m gpu > $O/r06_gpu_tests.log 2>&1
tail -3 $O/r06_gpu_tests.log
bash tools/round_profiles.sh 06 > $O/round_profiles.log 2>&1
tail -25 $O/round_profiles.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 1 --corpus-pages 640 --no-extras --no-cpu-baseline > $O/r06_bench_n2_one_gpu.json 2> $O/n2.err
tail -c 1500 $O/r06_bench_n2_one_gpu.json
