#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/c9; mkdir -p $O
timeout 900 python bench.py --steps 10 --warmup 3 --pipelined > $O/bench1.json 2> $O/bench1.err; echo "bench rc=$?"; tail -3 $O/bench1.err; cat $O/bench1.json | cut -c1-3000
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline --index-rows 20000 > $O/bench2.json 2> $O/bench2.err; echo "bench2 rc=$?"; tail -3 $O/bench2.err; cat $O/bench2.json | cut -c1-1500

