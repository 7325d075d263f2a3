# the torchrun path of bench.py on ONE GPU (two ranks share it, gloo rendezvous): does the N>1 code still run end to end?
cd $GRAFT_REPO_ROOT
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --corpus-pages 4000 --index-rows 20000 > gpurun_out/n2.log 2> gpurun_out/n2.err
echo rc=$?
tail -1 gpurun_out/n2.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('value', 'n_gpus', 'ms_per_step', 'scaling', 'queries_per_sec')})
print(json.dumps(d['search'])[:1500])
print(d.get('corpus_embed'))"
tail -5 gpurun_out/n2.err
