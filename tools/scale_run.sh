#!/bin/bash
# The 1 / 2 / 4 / 8-GPU scaling curve of BASELINE.json configs[3], unattended, on one node:
#   bash tools/scale_run.sh [OUTDIR] [STEPS] [WARMUP]
# For every N that the node has GPUs for it runs bench.py exactly as the driver does —
#   python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N
# (one process per GPU, backend "nccl" = RCCL over xGMI; pages sharded over the ranks with no data-path
# collective, index rows sharded, ONE all-gather of the packed [nq, k] keys per search) — and prints, per N, what
# RCCL saw (backend, world size), `value` (pages/s, whole job), queries_per_sec and the all-gather's own time.
# Efficiency is the reader's to compute from the per-N values.
# Nothing here has been measured on this pool (it hands out 1-GPU boxes): DESIGN.md section 6.
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/scale}; STEPS=${2:-8}; WARMUP=${3:-2}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
NGPU=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
echo "GPUs on this node: $NGPU"
for N in 1 2 4 8; do
  if [ "$N" -gt "$NGPU" ]; then echo "N=$N: skipped ($NGPU GPU(s) here)"; continue; fi
  PORT=$((29500 + N))
  if [ "$N" -eq 1 ]; then
    timeout 1800 python bench.py --gpus 1 --steps "$STEPS" --warmup "$WARMUP" --no-extras > "$OUT/bench_n$N.json" 2> "$OUT/bench_n$N.err"
  else
    timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
      bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARMUP" --no-extras --no-cpu-baseline > "$OUT/bench_n$N.json" 2> "$OUT/bench_n$N.err"
  fi
  python - "$OUT/bench_n$N.json" "$N" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
except Exception as e:
    print(f"N={sys.argv[2]}: no result line ({e}); see the .err file"); sys.exit(0)
ex = (j.get("search") or {}).get("exchange") or {}
print(f"N={j['n_gpus']}: value {j['value']} {j['unit']}  ms/step {j['ms_per_step']}  queries/s {j['queries_per_sec']}  "
      f"backend {ex.get('backend', '-')} world {ex.get('world_size', 1)}  all_gather {ex.get('all_gather_us', '-')} us")
PY
done
