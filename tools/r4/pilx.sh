python -m pytest tests/test_gpu_ops.py -q -x -k "stream_pairs" 2>&1 | tail -2
python tools/stream_pair_probe.py 8 2>/dev/null | grep -v "^streams [0-9],[0-9]*: .*pages/s$" | head -12
python bench.py --steps 10 --warmup 3 --corpus-pages 0 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench', d['value'], d['pipelined'], d['pil_pipeline']['pages_per_sec'], d['pil_pipeline']['vs_pipelined'])"
