cd $GRAFT_REPO_ROOT
for rows in 12500 25000; do
python bench.py --steps 3 --warmup 1 --corpus-pages $rows --index-rows $rows --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['search']
print($rows, s['ms_per_search'], s['stages_ms'], {k: v for k, v in s['certification'].items() if k != 'what'}, s['ids_vs_fp64']['queries_with_identical_ids'])"
done
