timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for cfg in "1 1 4" "1 8 16" "2 4 16" "2 8 32" "1 16 32" "2 16 64"; do
  set -- $cfg
  echo "== lanes=$1 coalesce=$2 steps=$3"
  timeout 300 python bench.py --steps $3 --warmup 2 --lanes $1 --coalesce $2 --cpu-sample 0 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['stage_ms']['ms_poa_kernel'], d['digest'], d['roofline']['frac'])
    else: print(l[:200])"
done
