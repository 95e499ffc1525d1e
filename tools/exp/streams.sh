timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
for cfg in "4 4 1 8 16" "6 6 1 8 16" "3 3 1 8 16" "4 8 2 8 32" "4 8 2 4 32" "3 6 2 8 32" "4 4 1 16 32"; do
  set -- $cfg
  echo "== streams=$1 hwq=$2 lanes=$3 coalesce=$4 steps=$5"
  LCD_STREAMS=$1 GPU_MAX_HW_QUEUES=$2 timeout 300 python bench.py --steps $5 --warmup $(( $3 * $4 )) --lanes $3 --coalesce $4 --cpu-sample 0 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['stage_ms']['ms_poa_kernel'], d['digest'])
    else: print(l[:200])"
done
