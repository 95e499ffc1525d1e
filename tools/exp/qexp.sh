for q in 4 8 24 64 128; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 8 --warmup 4 --lanes 4 --cpu-sample 0 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['stage_ms']['ms_poa_kernel'], d['stage_ms']['ms_total'])
    elif 'rror' in l: print(l.strip())
"
done
