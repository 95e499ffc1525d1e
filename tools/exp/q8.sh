for q in 16 24 32; do
  echo "== lanes=8 queues=$q"
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --steps 32 --warmup 2 --lanes 8 --cpu-sample 0 2>&1 | tail -1 | cut -c1-120
done
echo "== lanes=6 queues=24"; timeout 300 python bench.py --steps 24 --warmup 2 --lanes 6 --cpu-sample 0 2>&1 | tail -1 | cut -c1-120
