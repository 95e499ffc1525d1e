for cfg in "1 16 4 4 64" "2 8 4 8 64" "1 24 4 4 72" "1 32 4 4 64" "2 16 4 8 64"; do
  set -- $cfg
  echo "== lanes=$1 coalesce=$2 streams=$3 hwq=$4"
  LCD_STREAMS=$3 GPU_MAX_HW_QUEUES=$4 timeout 400 python bench.py --steps $5 --lanes $1 --coalesce $2 --cpu-sample 0 2>&1 | tail -1 | cut -c1-130
done
