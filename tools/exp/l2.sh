for cfg in "1 16 4 4" "2 8 4 8" "2 16 4 8" "2 16 4 4" "3 8 4 4"; do
  set -- $cfg
  echo "== lanes=$1 coalesce=$2 streams=$3 hwq=$4"
  LCD_STREAMS=$3 GPU_MAX_HW_QUEUES=$4 timeout 300 python bench.py --steps 64 --lanes $1 --coalesce $2 --cpu-sample 0 2>&1 | tail -1 | cut -c1-130
done
