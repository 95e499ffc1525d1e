for cfg in "12 24" "16 24" "16 48" "24 48"; do
  set -- $cfg
  echo "== lanes=$1 queues=$2"
  GPU_MAX_HW_QUEUES=$2 timeout 300 python bench.py --steps $(( $1 * 3 )) --warmup 2 --lanes $1 --cpu-sample 0 2>&1 | tail -3 | cut -c1-160
done
rocm-smi --showmeminfo vram 2>/dev/null | head -5
