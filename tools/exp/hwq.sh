for q in 2 4 8; do
echo "== GPU_MAX_HW_QUEUES=$q coalesce 8"
GPU_MAX_HW_QUEUES=$q LCD_PROFILE_CHAINS=1 python bench.py --steps 8 --warmup 8 --lanes 1 --coalesce 8 --cpu-sample 0 2>&1 | grep -A1 "class 1024\|metric" | tail -3 | cut -c1-170
done
