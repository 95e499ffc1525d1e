for ne in "" 2.0; do
echo "== LCD_NODE_EST=$ne"
if [ -n "$ne" ]; then export LCD_NODE_EST=$ne; fi
LCD_PROFILE_CHAINS=1 python bench.py --steps 8 --warmup 8 --lanes 1 --coalesce 8 --cpu-sample 0 2>&1 | grep -A1 "class 1024\|class  512\|metric" | tail -5 | cut -c1-200
python bench.py --steps 32 --warmup 16 --lanes 2 --coalesce 8 --cpu-sample 0 2>&1 | tail -1 | cut -c1-100
done
