for sp in 0 96 128; do
echo "== LCD_CU_SPLIT=$sp"
LCD_CU_SPLIT=$sp LCD_PROFILE_CHAINS=1 python bench.py --steps 8 --warmup 8 --lanes 1 --coalesce 8 --cpu-sample 0 2>&1 | grep -A1 "class 1024\|class  512\|metric" | tail -5 | cut -c1-200
done
