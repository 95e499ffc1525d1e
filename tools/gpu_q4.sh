#!/bin/bash
# re-sort variants: default, never compact (LCD_DBG=64), compact wherever it fits (LCD_DBG=128).  Usage: bash tools/gpu_q4.sh
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"stage_ms\"][\"ms_poa_kernel\"], d[\"digest\"])"; }
for dbg in 0 64 128; do
  echo "LCD_DBG=$dbg ont"; LCD_DBG=$dbg timeout 600 python bench.py --shape ont --steps 48 --cpu-sample 0 --depth-profile 0 --overlap 0 --repeats 1 2>/dev/null | tail -1 | line
  echo "LCD_DBG=$dbg hifi"; LCD_DBG=$dbg timeout 600 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --depth-profile 0 --overlap 0 2>/dev/null | tail -1 | line
done
echo "sv"; timeout 900 python bench.py --shape sv --steps 8 --cpu-sample 0 --depth-profile 0 --overlap 0 --repeats 1 2>/dev/null | tail -1 | line
