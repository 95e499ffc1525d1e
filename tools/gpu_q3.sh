#!/bin/bash
# chain profiles of the noisy shapes and of a lone batch.  Usage: bash tools/gpu_q3.sh <tag>
tag=${1:-q3}
mkdir -p gpurun_out
LCD_PROFILE_CHAINS=1 timeout 600 python bench.py --shape ont --steps 48 --cpu-sample 0 --depth-profile 0 --overlap 0 --repeats 1 > gpurun_out/${tag}_ont.json 2> gpurun_out/${tag}_ont.err
LCD_PROFILE_CHAINS=1 timeout 600 python bench.py --shape sv --steps 8 --cpu-sample 0 --depth-profile 0 --overlap 0 --repeats 1 > gpurun_out/${tag}_sv.json 2> gpurun_out/${tag}_sv.err
LCD_PROFILE_CHAINS=1 LCD_TIME_HOST=1 timeout 600 python bench.py --steps 1 --warmup 1 --coalesce 1 --cpu-sample 0 --depth-profile 0 --overlap 0 --repeats 1 > gpurun_out/${tag}_d1.json 2> gpurun_out/${tag}_d1.err
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --overlap 0 --depth-profile 1 > gpurun_out/${tag}_depth.json 2>/dev/null
for s in ont sv d1 depth; do python - gpurun_out/${tag}_$s.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); print(sys.argv[1], j["value"], j["digest"], j["stage_ms"], j.get("depth"), j["device_memory"])
PY
done
